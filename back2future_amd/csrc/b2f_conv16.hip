// nn.SpatialConvolution(16,16,3,3,1,1,1,1) + LeakyReLU(0.2): the second conv of the level-2 convUnit
// (/root/reference/models/pwc.lua:62 with 16 planes), the largest map of the pyramid (H/2 x W/2, three frames).
// With 16 outputs half of a 32-wide MFMA N tile would be padding and K = 144 is too short for the chunk pipeline
// of the general kernels to pay off (their fixed per-block cost dominates: 2.0-2.3 ms per step), so this
// shape gets its own single-pass direct kernel on v_mfma_f32_16x16x4_f32 (exact fp32, same 256 FLOP/clk/CU):
//   D[co 16][pixel 16] += A[co][k 4] * B[k][pixel],  K = 9 taps x 16 channels = 36 MFMAs per 16 pixels.
// Block = 256 threads -> 16 x 32 output pixels; the 18 x 34 x 16-channel patch is staged once into LDS as
// [channel quad kg 4][pixel] float4 (39 KB, four blocks per CU); lane (n = l & 15, kg = l >> 4) reads the
// float4 of pixel n / quad kg per tap (conflict-free) and feeds 4 MFMAs (k = 4 kg + j); the weights of
// the lane's (co = l & 15, kg) stay in 36 VGPRs for the whole kernel.  A lane ends up with 4 consecutive output
// channels of one pixel: bias, LeakyReLU, one 16-byte store.  Chunk-planar in and out.
#include "b2f_internal.h"

#include <vector>

namespace b2f {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace c16 {
constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;          // 612
constexpr int PLANE = 628;             // float4 per channel-quad plane (>= NPIX; 628 * 16 B = 64 B mod 256: the four planes start 16 banks apart)
}  // namespace c16

__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const ConvLaunch p)
{
    using namespace c16;
    __shared__ f32x4 patch[4][PLANE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kg = lane >> 4;

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;
    const int ox0 = tx_i * TW, oy0 = ty_i * TH;

    // weights of (co = n, channel quad kg): 9 taps x float4 (ci = 4 kg + j)
    f32x4 wv[9];
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(p.wpk) + kg * 16 + n;
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = wp[t * 64];
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + 4 * kg);

    // ---- stage the patch: item = (pixel, kg); channel quad kg lives in chunk kg >> 1, half kg & 1 ----
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.seg[0].ptr + (size_t)img * p.seg[0].img_stride), 0, 0x7fffffff, 0x00020000);
    // 2448 items = 9.6 per thread: all loads (from clamped addresses, no branch) are issued before the first LDS
    // write, so the block pays one memory round trip instead of ten
    {
        constexpr int NIT = (4 * NPIX + 255) / 256;
        f32x4 v[NIT];
        bool ok[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * 256;
            const int pix = i >> 2, q = i & 3;
            const int py = pix / PW, px = pix - py * PW;
            const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
            ok[k] = i < 4 * NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const unsigned gp = ok[k] ? (unsigned)(gy * p.W + gx) : 0u;
            v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsrc, (int)((gp * (unsigned)p.seg[0].pix_stride + (q & 1) * 4 + (unsigned)(q >> 1) * (unsigned)p.seg[0].chunk_stride) * 4u), 0, 0));
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + k * 256;
            if (i < 4 * NPIX) patch[i & 3][i >> 2] = ok[k] ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();

    float *ob = p.out + (size_t)img * p.out_img_stride + (size_t)(kg >> 1) * p.out_chunk_stride + (kg & 1) * 4;
    // 32 tiles of 16 pixels (16 rows x 2 column halves), 8 per wave
#pragma unroll 2
    for (int q = 0; q < 8; ++q) {
        const int mt = wave * 8 + q;
        const int ry = mt >> 1, xt = mt & 1;
        const f32x4 *pp = &patch[kg][ry * PW + xt * 16 + n];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 a = pp[ky * PW + kx];
                const f32x4 w = wv[ky * 3 + kx];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], a[j], acc, 0, 0, 0);
            }
        // lane (n, kg) holds D[co = 4 kg + r][pixel n], r = 0..3
        f32x4 v = acc + bias;
        if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);
        const int oy = oy0 + ry, ox = ox0 + xt * 16 + n;
        if (oy < p.H && ox < p.W) *reinterpret_cast<f32x4 *>(ob + ((size_t)oy * p.W + ox) * p.out_pix_stride) = v;
    }
}

bool c16_supported(const ConvLaunch &p)
{
    return p.stride == 1 && p.cout == 16 && p.nseg == 1 && p.seg[0].nchunks == 2 && p.H == p.Ho && p.W == p.Wo &&
           (p.seg[0].pix_stride & 3) == 0 && (p.seg[0].chunk_stride & 3) == 0 && (p.out_pix_stride & 3) == 0 &&
           (p.out_chunk_stride & 3) == 0 &&
           ((double)p.H * p.W * p.seg[0].pix_stride + (double)p.seg[0].chunk_stride) * 4.0 < 2147483648.0;   // 32-bit buffer offsets
}

hipError_t launch_conv3x3_c16(const ConvLaunch &p, hipStream_t s)
{
    if (!c16_supported(p)) return hipErrorInvalidValue;
    const int tiles = ((p.W + c16::TW - 1) / c16::TW) * ((p.H + c16::TH - 1) / c16::TH);
    hipLaunchKernelGGL(conv3x3_c16_kernel, dim3((unsigned)(tiles * p.nimg)), dim3(256), 0, s, p);
    return hipGetLastError();
}

size_t c16_wpk_floats() { return 9 * 4 * 16 * 4; }

// [tap 9][kg 4][co 16][j 4] = W[co][ci = 4 kg + j][tap]; channels missing from cin_map get zero weights
void c16_pack_weights(const float *w, const float *b, int Ci, const int *cin_map, float *wpk, float *bpk)
{
    for (int t = 0; t < 9; ++t)
        for (int kg = 0; kg < 4; ++kg)
            for (int co = 0; co < 16; ++co)
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * kg + j;
                    const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                    wpk[((t * 4 + kg) * 16 + co) * 4 + j] = ci >= 0 ? w[((size_t)co * Ci + ci) * 9 + t] : 0.f;
                }
    for (int i = 0; i < 16; ++i) bpk[i] = b[i];
}

}  // namespace b2f
