// The two 16-channel layers of the head of the pyramid on the BF16 matrix pipe with exactly split fp32 operands (round 4):
//   nn.SpatialConvolution(16,16,3,3,1,1,1,1) + LeakyReLU(0.2)   second conv of the level-2 convUnit (/root/reference/models/pwc.lua:62)
// Same interface, weight packing and tile walk as conv3x3_c16_kernel (b2f_conv16.hip), which runs the layer on the fp32 MFMA at 76 %
// of that pipe -- the fp32 MFMA is the bound there, and it holds the SIMD's VALU while it runs.  Here every fp32 operand is split
// exactly into three bf16 terms (x = xh + xm + xl, round to nearest even) and six of the nine term products are kept (all but
// m*l, l*m, l*l: relative error ~2^-24 per product, fp32 accumulation -- the scheme of b2f_wino4s.hip / b2f_wino2s.hip):
//   D[co 16][pixel 16] += W[co][k 32] X[k 32][pixel]      v_mfma_f32_16x16x32_bf16, 16 cycles, 1 024 FLOP/clk/SIMD
// A lane (row / column = lane & 15, kg = lane >> 4) holds the 4 channels 4 kg .. 4 kg + 3 of its output channel / pixel as two
// overlapping windows of bf16 pairs, Wa = [m01 m23 h01 h23], Wb = [h01 h23 l01 l23] (Xa, Xb alike); the K = 32 of one MFMA is
// (4 channel quads) x (2 terms x 4 channels) and a tap costs three MFMAs:  Wa Xa = Wm Xm + Wh Xh,  Wb Xa = Wh Xm + Wl Xh,
// Wa Xb = Wm Xh + Wh Xl.  Per 16 pixels x 16 outputs: 27 MFMAs x 16 cycles = 432 cycles against 36 x 32 = 1 152 on the fp32 pipe.
//   * the patch (18 x 34 pixels x 16 channels) is split ONCE while it is staged: global -> registers (requested a tile ahead,
//     under the MFMAs) -> 22 VALU per channel quad -> LDS as two planes [window a | b][kg][pixel] of 16 bytes (80 KB, two
//     blocks per CU: one block's split pass runs under the other's MFMAs -- the bf16 MFMA leaves the VALU free);
//   * a wave owns 8 output rows x 16 columns and sweeps the 10 patch rows under them: the two windows of (patch row, kx) are
//     read once and feed the three output rows they belong to (ky = 0, 1, 2): 60 ds_read_b128 per 216 MFMAs;
//   * the weight windows of the lane's (co, kg) are split in the kernel's prologue from the fp32 packing of the fp32 kernel
//     ([tap][kg][co][4]) and stay in 72 VGPRs.
#include "b2f_internal.h"

namespace b2f {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace c16b {
constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;          // 612
constexpr int PLANE = 628;             // 16-byte slots per (window, kg) plane: 628 * 16 B = 64 B mod 256, the four kg planes start 16 banks apart
constexpr int LDS_BYTES = 2 * 4 * PLANE * 16;   // 80 384: two blocks per CU
constexpr int NITEM = 4 * NPIX;        // (pixel, channel quad) staging items per tile: 2 448
constexpr int NJ = (NITEM + 255) / 256;   // 10 per thread
}  // namespace c16b

__device__ __forceinline__ unsigned c16b_pk(float a, float b)
{
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));   // one v_cvt_pk_bf16_f32 (RNE)
}
// fp32 quad -> windows a = [m01 m23 h01 h23], b = [h01 h23 l01 l23]  (v = h + m + l exactly)
__device__ __forceinline__ void c16b_split(const f32x4 v, u32x4 &wa, u32x4 &wb)
{
    const unsigned h01 = c16b_pk(v[0], v[1]), h23 = c16b_pk(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, h01 << 16), r1 = v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h23 << 16), r3 = v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u);
    const unsigned m01 = c16b_pk(r0, r1), m23 = c16b_pk(r2, r3);
    const float l0 = r0 - __builtin_bit_cast(float, m01 << 16), l1 = r1 - __builtin_bit_cast(float, m01 & 0xffff0000u);
    const float l2 = r2 - __builtin_bit_cast(float, m23 << 16), l3 = r3 - __builtin_bit_cast(float, m23 & 0xffff0000u);
    wa = u32x4{m01, m23, h01, h23};
    wb = u32x4{h01, h23, c16b_pk(l0, l1), c16b_pk(l2, l3)};
}

__global__ __launch_bounds__(256, 2) void conv3x3_c16b_kernel(const ConvLaunch p, const int ntiles, const int tiles_x, const int tiles_y)
{
    using namespace c16b;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *L = reinterpret_cast<u32x4 *>(smem);                             // [window 2][kg 4][PLANE]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;

    // weight windows of (co = n, channel quad kg), 9 taps
    u32x4 wa[9], wb[9];
    {
        const f32x4 *wp = reinterpret_cast<const f32x4 *>(p.wpk) + kg * 16 + n;
#pragma unroll
        for (int t = 0; t < 9; ++t) c16b_split(wp[t * 64], wa[t], wb[t]);
    }
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + 4 * kg);

    // staging items of this thread: i = tid + 256 j = ((chunk * NPIX + pixel) * 2 + half): lanes 2 i, 2 i + 1 fetch the two 16-byte
    // halves of a pixel's 8-channel chunk (whole 32-byte pieces, consecutive pixels contiguous)
    int s_pyx[NJ], s_dst[NJ], s_goff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = tid + 256 * j;
        const int half = i & 1, cp = i >> 1;
        const int ch = cp >= NPIX ? 1 : 0, pix = cp - ch * NPIX;
        const int py = pix / PW, px = pix - py * PW;
        const bool on = i < NITEM;
        s_pyx[j] = on ? (py << 8 | px) : -1;
        s_dst[j] = (2 * ch + half) * PLANE + pix;
        s_goff[j] = ch * (int)p.seg[0].chunk_stride + half * 4;
    }

    struct Tile { int img, ox0, oy0; };
    auto decode = [&](int v) {
        int bid = xcd_remap(v, ntiles);
        Tile t;
        const int tx_i = bid % tiles_x;
        bid /= tiles_x;
        const int ty_i = bid % tiles_y;
        t.img = bid / tiles_y;
        t.ox0 = tx_i * TW; t.oy0 = ty_i * TH;
        return t;
    };
    f32x4 raw[NJ];
    auto request = [&](const Tile &t) {
        const float *base = p.seg[0].ptr + (size_t)t.img * p.seg[0].img_stride;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int gy = t.oy0 - 1 + (s_pyx[j] >> 8), gx = t.ox0 - 1 + (s_pyx[j] & 255);
            const bool ok = s_pyx[j] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            raw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) raw[j] = *reinterpret_cast<const f32x4 *>(base + (size_t)(gy * p.W + gx) * p.seg[0].pix_stride + s_goff[j]);
        }
    };

    const int xt = wave & 1, rh = wave >> 1;                                // this wave: columns 16 xt .. + 15, output rows 8 rh .. + 7
    const u32x4 *xa_base = L + kg * PLANE + (rh * 8) * PW + xt * 16 + n;    // window a of (patch row 8 rh, patch column 16 xt + n)
    const int G = gridDim.x;
    int v = blockIdx.x;
    Tile cur = decode(v < ntiles ? v : 0);
    if (v < ntiles) request(cur);
    for (; v < ntiles; v += G) {
        // ---- split the staged patch into LDS ----
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (s_pyx[j] >= 0) {
                u32x4 a, b;
                c16b_split(raw[j], a, b);
                L[s_dst[j]] = a;
                L[4 * PLANE + s_dst[j]] = b;
            }
        }
        __syncthreads();
        const int vn = v + G;
        const Tile nxt = vn < ntiles ? decode(vn) : cur;
        if (vn < ntiles) request(nxt);                                      // lands under the MFMAs below
        // ---- multiply: sweep the 10 patch rows under this wave's 8 output rows ----
        f32x4 acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 10; ++r)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const bf16x8 xa = __builtin_bit_cast(bf16x8, xa_base[r * PW + kx]);
                const bf16x8 xb = __builtin_bit_cast(bf16x8, xa_base[4 * PLANE + r * PW + kx]);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int o = r - ky;
                    if (o >= 0 && o < 8) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, wa[ky * 3 + kx]), b = __builtin_bit_cast(bf16x8, wb[ky * 3 + kx]);
                        acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xa, acc[o], 0, 0, 0);
                        acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, xa, acc[o], 0, 0, 0);
                        acc[o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb, acc[o], 0, 0, 0);
                    }
                }
            }
        // ---- lane (n, kg) holds D[co = 4 kg + r][pixel n]: bias, LeakyReLU, one 16-byte store per output row ----
        float *ob = p.out + (size_t)cur.img * p.out_img_stride + (size_t)(kg >> 1) * p.out_chunk_stride + (kg & 1) * 4;
        const int ox = cur.ox0 + xt * 16 + n;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            f32x4 o4 = acc[o] + bias;
            if (p.leaky) o4 = __builtin_elementwise_max(o4, 0.2f * o4);
            const int oy = cur.oy0 + rh * 8 + o;
            if (oy < p.H && ox < p.W) *reinterpret_cast<f32x4 *>(ob + ((size_t)oy * p.W + ox) * p.out_pix_stride) = o4;
        }
        __syncthreads();                                                    // everyone is done reading the patch
        cur = nxt;
    }
}

hipError_t launch_conv3x3_c16b(const ConvLaunch &p, hipStream_t s)
{
    using namespace c16b;
    static bool attr_done_dev[64] = {false};
    static int n_cu_dev[64] = {0};
    const int slot = attr_slot();
    bool &attr_done = attr_done_dev[slot];
    int &n_cu = n_cu_dev[slot];
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * p.nimg;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_c16b_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (!n_cu) {
        int dev = 0, nn = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&nn, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || nn <= 0) nn = 256;
        nn &= ~7;
        n_cu = nn < 8 ? 8 : nn;
    }
    int grid = 2 * n_cu < ntiles ? 2 * n_cu : ntiles;
    if (grid >= 8) grid &= ~7;
    hipLaunchKernelGGL(conv3x3_c16b_kernel, dim3((unsigned)grid), dim3(256), LDS_BYTES, s, p, ntiles, tiles_x, tiles_y);
    return hipGetLastError();
}

}  // namespace b2f
