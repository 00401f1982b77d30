// nn.SpatialConvolution(16,16,3,3,1,1,1,1) + LeakyReLU(0.2): the second conv of the level-2 convUnit
// (/root/reference/models/pwc.lua:62 with 16 planes), the largest map of the pyramid (H/2 x W/2, three frames).
// With 16 outputs half of a 32-wide MFMA N tile would be padding and K = 144 is too short for the chunk pipeline
// of the general kernels to pay off (their fixed per-block cost dominates: 2.0-2.3 ms per step), so this
// shape gets its own single-pass direct kernel on v_mfma_f32_16x16x4_f32 (exact fp32, same 256 FLOP/clk/CU):
//   D[co 16][pixel 16] += A[co][k 4] * B[k][pixel],  K = 9 taps x 16 channels = 36 MFMAs per 16 pixels.
// Block = 256 threads -> 16 x 32 output pixels; the 18 x 34 x 16-channel patch is staged once into LDS as
// [channel quad kg 4][pixel] float4 (39 KB; round 3: double-buffered, two persistent blocks per CU); lane (n = l & 15, kg = l >> 4) reads the
// float4 of pixel n / quad kg per tap (conflict-free) and feeds 4 MFMAs (k = 4 kg + j); the weights of
// the lane's (co = l & 15, kg) stay in 36 VGPRs for the whole kernel.  A lane ends up with 4 consecutive output
// channels of one pixel: bias, LeakyReLU, one 16-byte store.  Chunk-planar in and out.
#include "b2f_internal.h"

#include <vector>

namespace b2f {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef int i32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA of one piece: lane i's 16 bytes at (resource base + voff) land at LDS byte address lds + 16 i; an offset past the
// resource's size reads as zero.  Invisible to the compiler's s_waitcnt bookkeeping: the kernels wait themselves.
__device__ __forceinline__ void c16_dma(int voff, i32x4 rsrc, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane((int)lds)) : "memory");
}

namespace c16 {
constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;          // 612
constexpr int PLANE = 628;             // float4 per channel-quad plane (>= NPIX; 628 * 16 B = 64 B mod 256: the four planes start 16 banks apart)
constexpr int NPC = (NPIX + 63) / 64;  // 1-KB DMA pieces per plane: 10 (the last one 36 lanes wide)
constexpr int BUF_F4 = 4 * PLANE;
constexpr int LDS_BYTES = 2 * BUF_F4 * 16;   // 80 384: two blocks per CU
}  // namespace c16

// Round 3: persistent (two blocks per CU walk the tiles), the patch of the NEXT tile brought into the other half of a
// double-buffered LDS patch by LDS-DMA (wave w = channel quad w, one 1-KB piece per tap under the MFMAs of the first M tiles;
// out-of-image pixels read as zero through the buffer resource's range check) -- see conv3x3_c16s2_kernel below, same scheme.
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const ConvLaunch p, const int ntiles, const int tiles_x, const int tiles_y)
{
    using namespace c16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const f32x4 *L = reinterpret_cast<const f32x4 *>(smem);                 // [2][4][PLANE]
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;

    // weights of (co = n, channel quad kg): 9 taps x float4 (ci = 4 kg + j)
    f32x4 wv[9];
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(p.wpk) + kg * 16 + n;
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = wp[t * 64];
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + 4 * kg);

    // DMA items of this wave: channel quad `wave` (chunk wave >> 1, half wave & 1), patch pixel 64 j + lane
    int pyx[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int slot = 64 * j + lane;
        const int py = slot / PW, px = slot - py * PW;
        pyx[j] = slot < NPIX ? (py << 8 | px) : -1;
    }
    const int qoff = ((wave & 1) * 4 + (wave >> 1) * (int)p.seg[0].chunk_stride) * 4;
    const int img_bytes = (int)(p.seg[0].img_stride * 4);

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
    auto resource = [&](const Tile &t) {
        const unsigned long long base = reinterpret_cast<unsigned long long>(p.seg[0].ptr + (size_t)t.img * p.seg[0].img_stride);
        const i32x4 rsrc = {(int)(unsigned)base, (int)((unsigned)(base >> 32) & 0xffffu), img_bytes, 0x00020000};
        return rsrc;
    };
    auto issue_piece = [&](const Tile &t, const i32x4 rsrc, int buf, int j) {
        if (pyx[j] >= 0) {
            const int gy = t.oy0 - 1 + (pyx[j] >> 8), gx = t.ox0 - 1 + (pyx[j] & 255);
            const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const int off = ok ? (gy * p.W + gx) * (int)p.seg[0].pix_stride * 4 + qoff : 0x7ffffff0;   // past the image: zero
            c16_dma(off, rsrc, lds0 + (unsigned)((buf * BUF_F4 + wave * PLANE + 64 * j) * 16));
        }
    };

    const int G = gridDim.x;
    int v = blockIdx.x;
    Tile cur = decode(v < ntiles ? v : 0);
    if (v < ntiles) {
        const i32x4 r0 = resource(cur);
#pragma unroll
        for (int j = 0; j < NPC; ++j) issue_piece(cur, r0, 0, j);
    }
    static_assert(NPC <= 18, "the pieces are issued under the taps of the first two M tiles");
    bool prev_full = false;            // the previous tile issued all eight stores of a lane (seven of them behind this tile's last piece)
    int buf = 0;
    for (; v < ntiles; v += G, buf ^= 1) {
        // this wave's pieces of the current tile have landed (vmcnt counts in order: the last piece was issued in front of the
        // stores of M tiles 1..7 of the previous tile, which may stay in flight when there are exactly seven of them)
        if (prev_full) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();               // everyone's pieces landed; everyone is done reading the other buffer
        const int vn = v + G;
        const bool more = vn < ntiles;
        const Tile nxt = more ? decode(vn) : cur;
        const i32x4 rn = resource(nxt);
        const f32x4 *patch = L + buf * BUF_F4 + kg * PLANE;
        float *ob = p.out + (size_t)cur.img * p.out_img_stride + (size_t)(kg >> 1) * p.out_chunk_stride + (kg & 1) * 4;
        // 32 tiles of 16 pixels (16 rows x 2 column halves), 8 per wave
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int mt = wave * 8 + q;
            const int ry = mt >> 1, xt = mt & 1;
            const f32x4 *pp = patch + ry * PW + xt * 16 + n;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int step = q * 9 + ky * 3 + kx;
                    if (more && step < NPC) issue_piece(nxt, rn, buf ^ 1, step);     // one piece per tap, the first NPC taps
                    const f32x4 a = pp[ky * PW + kx];
                    const f32x4 w = wv[ky * 3 + kx];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], a[j], acc, 0, 0, 0);
                }
            // lane (n, kg) holds D[co = 4 kg + r][pixel n], r = 0..3
            f32x4 o4 = acc + bias;
            if (p.leaky) o4 = __builtin_elementwise_max(o4, 0.2f * o4);
            const int oy = cur.oy0 + ry, ox = cur.ox0 + xt * 16 + n;
            if (oy < p.H && ox < p.W) *reinterpret_cast<f32x4 *>(ob + ((size_t)oy * p.W + ox) * p.out_pix_stride) = o4;
        }
        prev_full = cur.oy0 + TH <= p.H && cur.ox0 + TW <= p.W;
        cur = nxt;
    }
}

bool c16_supported(const ConvLaunch &p)
{
    return p.stride == 1 && p.cout == 16 && p.nseg == 1 && p.seg[0].nchunks == 2 && p.H == p.Ho && p.W == p.Wo &&
           (p.seg[0].pix_stride & 3) == 0 && (p.seg[0].chunk_stride & 3) == 0 && (p.out_pix_stride & 3) == 0 &&
           (p.out_chunk_stride & 3) == 0 &&
           (double)p.seg[0].img_stride * 4.0 < 2147483632.0;   // 32-bit buffer offsets; the image is the buffer resource
}

// persistent launch shared by the two kernels of this file: BPC blocks per CU (fewer when the launch has fewer tiles)
template <typename K>
static hipError_t launch_persistent(K kernel, const ConvLaunch &p, int tiles_x, int tiles_y, int lds_bytes, int bpc, bool &attr_done, int &n_cu, hipStream_t s)
{
    const int ntiles = tiles_x * tiles_y * p.nimg;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (!n_cu) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n &= ~7;                                  // the XCD banding of the logical tile index wants a multiple of 8
        n_cu = n < 8 ? 8 : n;
    }
    int grid = bpc * n_cu < ntiles ? bpc * n_cu : ntiles;
    if (grid >= 8) grid &= ~7;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), lds_bytes, s, p, ntiles, tiles_x, tiles_y);
    return hipGetLastError();
}

hipError_t launch_conv3x3_c16(const ConvLaunch &p, hipStream_t s)
{
    if (!c16_supported(p)) return hipErrorInvalidValue;
#if B2F_EXPERIMENTS
    if (p.bf16_direct) return launch_conv3x3_c16b(p, s);
#endif
    static bool attr_done_dev[64] = {false};
    static int n_cu_dev[64] = {0};
    const int slot = attr_slot();
    return launch_persistent(conv3x3_c16_kernel, p, (p.W + c16::TW - 1) / c16::TW, (p.H + c16::TH - 1) / c16::TH, c16::LDS_BYTES, 2,
                             attr_done_dev[slot], n_cu_dev[slot], s);
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

// nn.SpatialConvolution(16,32,3,3,2,2,1,1) + LeakyReLU(0.2): the first conv of the level-3 convUnit (pwc.lua:60 with 16 -> 32
// planes), H/2 x W/2 -> H/4 x W/4 on three frames.  K = 144 again: the general stride-2 kernel walks it as two 8-channel chunks
// with a barrier each and spends more time in its per-block prologue than in its 72 MFMAs per wave (0.69 ms per step at 3.3 TB/s
// and half the matrix pipe).  Single-pass like the kernel above, and persistent: two blocks per CU walk the tiles (4 x 32 output
// pixels x 32 channels each); the 9 x 65 x 16-channel input patch of the NEXT tile is brought into the other half of a
// double-buffered LDS patch ([channel quad][pixel] float4, 37 KB per buffer) by LDS-DMA (buffer_load_dwordx4 ... lds: no
// registers, out-of-image pixels read as zero through the buffer's range check), one piece per tap under the MFMAs of the
// current tile; wave w = output row w and the DMA of channel quad w, lane (n, kg) reads pixel 2 n of its 16-pixel M tile (stride
// 2 in LDS) and feeds 2 x 4 MFMAs per tap -- the weights of the lane's (co = n and 16 + n, kg) stay in 72 VGPRs for the whole
// kernel.  Measured (batch 16 x 3 frames, 512 x 960 -> 256 x 480): general kernel 0.69 ms; this kernel one tile per block,
// register staging 0.677; persistent + DMA 0.577; + pieces spread over the taps 0.555 = 4.4 TB/s of patch reads + output writes
// (the memory side is the limit now: -DB2F_C16S2_TH=2, 2 x 32 tiles and three blocks per CU, runs the same 0.56).
#ifndef B2F_C16S2_TH
#define B2F_C16S2_TH 4
#endif
namespace c16s2 {
constexpr int TH = B2F_C16S2_TH, TW = 32, PH = 2 * TH + 1, PW = 2 * TW + 1;   // TH 4: wave = output row; TH 2: wave = (row, column half)
static_assert(TH == 2 || TH == 4, "two or four output rows per tile");
constexpr int NXT = TH == 4 ? 2 : 1;   // 16-pixel M tiles of a wave
constexpr int BLOCKS_PER_CU = TH == 4 ? 2 : 3;
constexpr int NPIX = PH * PW;          // 585 | 325
constexpr int PLANE = TH == 4 ? 588 : 332;   // float4 per channel-quad plane: x 16 B = 192 B mod 256, the four planes start 16 banks apart
constexpr int NPC = (NPIX + 63) / 64;  // 1-KB DMA pieces per plane: 10 | 6 (the last one partly masked)
static_assert(NPC <= 9 * NXT, "one piece per tap");
constexpr int BUF_F4 = 4 * PLANE;
constexpr int LDS_BYTES = 2 * BUF_F4 * 16;   // 75 264 | 42 496
}  // namespace c16s2

__global__ __launch_bounds__(256) void conv3x3_c16s2_kernel(const ConvLaunch p, const int ntiles, const int tiles_x, const int tiles_y)
{
    using namespace c16s2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const f32x4 *L = reinterpret_cast<const f32x4 *>(smem);                 // [2][4][PLANE]
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;

    // weights of (co = 16 c2 + n, channel quad kg): 2 x 9 taps x float4 (ci = 4 kg + j)
    f32x4 wv[2][9];
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(p.wpk) + kg * 16 + n;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        wv[0][t] = wp[(t * 2) * 64];
        wv[1][t] = wp[(t * 2 + 1) * 64];
    }
    const f32x4 bias0 = *reinterpret_cast<const f32x4 *>(p.bias + 4 * kg), bias1 = *reinterpret_cast<const f32x4 *>(p.bias + 16 + 4 * kg);

    // DMA items of this wave: channel quad `wave` (chunk wave >> 1, half wave & 1), patch pixel 64 j + lane
    int pyx[NPC];
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int slot = 64 * j + lane;
        const int py = slot / PW, px = slot - py * PW;
        pyx[j] = slot < NPIX ? (py << 8 | px) : -1;
    }
    const int qoff = ((wave & 1) * 4 + (wave >> 1) * (int)p.seg[0].chunk_stride) * 4;
    const int img_bytes = (int)(p.seg[0].img_stride * 4);

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
    // raw buffer resource of a tile's image: {base lo, base hi (stride 0), bytes, flags as __builtin_amdgcn_make_buffer_rsrc sets them}
    auto resource = [&](const Tile &t) {
        const unsigned long long base = reinterpret_cast<unsigned long long>(p.seg[0].ptr + (size_t)t.img * p.seg[0].img_stride);
        const i32x4 rsrc = {(int)(unsigned)base, (int)((unsigned)(base >> 32) & 0xffffu), img_bytes, 0x00020000};
        return rsrc;
    };
    auto issue_piece = [&](const Tile &t, const i32x4 rsrc, int buf, int j) {
        if (pyx[j] >= 0) {
            const int gy = 2 * t.oy0 - 1 + (pyx[j] >> 8), gx = 2 * t.ox0 - 1 + (pyx[j] & 255);
            const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const int off = ok ? (gy * p.W + gx) * (int)p.seg[0].pix_stride * 4 + qoff : 0x7ffffff0;   // past the image: zero
            c16_dma(off, rsrc, lds0 + (unsigned)((buf * BUF_F4 + wave * PLANE + 64 * j) * 16));
        }
    };

    const int G = gridDim.x;
    int v = blockIdx.x;
    Tile cur = decode(v < ntiles ? v : 0);
    if (v < ntiles) {
        const i32x4 r0 = resource(cur);
#pragma unroll
        for (int j = 0; j < NPC; ++j) issue_piece(cur, r0, 0, j);
    }
    bool prev_full = false;            // the previous tile issued both stores of its second half (behind this tile's DMA in the queue)
    int buf = 0;
    for (; v < ntiles; v += G, buf ^= 1) {
        // this wave's pieces of the current tile have landed (vmcnt counts in order: piece 9 was issued before the previous tile's
        // last two stores, which may stay in flight when there are exactly two of them)
        if (prev_full) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();               // everyone's pieces landed; everyone is done reading the other buffer
        // the next tile's pieces are issued one per tap inside the first MFMA loops below: ten at once fill the CU's vector-memory
        // queue and a wave that stands at issue feeds no MFMAs; later than the first half of the tile they would not land in time
        const int vn = v + G;
        const bool more = vn < ntiles;
        const Tile nxt = more ? decode(vn) : cur;
        const i32x4 rn = resource(nxt);
        const f32x4 *patch = L + buf * BUF_F4 + kg * PLANE;
        float *ob = p.out + (size_t)cur.img * p.out_img_stride + (size_t)(kg >> 1) * p.out_chunk_stride + (kg & 1) * 4;
        const int row = TH == 4 ? wave : (wave >> 1);
        const int oy = cur.oy0 + row;
#pragma unroll
        for (int xi = 0; xi < NXT; ++xi) {
            const int xt = TH == 4 ? xi : (wave & 1);
            const f32x4 *pp = patch + (2 * row) * PW + 2 * (xt * 16 + n);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int step = xi * 9 + ky * 3 + kx;
                    if (more && step < NPC) issue_piece(nxt, rn, buf ^ 1, step);     // one piece per tap, the first NPC taps
                    const f32x4 a = pp[ky * PW + kx];
                    const f32x4 w0 = wv[0][ky * 3 + kx], w1 = wv[1][ky * 3 + kx];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], a[j], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], a[j], acc1, 0, 0, 0);
                    }
                }
            // lane (n, kg) holds D[co = 16 c2 + 4 kg + r][pixel n], r = 0..3
            f32x4 v0 = acc0 + bias0, v1 = acc1 + bias1;
            if (p.leaky) {
                v0 = __builtin_elementwise_max(v0, 0.2f * v0);
                v1 = __builtin_elementwise_max(v1, 0.2f * v1);
            }
            const int ox = cur.ox0 + xt * 16 + n;
            if (oy < p.Ho && ox < p.Wo) {
                float *o = ob + ((size_t)oy * p.Wo + ox) * p.out_pix_stride;
                *reinterpret_cast<f32x4 *>(o) = v0;
                *reinterpret_cast<f32x4 *>(o + 2 * p.out_chunk_stride) = v1;
            }
        }
        prev_full = cur.oy0 + TH <= p.Ho && cur.ox0 + TW <= p.Wo;
        cur = nxt;
    }
}

bool c16s2_supported(const ConvLaunch &p)
{
    return p.stride == 2 && p.cout == 32 && p.nseg == 1 && p.seg[0].nchunks == 2 && p.Ho == (p.H - 1) / 2 + 1 && p.Wo == (p.W - 1) / 2 + 1 &&
           (p.seg[0].pix_stride & 3) == 0 && (p.seg[0].chunk_stride & 3) == 0 && (p.out_pix_stride & 3) == 0 &&
           (p.out_chunk_stride & 3) == 0 &&
           (double)p.seg[0].img_stride * 4.0 < 2147483632.0;   // 32-bit buffer offsets; the image is the buffer resource
}

hipError_t launch_conv3x3_c16s2(const ConvLaunch &p, hipStream_t s)
{
    if (!c16s2_supported(p)) return hipErrorInvalidValue;
    static bool attr_done_dev[64] = {false};
    static int n_cu_dev[64] = {0};
    const int slot = attr_slot();
    return launch_persistent(conv3x3_c16s2_kernel, p, (p.Wo + c16s2::TW - 1) / c16s2::TW, (p.Ho + c16s2::TH - 1) / c16s2::TH, c16s2::LDS_BYTES,
                             c16s2::BLOCKS_PER_CU, attr_done_dev[slot], n_cu_dev[slot], s);
}

size_t c16s2_wpk_floats() { return 9 * 2 * 4 * 16 * 4; }

// [tap 9][c2 2][kg 4][co 16][j 4] = W[16 c2 + co][ci = 4 kg + j][tap]; channels missing from cin_map get zero weights
void c16s2_pack_weights(const float *w, const float *b, int Ci, const int *cin_map, float *wpk, float *bpk)
{
    for (int t = 0; t < 9; ++t)
        for (int c2 = 0; c2 < 2; ++c2)
            for (int kg = 0; kg < 4; ++kg)
                for (int co = 0; co < 16; ++co)
                    for (int j = 0; j < 4; ++j) {
                        const int k = 4 * kg + j;
                        const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                        wpk[((((size_t)t * 2 + c2) * 4 + kg) * 16 + co) * 4 + j] = ci >= 0 ? w[((size_t)(16 * c2 + co) * Ci + ci) * 9 + t] : 0.f;
                    }
    for (int i = 0; i < 32; ++i) bpk[i] = b[i];
}

}  // namespace b2f
