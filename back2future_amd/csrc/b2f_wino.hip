// Winograd F(2x2, 3x3) convolution on the fp32 MFMA for gfx950 (MI355X): the stride-1
// nn.SpatialConvolution(Ci,Co,3,3,1,1,1,1) [+ LeakyReLU(0.2)] layers of /root/reference/models/
// pwc.lua:62 (second conv of every convUnit) and :78-82 (decoder layers), i.e. ~85 % of the
// FLOPs of computeFlow.  Same maths as b2f_conv.hip (cross-correlation, zero padding, bias first,
// LeakyReLU fused, chunk-planar in/out, up to two input K-segments) with 2.25x fewer MFMA MACs:
//
//   Y(2x2) = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        d = 4x4 input tile, g = 3x3 filter
//
// The 16 element-wise products become 16 independent GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci]
// U_xi[ci][co]  (xi = 4a + b), run on v_mfma_f32_32x32x2_f32 (exact fp32).  U = G g G^T is
// pre-computed on the host (wino_pack_weights); only additions happen on the device besides the
// MFMAs, so the result differs from the direct kernel by fp32 re-association only (~1e-6 rel.).
//
// Block = 512 threads (8 waves) -> 16 x 16 output pixels = 64 Winograd tiles, NT*32 output
// channels; wave w owns xi = 2w, 2w+1 for both 32-tile M halves and all NT N tiles
// (2 x 2 x NT accumulator tiles = 64*NT VGPRs).  K is walked in chunks of 8 input channels:
//   raw  [2 buf][2 k4][18 x 18] float4            input patch with halo
//   V    [2 buf][16 xi][2 k4][64 tiles] float4   transformed input  (A operand)
//   U    [16 xi][2 k4][NT*32 co] float4 per chunk: B operand, read straight from global (never in LDS)
// One barrier per chunk: while the MFMAs of chunk c run, the same waves transform chunk c+1
// (raw -> V: b128 LDS traffic only, 8 vector adds per thread, hidden under the 64-cycle MFMAs) and the
// global loads of raw(c+2) and U(c+1) are in flight.  After the last chunk the accumulators go
// through LDS ([xi][tile][co]) so that one thread holds all 16 xi of a (tile, co) pair, applies
// A^T . A, bias, LeakyReLU and stores the 2x2 outputs.
#include "b2f_internal.h"

#include <cstdlib>
#include <vector>

namespace b2f {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace wino {
constexpr int PW = 18;                 // patch width/height: 16 outputs + 2 halo
constexpr int RAW_P = 328;             // float4 per k4 plane (18*18 = 324 pixels, padded)
constexpr int RAW_F4 = 2 * RAW_P;      // float4 per raw buffer: [k4][pixel]
constexpr int V_F4 = 16 * 2 * 64;      // float4 per V buffer
constexpr int A_F4 = 2 * PW * PW;      // float4 items of one patch chunk (pixel, k4)
}  // namespace wino

template <int NT>
__global__ __launch_bounds__(512) void conv3x3_wino(const ConvLaunch p)
{
    using namespace wino;
    constexpr int NB = NT * 32;
    constexpr int U_F4 = 16 * 2 * NB;
    constexpr int A_PER_THREAD = (A_F4 + 511) / 512; // 2

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *Vb = reinterpret_cast<f32x4 *>(smem);                 // [2][V_F4]
    f32x4 *Rb = Vb + 2 * V_F4;                                    // [2][RAW_F4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, half = lane >> 5;

    const int tiles_x = (p.Wo + 15) / 16, tiles_y = (p.Ho + 15) / 16;
    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;
    const int nb = blockIdx.y;
    const int ox0 = tx_i * 16, oy0 = ty_i * 16;
    const int ix0 = ox0 - 1, iy0 = oy0 - 1;

    // ---- staging coordinates of the raw patch (fixed over chunks) ----
    int a_goff[A_PER_THREAD], a_pix[A_PER_THREAD];
    bool a_ok[A_PER_THREAD];
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
        const int idx = tid + i * 512;
        const int pix = idx >> 1;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = iy0 + py, gx = ix0 + px;
        a_ok[i] = (idx < A_F4) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        a_goff[i] = a_ok[i] ? (gy * p.W + gx) : 0;
        a_pix[i] = (idx < A_F4) ? pix : -1;
    }
    const int a_h = tid & 1;

    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(p.wpk) + (size_t)nb * nchunks * U_F4;

    f32x4 ra[A_PER_THREAD];
#define WINO_LOAD_RAW(c_)                                                                           \
    do {                                                                                            \
        const int c__ = (c_);                                                                       \
        const bool s1 = c__ >= p.seg[0].nchunks;                                                    \
        const float *base = s1 ? p.seg[1].ptr : p.seg[0].ptr;                                       \
        const long istr = s1 ? p.seg[1].img_stride : p.seg[0].img_stride;                           \
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                       \
        const int pstr = s1 ? p.seg[1].pix_stride : p.seg[0].pix_stride;                            \
        const int cc = s1 ? c__ - p.seg[0].nchunks : c__;                                           \
        const float *ib = base + (size_t)img * istr + (size_t)cc * cstr + a_h * 4;                  \
        _Pragma("unroll") for (int i = 0; i < A_PER_THREAD; ++i)                                    \
            ra[i] = *reinterpret_cast<const f32x4 *>(ib + (size_t)a_goff[i] * pstr);                \
    } while (0)
#define WINO_LOAD_U(dst_, c_)                                                                       \
    do {                                                                                            \
        const f32x4 *wb = wsrc + (size_t)(c_) * U_F4 + b_off;                                       \
        _Pragma("unroll") for (int x = 0; x < 2; ++x)                                               \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) dst_[x * NT + nt] = wb[x * 2 * NB + nt * 32]; \
    } while (0)
#define WINO_WRITE_RAW(buf_)                                                                        \
    do {                                                                                            \
        f32x4 *r = Rb + (buf_) * RAW_F4 + a_h * RAW_P;                                              \
        _Pragma("unroll") for (int i = 0; i < A_PER_THREAD; ++i) {                                  \
            const f32x4 v = a_ok[i] ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};                            \
            if ((i + 1) * 512 <= A_F4) r[a_pix[i]] = v;                                             \
            else if (a_pix[i] >= 0) r[a_pix[i]] = v;                                                \
        }                                                                                           \
    } while (0)

    // input transform of one chunk, V_xi = (B^T d B)[a][b] with xi = 4a + b, on float4 = 4 channels:
    // thread = (tile t = tid & 63, row a = (tid >> 6) & 3, k4 = tid >> 8); row a of B^T d needs two
    // rows of the 4x4 input tile, so a thread reads 8 float4 and writes the 4 float4 V[4a + b].
    // a and k4 are wave-uniform; all LDS traffic is b128 and the V writes are conflict-free.
    const int t_tile = tid & 63, t_a = (tid >> 6) & 3, t_k4 = tid >> 8;
    const int t_r0 = (t_a == 0) ? 0 : 1;                 // rows of d combined by B^T row a:
    const int t_r1 = (t_a == 3) ? 3 : 2;                 //   a=0: d0-d2, a=1: d1+d2, a=2: d2-d1, a=3: d1-d3
    const float t_s0 = (t_a == 2) ? -1.f : 1.f;
    const float t_s1 = (t_a == 1 || t_a == 2) ? 1.f : -1.f;
    const int t_src = t_k4 * RAW_P + (2 * (t_tile >> 3)) * PW + 2 * (t_tile & 7);
    const int t_dst = (t_a * 4 * 2 + t_k4) * 64 + t_tile;   // float4 index of V[xi = 4a][k4][t]; xi+1 -> +128
#define WINO_TRANSFORM(rbuf_, vbuf_)                                                                \
    do {                                                                                            \
        const f32x4 *r = Rb + (rbuf_) * RAW_F4 + t_src;                                             \
        f32x4 *v = Vb + (vbuf_) * V_F4 + t_dst;                                                     \
        f32x4 w[4];                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) w[j] = t_s0 * r[t_r0 * PW + j] + t_s1 * r[t_r1 * PW + j]; \
        v[0] = w[0] - w[2]; v[128] = w[1] + w[2]; v[256] = w[2] - w[1]; v[384] = w[1] - w[3];       \
    } while (0)

    f32x16 acc[2][2][NT];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][mt][nt][r] = 0.f;

    // B operands (transformed weights) never touch LDS: wave w only ever needs rows xi = 2w, 2w+1 of
    // the chunk's U slab, so each lane loads its own float4 (k4 = half, co = nt*32 + m) straight from
    // the packed global layout (512 contiguous bytes per half-wave), one chunk ahead.
    const int a_off = (2 * wave * 2 + half) * 64 + m;        // xi = 2*wave (+1: +128), k4 = half
    const int b_off = (2 * wave * 2 + half) * NB + m;
    f32x4 bcur[2 * NT], bnxt[2 * NT];
    const int ntv = min(NT, (p.cout - nb * NB + 31) / 32);   // N tiles of this block that hold real channels

    // ---- prologue ----
    {   // raw(0) and raw(1) in flight together (one memory round trip instead of two)
        f32x4 rb1[A_PER_THREAD];
        WINO_LOAD_RAW(min(1, nchunks - 1));
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) rb1[i] = ra[i];
        WINO_LOAD_RAW(0);
        WINO_LOAD_U(bcur, 0);
        WINO_WRITE_RAW(0);
#pragma unroll
        for (int i = 0; i < A_PER_THREAD; ++i) ra[i] = rb1[i];
        WINO_WRITE_RAW(1);
    }
    __syncthreads();
    WINO_TRANSFORM(0, 0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        // Branch-free body, hand-ordered with scheduling barriers so that every latency sits under
        // MFMAs of the same SIMD (its own or the other resident wave's):
        //   (1) issue the global loads of raw(c+2) / B(c+1) and ALL LDS reads of this iteration
        //       (A operands of both xi, the 8 raw float4 of the transform of chunk c+1);
        //   (2) 8*NT MFMAs of xi = 2w        -- the reads land meanwhile;
        //   (3) the transform's vector adds and its 4 V writes (VALU/LDS under the MFMA tail);
        //   (4) 8*NT MFMAs of xi = 2w+1;
        //   (5) raw(c+2) -> LDS, B registers roll over, barrier.
        // Past the last chunk the loads re-fetch the last chunk and the writes go to dead buffers.
        if (!(p.ablate & 2)) WINO_LOAD_RAW(min(c + 2, nchunks - 1));
        if (!(p.ablate & 4)) WINO_LOAD_U(bnxt, min(c + 1, nchunks - 1));
        const f32x4 *Vc = Vb + (c & 1) * V_F4 + a_off;
        f32x4 a0[2], a1[2], tr0[4], tr1[4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { a0[mt] = Vc[mt * 32]; a1[mt] = Vc[128 + mt * 32]; }
        {
            const f32x4 *r = Rb + ((c + 1) & 1) * RAW_F4 + t_src;
#pragma unroll
            for (int j = 0; j < 4; ++j) { tr0[j] = r[t_r0 * PW + j]; tr1[j] = r[t_r1 * PW + j]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.ablate & 8)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if (nt < ntv) acc[0][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt][j], bcur[nt][j], acc[0][mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.ablate & 1)) {
            f32x4 *v = Vb + ((c + 1) & 1) * V_F4 + t_dst;
            f32x4 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = t_s0 * tr0[j] + t_s1 * tr1[j];
            v[0] = w[0] - w[2]; v[128] = w[1] + w[2]; v[256] = w[2] - w[1]; v[384] = w[1] - w[3];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.ablate & 8)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if (nt < ntv) acc[1][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mt][j], bcur[NT + nt][j], acc[1][mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.ablate & 2)) WINO_WRITE_RAW(c & 1);
#pragma unroll
        for (int i = 0; i < 2 * NT; ++i) bcur[i] = bnxt[i];
        __syncthreads();
    }
#undef WINO_LOAD_RAW
#undef WINO_LOAD_U
#undef WINO_WRITE_RAW
#undef WINO_TRANSFORM

    // ---- output: accumulators -> LDS [xi][tile][co] -> A^T M A + bias (+ LeakyReLU) -> store ----
    float *X = reinterpret_cast<float *>(smem);   // 16 * 64 * 32 floats = 128 KB (V and U are dead)
    float *ob = p.out + (size_t)img * p.out_img_stride;
    // one (tile, 4 consecutive couts) item per thread and N tile: 16 ds_read_b128, vector adds,
    // four 16-byte stores (the 2x2 output pixels)
    const int o_t = tid >> 3, o_cq = tid & 7;
    const bool vec_ok = ((p.out_pix_stride | (int)p.out_chunk_stride) & 3) == 0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    X[((2 * wave + x) * 64 + t) * 32 + m] = acc[x][mt][nt][r];
                }
        __syncthreads();
        const int co0 = nb * NB + nt * 32 + 4 * o_cq;
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + co0);
        f32x4 mm[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bq = 0; bq < 4; ++bq)
                mm[a][bq] = *reinterpret_cast<const f32x4 *>(X + ((a * 4 + bq) * 64 + o_t) * 32 + 4 * o_cq);
        f32x4 sr[2][4];
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            sr[0][bq] = mm[0][bq] + mm[1][bq] + mm[2][bq];
            sr[1][bq] = mm[1][bq] - mm[2][bq] - mm[3][bq];
        }
        const int oy = oy0 + 2 * (o_t >> 3), ox = ox0 + 2 * (o_t & 7);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 y[2];
            y[0] = bias + (sr[i][0] + sr[i][1] + sr[i][2]);
            y[1] = bias + (sr[i][1] - sr[i][2] - sr[i][3]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = y[j];
                if (p.leaky) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                }
                if (oy + i < p.Ho && ox + j < p.Wo && co0 < p.cout) {
                    float *dst = ob + (size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)((oy + i) * p.Wo + ox + j) * p.out_pix_stride + (co0 & 7);
                    if (vec_ok && co0 + 3 < p.cout) {
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (co0 + e < p.cout)
                                ob[(size_t)((co0 + e) >> 3) * p.out_chunk_stride + (size_t)((oy + i) * p.Wo + ox + j) * p.out_pix_stride + ((co0 + e) & 7)] = v[e];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Second-generation kernel: same maths and packed weights, smaller blocks so that TWO independent
// blocks share a CU.  In the 512-thread kernel above every non-MFMA phase (prologue, the part of
// staging/transform that does not hide, the LDS output exchange) stalls the whole CU because the
// only resident block is in the same phase on all SIMDs; with two resident blocks that drift apart
// those phases of one block run under the MFMAs of the other.
//
// Block = 256 threads (4 waves) -> 8 x 16 output pixels = 32 Winograd tiles (exactly one 32-row
// MFMA M tile), NT*32 output channels; wave w owns row a = w of the transformed 4x4 tile
// (xi = 4w .. 4w+3, all NT N tiles: 4*NT accumulators = 64*NT VGPRs).  Per 8-channel chunk and
// wave: 4 A operands from LDS, 4*NT B operands straight from global in two halves (xi pair 0/1 is
// fetched while pair 2/3 multiplies and vice versa: 2 x 2*NT float4 of registers), 16*NT MFMAs.
//   raw  [2 buf][2 k4][10 x 18] float4,   V [2 buf][16 xi][2 k4][32 tiles] float4  (44.5 KB)
// Output: the first half of A^T M A (along b) happens in registers because a wave holds a whole
// row a; LDS only carries T[a][j][tile][co] (8 instead of 16 planes, 72 KB for NT = 2).
namespace wino2 {
constexpr int TH = 8, TW = 16;
constexpr int PH = TH + 2, PW = TW + 2;
constexpr int RAW_P = 256;             // float4 per k4 plane: 10*18 = 180 pixels + dummy slots, so that every one
                                       // of the 2 x 256 staging items has its own slot and the writes need no guard
constexpr int RAW_F4 = 2 * RAW_P;
constexpr int V_F4 = 16 * 2 * 32;
constexpr int A_F4 = 2 * PH * PW;      // 360 (pixel, k4) items per chunk
constexpr int lds_bytes(int nt)
{
    const int stage = 16 * (2 * V_F4 + 2 * RAW_F4);
    const int xch = 8 * 32 * (nt * 32 + 8) * 4;
    return stage > xch ? stage : xch;
}
}  // namespace wino2

// NTV = N tiles that hold real output channels (the last n-block of a layer whose cout is not a
// multiple of NT*32 runs the NTV < NT instantiation: same packed layout, fewer accumulators).
// Everything in the main loop is unconditional: a wave-uniform branch around a load or an MFMA
// makes the compiler's s_waitcnt insertion pessimistic (it then waits for the loads it has just
// issued), so profiling ablations are compile-time only (-DB2F_WINO2_ABLATE=bits).
#ifndef B2F_WINO2_ABLATE
#define B2F_WINO2_ABLATE 0
#endif
// Main-loop schedule: 0 = phases kept apart by scheduling barriers (loads + LDS reads | MFMAs of xi
// pair 0 | transform | MFMAs of xi pair 1 | staging), 1 = one MFMA then a few non-MFMA instructions
// (sched_group_barrier pipeline).  Measured: 0 is 16 % faster -- on gfx950 VALU instructions do not
// execute in the shadow of an fp32 MFMA (tools/mfma_overlap.hip), so interleaving buys nothing and
// shortens the distance between a load and its use.
#ifndef B2F_WINO2_SCHED
#define B2F_WINO2_SCHED 0
#endif
#if B2F_WINO2_SCHED == 0
#define W2_PHASE() __builtin_amdgcn_sched_barrier(0)
#else
#define W2_PHASE() do {} while (0)
#endif
template <int NT, int NTV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wino2(const ConvLaunch p)
{
    using namespace wino2;
    constexpr int ABL = B2F_WINO2_ABLATE;   // 1 no transform, 2 no raw staging, 4 no B loads, 8 no MFMAs
    constexpr int NB = NT * 32;
    constexpr int U_F4 = 16 * 2 * NB;
    constexpr int XS = NB + 8;             // floats per (plane, tile) row of the output exchange

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *Vb = reinterpret_cast<f32x4 *>(smem);                 // [2][V_F4]
    f32x4 *Rb = Vb + 2 * V_F4;                                    // [2][RAW_F4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, half = lane >> 5;

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;
    const int nb = blockIdx.y + p.nb0;
    const int ox0 = tx_i * TW, oy0 = ty_i * TH;
    const int ix0 = ox0 - 1, iy0 = oy0 - 1;

    // ---- staging coordinates of the raw patch (fixed over chunks): item idx = tid + 256 i ----
    int a_goff[2], a_pix[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        const int pix = idx >> 1;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = iy0 + py, gx = ix0 + px;
        a_ok[i] = (idx < A_F4) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        a_goff[i] = a_ok[i] ? (gy * p.W + gx) : 0;
        a_pix[i] = pix;
    }
    const int a_h = tid & 1;

    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(p.wpk) + (size_t)nb * nchunks * U_F4;

    f32x4 ra[2];
#define W2_LOAD_RAW(c_)                                                                             \
    do {                                                                                            \
        const int c__ = (c_);                                                                       \
        const bool s1 = c__ >= p.seg[0].nchunks;                                                    \
        const float *base = s1 ? p.seg[1].ptr : p.seg[0].ptr;                                       \
        const long istr = s1 ? p.seg[1].img_stride : p.seg[0].img_stride;                           \
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                       \
        const int pstr = s1 ? p.seg[1].pix_stride : p.seg[0].pix_stride;                            \
        const int cc = s1 ? c__ - p.seg[0].nchunks : c__;                                           \
        const float *ib = base + (size_t)img * istr + (size_t)cc * cstr + a_h * 4;                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                               \
            ra[i] = *reinterpret_cast<const f32x4 *>(ib + (size_t)a_goff[i] * pstr);                \
    } while (0)
    // B operands of xi pair q (xi = 4w + 2q, 4w + 2q + 1) of chunk c_: lane (m, half) loads U[xi][k4 = half][nt*32 + m]
#define W2_LOAD_U(dst_, c_, q_)                                                                     \
    do {                                                                                            \
        const f32x4 *wb = wsrc + (size_t)(c_) * U_F4 + b_off + (q_) * 4 * NB;                       \
        _Pragma("unroll") for (int x = 0; x < 2; ++x)                                               \
            _Pragma("unroll") for (int nt = 0; nt < NTV; ++nt) dst_[x * NTV + nt] = wb[x * 2 * NB + nt * 32]; \
    } while (0)
#define W2_WRITE_RAW(buf_)                                                                          \
    do {                                                                                            \
        f32x4 *r = Rb + (buf_) * RAW_F4 + a_h * RAW_P;                                              \
        r[a_pix[0]] = a_ok[0] ? ra[0] : f32x4{0.f, 0.f, 0.f, 0.f};                                  \
        r[a_pix[1]] = a_ok[1] ? ra[1] : f32x4{0.f, 0.f, 0.f, 0.f};                                  \
    } while (0)

    // input transform, thread = (tile t = tid & 31, row a = (tid >> 5) & 3, k4 = tid >> 7): reads the two
    // patch rows that B^T row a combines (8 float4), writes V[4a + b][k4][t], b = 0..3 (4 float4)
    const int t_tile = tid & 31, t_a = (tid >> 5) & 3, t_k4 = tid >> 7;
    const int t_r0 = (t_a == 0) ? 0 : 1;                 // a=0: d0-d2, a=1: d1+d2, a=2: d2-d1, a=3: d1-d3
    const int t_r1 = (t_a == 3) ? 3 : 2;
    const float t_s0 = (t_a == 2) ? -1.f : 1.f;
    const float t_s1 = (t_a == 1 || t_a == 2) ? 1.f : -1.f;
    const int t_src0 = t_k4 * RAW_P + (2 * (t_tile >> 3) + t_r0) * PW + 2 * (t_tile & 7);
    const int t_src1 = t_k4 * RAW_P + (2 * (t_tile >> 3) + t_r1) * PW + 2 * (t_tile & 7);
    const int t_dst = (t_a * 4 * 2 + t_k4) * 32 + t_tile;   // float4 index of V[xi = 4a][k4][t]; xi+1 -> +64

    f32x16 acc[4][NTV];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int nt = 0; nt < NTV; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][nt][r] = 0.f;

    const int a_off = (4 * wave * 2 + half) * 32 + m;        // V[xi = 4w][k4 = half][tile m]; xi+1 -> +64
    const int b_off = (4 * wave * 2 + half) * NB + m;        // U[xi = 4w][k4 = half][co m];   xi+1 -> +2*NB
    f32x4 b0[2 * NTV], b1[2 * NTV];

    // ---- prologue: raw(0), raw(1) and the first B pair in flight together ----
    {
        f32x4 rb1[2];
        W2_LOAD_RAW(min(1, nchunks - 1));
        rb1[0] = ra[0]; rb1[1] = ra[1];
        W2_LOAD_RAW(0);
        W2_LOAD_U(b0, 0, 0);
        W2_WRITE_RAW(0);
        ra[0] = rb1[0]; ra[1] = rb1[1];
        W2_WRITE_RAW(1);
    }
    __syncthreads();
    {
        const f32x4 *r = Rb;
        f32x4 *v = Vb + t_dst;
        f32x4 w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = t_s0 * r[t_src0 + j] + t_s1 * r[t_src1 + j];
        v[0] = w[0] - w[2]; v[64] = w[1] + w[2]; v[128] = w[2] - w[1]; v[192] = w[1] - w[3];
    }
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        // Branch-free, hand-ordered body (see the 512-thread kernel): loads and all LDS reads first,
        // MFMAs of xi pair 0, transform of chunk c+1, MFMAs of xi pair 1, staging of raw(c+2), barrier.
        if (!(ABL & 4)) W2_LOAD_U(b1, c, 1);
        if (!(ABL & 2)) W2_LOAD_RAW(min(c + 2, nchunks - 1));
        const f32x4 *Vc = Vb + (c & 1) * V_F4 + a_off;
        f32x4 av[4], tr0[4], tr1[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) av[x] = Vc[x * 64];
        {
            const f32x4 *r = Rb + ((c + 1) & 1) * RAW_F4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { tr0[j] = r[t_src0 + j]; tr1[j] = r[t_src1 + j]; }
        }
        W2_PHASE();
        if (!(ABL & 8)) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NTV; ++nt)
                        acc[x][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][j], b0[x * NTV + nt][j], acc[x][nt], 0, 0, 0);
        }
        W2_PHASE();
        if (!(ABL & 1)) {
            f32x4 *v = Vb + ((c + 1) & 1) * V_F4 + t_dst;
            f32x4 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = t_s0 * tr0[j] + t_s1 * tr1[j];
            v[0] = w[0] - w[2]; v[64] = w[1] + w[2]; v[128] = w[2] - w[1]; v[192] = w[1] - w[3];
        }
        if (!(ABL & 4)) W2_LOAD_U(b0, min(c + 1, nchunks - 1), 0);
        W2_PHASE();
        if (!(ABL & 8)) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NTV; ++nt)
                        acc[2 + x][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2 + x][j], b1[x * NTV + nt][j], acc[2 + x][nt], 0, 0, 0);
        }
        W2_PHASE();
        if (!(ABL & 2)) W2_WRITE_RAW(c & 1);
        // alternative issue order (B2F_WINO2_SCHED=1): one MFMA, then a few non-MFMA instructions, 32 times
        if (B2F_WINO2_SCHED == 1) {
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);          // A operands
#pragma unroll
            for (int i = 0; i < 16 * NTV; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                if (i < 2 * NTV + 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // B(q=1), raw loads
                else if (i >= 8 * NTV && i < 10 * NTV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // B(q=0) of c+1
                if (i < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // transform sources
                if (i >= 4) __builtin_amdgcn_sched_group_barrier(0x002, NTV == 2 ? 3 : 6, 0);      // transform / staging VALU
                if (i >= 16 * NTV - 8 && i < 16 * NTV - 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // V and raw writes
            }
        }
        __syncthreads();
    }
#undef W2_LOAD_RAW
#undef W2_LOAD_U
#undef W2_WRITE_RAW

    // ---- output: T[a][j] = sum_b A^T[j][b] M[a][b] in registers (wave = row a), exchange through
    // LDS [a][j][tile][co], then Y[i][j] = sum_a A^T[i][a] T[a][j] + bias (+ LeakyReLU) ----
    float *X = reinterpret_cast<float *>(smem);   // staging buffers are dead (barrier at loop end)
#pragma unroll
    for (int nt = 0; nt < NTV; ++nt) {
        const f32x16 t0 = acc[0][nt] + acc[1][nt] + acc[2][nt];
        const f32x16 t1 = acc[1][nt] - acc[2][nt] - acc[3][nt];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = (r & 3) + 8 * (r >> 2) + 4 * half;
            X[((wave * 2 + 0) * 32 + t) * XS + nt * 32 + m] = t0[r];
            X[((wave * 2 + 1) * 32 + t) * XS + nt * 32 + m] = t1[r];
        }
    }
    __syncthreads();
    float *ob = p.out + (size_t)img * p.out_img_stride;
    const bool vec_ok = ((p.out_pix_stride | (int)p.out_chunk_stride) & 3) == 0;
#pragma unroll
    for (int it = 0; it < NTV; ++it) {
        const int idx = tid + it * 256;
        const int o_cq = idx % (NTV * 8), o_t = idx / (NTV * 8);
        const int co0 = nb * NB + 4 * o_cq;
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + co0);
        f32x4 mm[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                mm[a][j] = *reinterpret_cast<const f32x4 *>(X + ((a * 2 + j) * 32 + o_t) * XS + 4 * o_cq);
        const int oy = oy0 + 2 * (o_t >> 3), ox = ox0 + 2 * (o_t & 7);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = bias + (i == 0 ? (mm[0][j] + mm[1][j] + mm[2][j]) : (mm[1][j] - mm[2][j] - mm[3][j]));
                if (p.leaky) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.2f * v[e];
                }
                if (oy + i < p.Ho && ox + j < p.Wo && co0 < p.cout) {
                    float *dst = ob + (size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)((oy + i) * p.Wo + ox + j) * p.out_pix_stride + (co0 & 7);
                    if (vec_ok && co0 + 3 < p.cout) {
                        *reinterpret_cast<f32x4 *>(dst) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (co0 + e < p.cout)
                                ob[(size_t)((co0 + e) >> 3) * p.out_chunk_stride + (size_t)((oy + i) * p.Wo + ox + j) * p.out_pix_stride + ((co0 + e) & 7)] = v[e];
                    }
                }
            }
        }
    }
}

template <int NT, int NTV>
static hipError_t launch_wino2_t(const ConvLaunch &p, int nb0, int nblk, hipStream_t s)
{
    using namespace wino2;
    constexpr int lds = lds_bytes(NT);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wino2<NT, NTV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    ConvLaunch q = p;
    q.nb0 = nb0;
    const int tiles = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH);
    dim3 grid((unsigned)(tiles * p.nimg), (unsigned)nblk);
    hipLaunchKernelGGL((conv3x3_wino2<NT, NTV>), grid, dim3(256), lds, s, q);
    return hipGetLastError();
}

template <int NT>
static hipError_t launch_wino_t(const ConvLaunch &p, hipStream_t s)
{
    using namespace wino;
    const size_t lds = sizeof(f32x4) * (2 * V_F4 + 2 * RAW_F4);
    const size_t lds_need = lds > 131072 ? lds : 131072;   // the output exchange needs 128 KB
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wino<NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((p.Wo + 15) / 16) * ((p.Ho + 15) / 16);
    dim3 grid((unsigned)(tiles * p.nimg), (unsigned)p.nblk);
    hipLaunchKernelGGL((conv3x3_wino<NT>), grid, dim3(512), lds_need, s, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino(const ConvLaunch &p_in, hipStream_t s)
{
    static const int ablate = getenv("B2F_WINO_ABLATE") ? atoi(getenv("B2F_WINO_ABLATE")) : 0;
    ConvLaunch p = p_in;
    p.ablate = ablate;
    if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo) return hipErrorInvalidValue;
    static const int gen = getenv("B2F_WINO_GEN") ? atoi(getenv("B2F_WINO_GEN")) : 2;
    if (gen == 2) {
        if (p.nt == 1) return launch_wino2_t<1, 1>(p, 0, p.nblk, s);
        if (p.nt != 2) return hipErrorInvalidValue;
        // n-blocks whose two N tiles both hold real channels, then the half-empty last one (cout = 96)
        const int nfull = p.cout / 64, part = (p.cout % 64) ? 1 : 0;
        const bool part_full = (p.cout % 64) > 32;
        hipError_t e = hipSuccess;
        if (nfull + (part_full ? 1 : 0) > 0) e = launch_wino2_t<2, 2>(p, 0, nfull + (part_full ? 1 : 0), s);
        if (e == hipSuccess && part && !part_full) e = launch_wino2_t<2, 1>(p, nfull, 1, s);
        return e;
    }
    if (p.nt == 1) return launch_wino_t<1>(p, s);
    if (p.nt == 2) return launch_wino_t<2>(p, s);
    return hipErrorInvalidValue;
}

void wino_choose_tiles(int cout, int *nt, int *nblk)
{
    if (cout <= 32) { *nt = 1; *nblk = 1; }
    else { *nt = 2; *nblk = (cout + 63) / 64; }
}

size_t wino_wpk_floats(int cin_chunks, int nt, int nblk)
{
    return (size_t)nblk * cin_chunks * 16 * 2 * nt * 32 * 4;
}

// U = G g G^T (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]) in double, rounded once to fp32;
// packed [nblk][chunk][xi 16][k4 2][NT*32 co][4 ci].
void wino_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks,
                       int nt, int nblk, float *wpk, float *bpk)
{
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int NB = nt * 32;
    std::vector<double> U((size_t)Co * Ci * 16);
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci) {
            const float *g = w + ((size_t)co * Ci + ci) * 9;
            double t[4][3];
            for (int a = 0; a < 4; ++a)
                for (int v = 0; v < 3; ++v) t[a][v] = G[a][0] * g[0 * 3 + v] + G[a][1] * g[1 * 3 + v] + G[a][2] * g[2 * 3 + v];
            for (int a = 0; a < 4; ++a)
                for (int bq = 0; bq < 4; ++bq)
                    U[((size_t)co * Ci + ci) * 16 + a * 4 + bq] = t[a][0] * G[bq][0] + t[a][1] * G[bq][1] + t[a][2] * G[bq][2];
        }
    for (int nbk = 0; nbk < nblk; ++nbk)
        for (int c = 0; c < cin_chunks; ++c)
            for (int xi = 0; xi < 16; ++xi)
                for (int h = 0; h < 2; ++h)
                    for (int nn = 0; nn < NB; ++nn)
                        for (int j = 0; j < 4; ++j) {
                            const int co = nbk * NB + nn;
                            const int k = c * kCK + h * 4 + j;
                            const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                            float v = 0.f;
                            if (co < Co && ci >= 0) v = (float)U[((size_t)co * Ci + ci) * 16 + xi];
                            wpk[((((((size_t)nbk * cin_chunks + c) * 16 + xi) * 2 + h) * NB + nn) * 4) + j] = v;
                        }
    for (int i = 0; i < nblk * NB; ++i) bpk[i] = i < Co ? b[i] : 0.f;
}

}  // namespace b2f
