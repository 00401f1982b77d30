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
// What shapes the kernel (measured on MI355X, tools/mfma_overlap.hip):
//   * an fp32 MFMA occupies its SIMD for 64 cycles and NOTHING of the VALU kind overlaps with it --
//     neither from the same wave nor from another wave of the SIMD; a wave64 VALU instruction costs
//     2.7-5 cycles and a global_load_dwordx4 ~16 cycles (up to ~45 when all SIMDs load at once) of
//     that same SIMD time.  LDS instructions and SALU are free.  So the loop is written to MINIMIZE
//     the VALU / VMEM instruction count per MFMA: loads use a scalar base + one 32-bit lane offset
//     (no per-chunk address arithmetic), zero padding is a masked LDS write into pre-zeroed slots
//     (no selects), the input transform is 8 float4 operations per thread (signs folded into a
//     per-lane fma multiplier and into the packed weights).
//   * accumulators (64 * NT VGPRs per wave) limit a SIMD to two waves, so memory latency is covered by
//     distance: B operands are fetched half a chunk ahead, the raw patch two chunks ahead.
//   * a runtime branch around a load or an MFMA makes s_waitcnt insertion pessimistic (the wave then
//     waits for loads it has just issued): the main loop is one basic block, N-tile count and
//     profiling ablations are template / preprocessor constants.
//
// Block = 256 threads (4 waves) -> 8 x 16 output pixels = 32 Winograd tiles (exactly one 32-row
// MFMA M tile), NT*32 output channels; two blocks per CU.  Wave w owns row a = w of the transformed
// 4x4 tile (xi = 4w .. 4w+3, all N tiles: 4*NTV accumulators).  Per 8-channel chunk and wave:
// 4 A operands from LDS, 4*NTV B operands straight from global in two halves (xi pair 0/1 is
// fetched while pair 2/3 multiplies and vice versa), 16*NTV MFMAs.
//   raw  [2 buf][2 k4][even | odd columns][10 rows][12] float4   input patch with halo (10 x 18 pixels), see RAW_* below
//   V    [2 buf][16 xi][2 k4][32 tiles] float4   transformed input (A operand)
// Output: the first half of A^T M A (along b) happens in registers because a wave holds a whole
// row a; LDS only carries T[a][j][tile][co] (8 instead of 16 planes, 72 KB for NT = 2).
#include "b2f_internal.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace b2f {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace wino {
constexpr int TH = 8, TW = 16;
constexpr int PH = TH + 2, PW = TW + 2;
// Raw patch in LDS (round 5: conflict-free; the row-major [k4][pixel] planes of rounds 1-4 gave 2-way conflicts on every staging
// write -- the two k4 planes of a pixel 4 096 bytes apart -- and on every transform read -- the second tile row of a 16-lane group
// 576 bytes behind the first --: 52 % of the kernel's LDS cycles, profiles/r04_sq_counters.json).  Even and odd patch columns
// apart: a tile's four columns 2 tx .. 2 tx + 3 are E[tx], O[tx], E[tx + 1], O[tx + 1], so the eight tiles of a tile row read 128
// contiguous bytes and the next tile row (two patch rows = 2 x 12 float4 = 384 bytes on) the other half of the banks; a 16-lane
// group of the staging write (8 consecutive pixels x 2 k4) lands in four 64-byte runs that tile 256 bytes (RAW_Q = 4, RAW_S = 8 mod 16).
constexpr int RAW_ROW = 12;            // float4 per half row (9 used)
constexpr int RAW_Q = 132;             // odd-column plane (>= 10 * 12, = 4 mod 16)
constexpr int RAW_S = 264;             // k4 = 1 plane (>= RAW_Q + 120, = 8 mod 16)
constexpr int RAW_F4 = 2 * RAW_S;
constexpr int V_F4 = 16 * 2 * 32;
constexpr int A_F4 = 2 * PH * PW;      // 360 (pixel, k4) items per chunk
constexpr int lds_bytes(int nt)
{
    const int stage = 16 * (2 * V_F4 + 2 * RAW_F4);
    const int xch = 8 * 32 * (nt * 32 + 8) * 4;
    return stage > xch ? stage : xch;
}
}  // namespace wino

// Profiling only (results are wrong): -DB2F_WINO_ABLATE=bits, 1 no transform, 2 no raw staging, 4 no B loads, 8 no MFMAs
#ifndef B2F_WINO_ABLATE
#define B2F_WINO_ABLATE 0
#endif

// Profiling only: -DB2F_WINO_TRACE=1 records clock64() at five points of every main-loop iteration of a
// few blocks (p.trace, set by the launcher when B2F_WINO_TRACE is in the environment).
#ifndef B2F_WINO_TRACE
#define B2F_WINO_TRACE 0
#endif
#if B2F_WINO_TRACE
#define W_T(k_) do { if (tr_on && lane == 0 && c < 32) tr_buf[(c * 5 + (k_))] = clock64(); } while (0)
#else
#define W_T(k_) do {} while (0)
#endif

// NTV = N tiles that hold real output channels (the last n-block of a layer whose cout is not a
// multiple of NT*32 runs the NTV < NT instantiation: same packed layout, fewer accumulators).
template <int NT, int NTV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wino(const ConvLaunch p)
{
    using namespace wino;
    constexpr int ABL = B2F_WINO_ABLATE;
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
    // 1-D grid, logical index = (image, tile row, tile column, n-block) with the n-block fastest, remapped so
    // that each XCD walks a contiguous range: the n-blocks of a tile and neighbouring tiles (which share the
    // raw patch resp. its halo) run on the same XCD at about the same time and find each other's lines in L2
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    // split launches (NT = 2, NTV = 1, p.nsplit = 1) enumerate 32-output N tiles: n-block = index >> 1, tile nt0 = index & 1
    const int nbx = bid % p.nblk;
    const int nt0 = (NT == 2 && NTV == 1) ? (nbx & p.nsplit) : 0;
    const int nb = (nbx >> ((NT == 2 && NTV == 1) ? p.nsplit : 0)) + p.nb0;
    bid /= p.nblk;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;
    const int ox0 = tx_i * TW, oy0 = ty_i * TH;
    const int ix0 = ox0 - 1, iy0 = oy0 - 1;

    // ---- staging of the raw patch: item idx = tid + 256 i -> (pixel idx >> 1, k4 = tid & 1).  Fixed over
    // chunks: a 32-bit byte offset inside the (image, chunk) plane per K segment; the plane base is scalar.
    // Items outside the image (zero padding) or past the patch load offset 0 and never write LDS:
    // their slots are zeroed once below.
    unsigned a_off0[2], a_off1[2];
    int a_slot[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256;
        const int pix = idx >> 1;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = iy0 + py, gx = ix0 + px;
        a_ok[i] = (idx < A_F4) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const unsigned g = a_ok[i] ? (unsigned)(gy * p.W + gx) : 0u;
        a_off0[i] = (g * (unsigned)p.seg[0].pix_stride + (tid & 1) * 4) * 4u;
        a_off1[i] = (g * (unsigned)p.seg[p.nseg > 1 ? 1 : 0].pix_stride + (tid & 1) * 4) * 4u;
        a_slot[i] = idx < A_F4 ? (tid & 1) * RAW_S + (px & 1) * RAW_Q + py * RAW_ROW + (px >> 1) : 0;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (!a_ok[i] && tid + i * 256 < A_F4) { Rb[a_slot[i]] = f32x4{0.f, 0.f, 0.f, 0.f}; Rb[RAW_F4 + a_slot[i]] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    const char *wsrc = reinterpret_cast<const char *>(p.wpk) + (size_t)nb * nchunks * U_F4 * 16;

    f32x4 ra[2];
    // buffer loads: scalar 128-bit resource (per K segment based at this image / the packed weights of this
    // n-block), scalar chunk offset, one 32-bit lane offset -- no per-load address arithmetic on the VALU
    const __amdgpu_buffer_rsrc_t r_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.seg[0].ptr + (size_t)img * p.seg[0].img_stride), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.seg[1].ptr + (size_t)img * p.seg[1].img_stride), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wsrc), 0, 0x7fffffff, 0x00020000);
#define W_LOAD_RAW(c_)                                                                              \
    do {                                                                                            \
        const int c__ = (c_);                                                                       \
        const bool s1 = c__ >= p.seg[0].nchunks;                                                    \
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                       \
        const int cc = s1 ? c__ - p.seg[0].nchunks : c__;                                           \
        const int so = (int)(cc * cstr * 4);                                                        \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                               \
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_rsrc1 : r_rsrc0,     \
                                                                                    (int)(s1 ? a_off1[i] : a_off0[i]), so, 0)); \
    } while (0)
    // B operands of xi pair q (xi = 4w + 2q, 4w + 2q + 1) of chunk c_: lane (m, half) loads U[xi][k4 = half][nt*32 + m]
#define W_LOAD_U(dst_, c_, q_)                                                                      \
    do {                                                                                            \
        const int wo = (c_) * (U_F4 * 16) + (q_) * (4 * NB * 16);                                   \
        _Pragma("unroll") for (int x = 0; x < 2; ++x)                                               \
            _Pragma("unroll") for (int nt = 0; nt < NTV; ++nt)                                      \
                dst_[x * NTV + nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(           \
                    w_rsrc, (int)b_off, wo + (x * 2 * NB + nt * 32) * 16, 0));                      \
    } while (0)
#define W_WRITE_RAW(buf_)                                                                           \
    do {                                                                                            \
        f32x4 *r = Rb + (buf_) * RAW_F4;                                                            \
        if (a_ok[0]) r[a_slot[0]] = ra[0];                                                          \
        if (a_ok[1]) r[a_slot[1]] = ra[1];                                                          \
    } while (0)

    // input transform, thread = (tile t = tid & 31, row a = (tid >> 5) & 3, k4 = tid >> 7): row a of B^T d
    // combines two patch rows, w = d[r0] + tau * d[r1]:
    //   a=0: d0 - d2,   a=1: d1 + d2,   a=2: d1 - d2 (= MINUS the textbook row; the packed weights of
    //   xi = 8..11 carry the other minus sign),   a=3: d1 - d3
    // then V[4a + b] = (w B)[b]:  w0 - w2,  w1 + w2,  w2 - w1,  w1 - w3.   8 b128 reads, 4 fma + 4 add/sub
    // on float4, 4 conflict-free b128 writes.
    const int t_tile = tid & 31, t_a = (tid >> 5) & 3, t_k4 = tid >> 7;
    const int t_r0 = (t_a == 0) ? 0 : 1;
    const int t_r1 = (t_a == 3) ? 3 : 2;
    const float t_tau = (t_a == 1) ? 1.f : -1.f;
    const f32x4 t_tau4 = {t_tau, t_tau, t_tau, t_tau};
    const int t_src0 = t_k4 * RAW_S + (2 * (t_tile >> 3) + t_r0) * RAW_ROW + (t_tile & 7);   // column 2 tx + j: + (j & 1) * RAW_Q + (j >> 1)
    const int t_src1 = t_k4 * RAW_S + (2 * (t_tile >> 3) + t_r1) * RAW_ROW + (t_tile & 7);
    const int t_dst = (t_a * 4 * 2 + t_k4) * 32 + t_tile;   // float4 index of V[xi = 4a][k4][t]; xi+1 -> +64
#define W_TRANSFORM(v_, s0_, s1_)                                                                   \
    do {                                                                                            \
        f32x4 w[4];                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) w[j] = __builtin_elementwise_fma(t_tau4, s1_[j], s0_[j]); \
        (v_)[0] = w[0] - w[2]; (v_)[64] = w[1] + w[2]; (v_)[128] = w[2] - w[1]; (v_)[192] = w[1] - w[3]; \
    } while (0)

    f32x16 acc[4][NTV];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int nt = 0; nt < NTV; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][nt][r] = 0.f;

    const int a_off = (4 * wave * 2 + half) * 32 + m;                    // V[xi = 4w][k4 = half][tile m]; xi+1 -> +64
    const unsigned b_off = ((4 * wave * 2 + half) * NB + nt0 * 32 + m) * 16u;       // bytes: U[xi = 4w][k4 = half][co m]; xi+1 -> +2*NB*16
    f32x4 b0[2 * NTV], b1[2 * NTV];

    // ---- prologue: raw(0), raw(1) and the first B pair in flight together ----
    {
        f32x4 rb1[2];
        W_LOAD_RAW(min(1, nchunks - 1));
        rb1[0] = ra[0]; rb1[1] = ra[1];
        W_LOAD_RAW(0);
        W_LOAD_U(b0, 0, 0);
        W_WRITE_RAW(0);
        ra[0] = rb1[0]; ra[1] = rb1[1];
        W_WRITE_RAW(1);
    }
    __syncthreads();
    {
        f32x4 s0[4], s1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s0[j] = Rb[t_src0 + (j & 1) * RAW_Q + (j >> 1)]; s1[j] = Rb[t_src1 + (j & 1) * RAW_Q + (j >> 1)]; }
        f32x4 *v = Vb + t_dst;
        W_TRANSFORM(v, s0, s1);
    }
    __syncthreads();

#if B2F_WINO_TRACE
    const int tr_slot = blockIdx.x == 1000 ? 0 : blockIdx.x == 1001 ? 1 : blockIdx.x == 5000 ? 2 : blockIdx.x == 5256 ? 3 : -1;
    const bool tr_on = p.trace && tr_slot >= 0;
    long long *tr_buf = p.trace + (tr_on ? (tr_slot * 4 + wave) * 160 : 0);
#endif
    for (int c = 0; c < nchunks; ++c) {
        W_T(0);
        // One basic block, phases kept apart by scheduling barriers:
        //   (1) issue the loads of B(c, xi pair 1) and raw(c+2) and ALL LDS reads of the iteration
        //   (2) MFMAs of xi pair 0 (B fetched during the previous iteration)
        //   (3) transform of chunk c+1 (raw -> V), then issue the loads of B(c+1, xi pair 0)
        //   (4) MFMAs of xi pair 1
        //   (5) raw(c+2) -> LDS, barrier.
        // Past the last chunk the loads re-fetch the last chunk and the writes go to dead buffers.
        if (!(ABL & 4)) W_LOAD_U(b1, c, 1);
        if (!(ABL & 2)) W_LOAD_RAW(min(c + 2, nchunks - 1));
        const f32x4 *Vc = Vb + (c & 1) * V_F4 + a_off;
        f32x4 av[4], tr0[4], tr1[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) av[x] = Vc[x * 64];
        {
            const f32x4 *r = Rb + ((c + 1) & 1) * RAW_F4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { tr0[j] = r[t_src0 + (j & 1) * RAW_Q + (j >> 1)]; tr1[j] = r[t_src1 + (j & 1) * RAW_Q + (j >> 1)]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 8)) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NTV; ++nt)
                        acc[x][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][j], b0[x * NTV + nt][j], acc[x][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        W_T(1);
        if (!(ABL & 1)) {
            f32x4 *v = Vb + ((c + 1) & 1) * V_F4 + t_dst;
            W_TRANSFORM(v, tr0, tr1);
        }
        if (!(ABL & 4)) W_LOAD_U(b0, min(c + 1, nchunks - 1), 0);
        __builtin_amdgcn_sched_barrier(0);
        W_T(2);
        if (!(ABL & 8)) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NTV; ++nt)
                        acc[2 + x][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2 + x][j], b1[x * NTV + nt][j], acc[2 + x][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        W_T(3);
        if (!(ABL & 2)) W_WRITE_RAW(c & 1);
        W_T(4);
        __syncthreads();
    }
#undef W_LOAD_RAW
#undef W_LOAD_U
#undef W_WRITE_RAW
#undef W_TRANSFORM

    // ---- output: T[a][j] = sum_b A^T[j][b] M[a][b] in registers (wave = row a), exchange through LDS
    // [a][j][tile][co], then Y[i][j] = sum_a A^T[i][a] T[a][j] + bias (+ LeakyReLU) ----
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
        const int co0 = nb * NB + nt0 * 32 + 4 * o_cq;
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
                if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);   // == v > 0 ? v : 0.2 v
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

// ======================================================================================================
// Eight-wave form of the one-N-tile instantiations (round 5): for launches of at most one block per CU -- a single triplet's
// coarse levels, where a launch is as long as ONE block's K loop and that loop is its 16 fp32 MFMAs per wave and chunk
// (1 024 cycles).  512 threads: wave w = (row a = w & 3 of the transformed tile, xi pair q = w >> 2) multiplies xi = 4 a + 2 q,
// + 1 only (8 MFMAs per chunk), the input transform is split the same way (thread = (tile, a, k4, pair): three of the four patch
// columns, two V values), staging is one item per thread.  All 16 M planes go through LDS (80 KB, one block per CU) and the
// output stage forms A^T M A from them with the four-wave kernel's operations in its order: same bits.
template <int NT>
__global__ __launch_bounds__(512) void conv3x3_wino8(const ConvLaunch p)
{
    using namespace wino;
    constexpr int NB = NT * 32;
    constexpr int U_F4 = 16 * 2 * NB;
    constexpr int XS8 = 32 + 8;              // floats per (plane, tile) row of the output exchange

    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *Vb = reinterpret_cast<f32x4 *>(smem);                 // [2][V_F4]
    f32x4 *Rb = Vb + 2 * V_F4;                                    // [2][RAW_F4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, half = lane >> 5;
    const int w_a = wave & 3, w_q = wave >> 2;

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nbx = bid % p.nblk;
    const int nt0 = NT == 2 ? (nbx & p.nsplit) : 0;
    const int nb = (nbx >> (NT == 2 ? p.nsplit : 0)) + p.nb0;
    bid /= p.nblk;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;
    const int ox0 = tx_i * TW, oy0 = ty_i * TH;
    const int ix0 = ox0 - 1, iy0 = oy0 - 1;

    // ---- staging: item tid -> (pixel tid >> 1, k4 = tid & 1) ----
    unsigned a_off0, a_off1;
    int a_slot;
    bool a_ok;
    {
        const int pix = tid >> 1;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = iy0 + py, gx = ix0 + px;
        a_ok = (tid < A_F4) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const unsigned g = a_ok ? (unsigned)(gy * p.W + gx) : 0u;
        a_off0 = (g * (unsigned)p.seg[0].pix_stride + (tid & 1) * 4) * 4u;
        a_off1 = (g * (unsigned)p.seg[p.nseg > 1 ? 1 : 0].pix_stride + (tid & 1) * 4) * 4u;
        a_slot = tid < A_F4 ? (tid & 1) * RAW_S + (px & 1) * RAW_Q + py * RAW_ROW + (px >> 1) : 0;
        if (!a_ok && tid < A_F4) { Rb[a_slot] = f32x4{0.f, 0.f, 0.f, 0.f}; Rb[RAW_F4 + a_slot] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    const char *wsrc = reinterpret_cast<const char *>(p.wpk) + (size_t)nb * nchunks * U_F4 * 16;
    const __amdgpu_buffer_rsrc_t r_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.seg[0].ptr + (size_t)img * p.seg[0].img_stride), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.seg[1].ptr + (size_t)img * p.seg[1].img_stride), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wsrc), 0, 0x7fffffff, 0x00020000);
    auto load_raw = [&](const int c) {
        const bool s1 = c >= p.seg[0].nchunks;
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;
        const int so = (int)((s1 ? c - p.seg[0].nchunks : c) * cstr * 4);
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_rsrc1 : r_rsrc0, (int)(s1 ? a_off1 : a_off0), so, 0));
    };
    // B operands of this wave's xi pair: lane (m, half) loads U[xi][k4 = half][nt0 * 32 + m]
    const unsigned b_off = (((4 * w_a + 2 * w_q) * 2 + half) * NB + nt0 * 32 + m) * 16u;
    auto load_u = [&](f32x4 (&dst)[2], const int c) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
            dst[x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)b_off, c * (U_F4 * 16) + x * 2 * NB * 16, 0));
    };
    // input transform: thread = (tile, row a, k4, pair bh): w = d[r0] + tau d[r1] of the columns bh .. bh + 2, then
    //   bh = 0: V[4a] = w0 - w2, V[4a + 1] = w1 + w2      bh = 1: V[4a + 2] = w2 - w1, V[4a + 3] = w1 - w3
    const int t_tile = tid & 31, t_a = (tid >> 5) & 3, t_k4 = (tid >> 7) & 1, t_bh = w_q;
    const int t_r0 = (t_a == 0) ? 0 : 1;
    const int t_r1 = (t_a == 3) ? 3 : 2;
    const float t_tau = (t_a == 1) ? 1.f : -1.f;
    const f32x4 t_tau4 = {t_tau, t_tau, t_tau, t_tau};
    const int t_src0 = t_k4 * RAW_S + (2 * (t_tile >> 3) + t_r0) * RAW_ROW + (t_tile & 7);
    const int t_src1 = t_k4 * RAW_S + (2 * (t_tile >> 3) + t_r1) * RAW_ROW + (t_tile & 7);
    const int t_dst = ((4 * t_a + 2 * t_bh) * 2 + t_k4) * 32 + t_tile;   // second value: + 64
    // patch column bh + i of the tile: slot + ((bh + i) & 1) * RAW_Q + ((bh + i) >> 1)
    const int t_c0 = t_bh ? RAW_Q : 0, t_c1 = t_bh ? 1 : RAW_Q, t_c2 = t_bh ? RAW_Q + 1 : 1;
    auto read_cols = [&](const f32x4 *r, f32x4 (&s0)[3], f32x4 (&s1)[3]) {
        s0[0] = r[t_src0 + t_c0]; s0[1] = r[t_src0 + t_c1]; s0[2] = r[t_src0 + t_c2];
        s1[0] = r[t_src1 + t_c0]; s1[1] = r[t_src1 + t_c1]; s1[2] = r[t_src1 + t_c2];
    };
    auto transform = [&](f32x4 *v, const f32x4 (&s0)[3], const f32x4 (&s1)[3]) {
        f32x4 w[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) w[j] = __builtin_elementwise_fma(t_tau4, s1[j], s0[j]);
        if (t_bh == 0) { v[0] = w[0] - w[2]; v[64] = w[1] + w[2]; }
        else { v[0] = w[1] - w[0]; v[64] = w[0] - w[2]; }
    };

    f32x16 acc[2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    const int a_off = ((4 * w_a + 2 * w_q) * 2 + half) * 32 + m;          // V[xi][k4 = half][tile m]; xi + 1 -> + 64
    f32x4 b0[2], b1[2];

    // ---- prologue: raw(0), raw(1), B(0) in flight together ----
    {
        const f32x4 r1 = load_raw(min(1, nchunks - 1));
        const f32x4 r0 = load_raw(0);
        load_u(b0, 0);
        if (a_ok) { Rb[a_slot] = r0; Rb[RAW_F4 + a_slot] = r1; }
    }
    __syncthreads();
    {
        f32x4 s0[3], s1[3];
        read_cols(Rb, s0, s1);
        transform(Vb + t_dst, s0, s1);
    }
    __syncthreads();

    // chunk c: [issue B(c + 1), raw(c + 2), the LDS reads] [8 MFMAs] [transform of chunk c + 1] [raw(c + 2) -> LDS] barrier
    auto iter = [&](const int c, const f32x4 (&bc)[2], f32x4 (&bn)[2]) __attribute__((always_inline)) {
        load_u(bn, min(c + 1, nchunks - 1));
        const f32x4 ra = load_raw(min(c + 2, nchunks - 1));
        const f32x4 *Vc = Vb + (c & 1) * V_F4 + a_off;
        f32x4 av[2], s0[3], s1[3];
        av[0] = Vc[0]; av[1] = Vc[64];
        read_cols(Rb + ((c + 1) & 1) * RAW_F4, s0, s1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][j], bc[x][j], acc[x], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        transform(Vb + ((c + 1) & 1) * V_F4 + t_dst, s0, s1);
        if (a_ok) Rb[(c & 1) * RAW_F4 + a_slot] = ra;
        __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 2) {
        iter(c, b0, b1);
        if (c + 1 < nchunks) iter(c + 1, b1, b0);
    }

    // ---- output: all 16 planes M[a][b] through LDS, then T[a][j] and Y as in the four-wave kernel ----
    float *X = reinterpret_cast<float *>(smem);   // staging buffers are dead (barrier at loop end)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = (r & 3) + 8 * (r >> 2) + 4 * half;
            X[((4 * w_a + 2 * w_q + x) * 32 + t) * XS8 + m] = acc[x][r];
        }
    __syncthreads();
    if (tid < 256) {
        float *ob = p.out + (size_t)img * p.out_img_stride;
        const bool vec_ok = ((p.out_pix_stride | (int)p.out_chunk_stride) & 3) == 0;
        const int o_cq = tid & 7, o_t = tid >> 3;
        const int co0 = nb * NB + nt0 * 32 + 4 * o_cq;
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + co0);
        f32x4 mm[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 mb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) mb[b] = *reinterpret_cast<const f32x4 *>(X + ((a * 4 + b) * 32 + o_t) * XS8 + 4 * o_cq);
            mm[a][0] = mb[0] + mb[1] + mb[2];
            mm[a][1] = mb[1] - mb[2] - mb[3];
        }
        const int oy = oy0 + 2 * (o_t >> 3), ox = ox0 + 2 * (o_t & 7);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = bias + (i == 0 ? (mm[0][j] + mm[1][j] + mm[2][j]) : (mm[1][j] - mm[2][j] - mm[3][j]));
                if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);
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

template <int NT>
static hipError_t launch_wino8_t(const ConvLaunch &p, int nb0, int nblk, hipStream_t s)
{
    using namespace wino;
    constexpr int stage = 16 * (2 * V_F4 + 2 * RAW_F4), xch = 16 * 32 * (32 + 8) * 4;
    constexpr int lds = stage > xch ? stage : xch;
    static bool attr_done_dev[64] = {false};
    bool &attr_done = attr_done_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wino8<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    ConvLaunch q = p;
    q.nb0 = nb0;
    q.nsplit = NT == 2 ? p.nsplit : 0;
    q.trace = nullptr;
    q.nblk = nblk;
    const int tiles = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH);
    hipLaunchKernelGGL((conv3x3_wino8<NT>), dim3((unsigned)(tiles * p.nimg * nblk)), dim3(512), lds, s, q);
    return hipGetLastError();
}

template <int NT, int NTV>
static hipError_t launch_wino_t(const ConvLaunch &p, int nb0, int nblk, hipStream_t s)
{
    using namespace wino;
    constexpr int lds = lds_bytes(NT);
    static bool attr_done_dev[64] = {false};
    bool &attr_done = attr_done_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wino<NT, NTV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    ConvLaunch q = p;
    q.nb0 = nb0;
    q.nsplit = (NT == 2 && NTV == 1) ? p.nsplit : 0;
    q.trace = nullptr;
#if B2F_WINO_TRACE
    static long long *trace_dev = nullptr;
    static int traced = 0;
    const bool do_trace = getenv("B2F_WINO_TRACE") && traced < 2 && NT == 2 && NTV == 2 && p.seg[0].nchunks == 16 && p.nseg == 1 &&
                          p.H * p.W >= 256 * 480;
    if (do_trace) {
        if (!trace_dev) hipMalloc(&trace_dev, 16 * 160 * sizeof(long long));
        hipMemsetAsync(trace_dev, 0, 16 * 160 * sizeof(long long), s);
        q.trace = trace_dev;
    }
#endif
    const int tiles = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH);
    q.nblk = nblk;                          // n-blocks of THIS launch (the kernel decodes them from the 1-D grid)
    dim3 grid((unsigned)(tiles * p.nimg * nblk));
    hipLaunchKernelGGL((conv3x3_wino<NT, NTV>), grid, dim3(256), lds, s, q);
#if B2F_WINO_TRACE
    if (do_trace) {
        ++traced;
        std::vector<long long> h(16 * 160);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), trace_dev, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        for (int b = 0; b < 4; ++b)
            for (int w = 0; w < 4; ++w) {
                const long long *t = h.data() + (b * 4 + w) * 160;
                fprintf(stderr, "wino trace block-slot %d wave %d (cycles since first stamp; top | mfma0 done | transform+Bissue done | mfma1 done | raw written):\n", b, w);
                for (int c = 0; c < 16; ++c)
                    fprintf(stderr, "  c=%2d  %7lld %7lld %7lld %7lld %7lld\n", c, t[c * 5] - t[0], t[c * 5 + 1] - t[0], t[c * 5 + 2] - t[0],
                            t[c * 5 + 3] - t[0], t[c * 5 + 4] - t[0]);
            }
    }
#endif
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino(const ConvLaunch &p, hipStream_t s)
{
    if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo) return hipErrorInvalidValue;
    // 32-bit byte offsets inside one (image, chunk) plane
    for (int i = 0; i < p.nseg; ++i)
        if ((double)p.H * p.W * p.seg[i].pix_stride * 4.0 >= 4294967296.0) return hipErrorInvalidValue;
    // one-N-tile launches of at most one block per CU: the eight-wave form (same bits; p.w8 = 0 switches it off)
    const long tiles8 = (long)((p.Wo + wino::TW - 1) / wino::TW) * ((p.Ho + wino::TH - 1) / wino::TH) * p.nimg;
    const bool w8 = p.w8 != 0;
    const long n_cu = device_cu_count();
    if (p.nt == 1) return (w8 && tiles8 * p.nblk <= n_cu) ? launch_wino8_t<1>(p, 0, p.nblk, s) : launch_wino_t<1, 1>(p, 0, p.nblk, s);
    if (p.nt != 2) return hipErrorInvalidValue;
    if (p.nsplit) {   // one block per 32 outputs
        const int n32 = (p.cout + 31) / 32;
        return (w8 && tiles8 * n32 <= n_cu) ? launch_wino8_t<2>(p, 0, n32, s) : launch_wino_t<2, 1>(p, 0, n32, s);
    }
    // n-blocks whose two N tiles both hold real channels, then the half-empty last one (cout = 96)
    const int nfull = p.cout / 64, part = (p.cout % 64) ? 1 : 0;
    const bool part_full = (p.cout % 64) > 32;
    hipError_t e = hipSuccess;
    if (nfull + (part_full ? 1 : 0) > 0) e = launch_wino_t<2, 2>(p, 0, nfull + (part_full ? 1 : 0), s);
    if (e == hipSuccess && part && !part_full) e = (w8 && tiles8 <= n_cu) ? launch_wino8_t<2>(p, nfull, 1, s) : launch_wino_t<2, 1>(p, nfull, 1, s);
    return e;
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
// packed [nblk][chunk][xi 16][k4 2][NT*32 co][4 ci].  Rows xi = 8..11 (a = 2) are stored NEGATED: the
// kernel's input transform produces -V for that row (one fma instead of mul + mul + add), and
// (-V)(-U) = V U exactly.
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
                            if ((xi >> 2) == 2) v = -v;
                            wpk[((((((size_t)nbk * cin_chunks + c) * 16 + xi) * 2 + h) * NB + nn) * 4) + j] = v;
                        }
    for (int i = 0; i < nblk * NB; ++i) bpk[i] = i < Co ? b[i] : 0.f;
}

}  // namespace b2f
