// Winograd F(2x2, 3x3) convolution on the BF16 matrix pipe with exactly split fp32 operands, for gfx950 (MI355X): the wide
// stride-1 nn.SpatialConvolution(Ci,Co,3,3,1,1,1,1) [+ LeakyReLU(0.2)] layers of /root/reference/models/pwc.lua:62,78-82 whose
// output channels come in blocks of 64.  Same interface as the other conv kernels (chunk-planar in / out, up to two input K
// segments, bias + LeakyReLU fused).
//
//   Y(2x2) = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A         d = 4x4 input tile, g = 3x3 filter
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// Why this form (profiles/r04_wino4s_notes.txt, r04_wino4_notes.txt): the fp32 MFMA holds a SIMD's matrix pipe AND its VALU; the bf16
// MFMA is 16x faster per MAC and leaves the VALU free, and with every fp32 operand split exactly into three bf16 terms (x = xh + xm
// + xl, six of the nine term products kept: fp32-level accuracy, b2f_wino4s.hip) a GEMM step costs 3 bf16 MFMAs instead of 4 fp32
// ones.  On F(4x4) that did not pay: its 36 accumulator planes cap a block at 32 tiles x 64 outputs with no register tiling, so per
// MAC it needs more operand bytes, more split work (every V element split by two waves) and more LDS traffic than two in-order waves
// per SIMD can overlap with the MFMAs.  F(2x2) has 16 planes:
//   * block = 64 tiles (16 x 16 output pixels) x 64 outputs; a wave owns xi planes for BOTH M tiles and BOTH N tiles: every A window
//     feeds two N tiles and every B window two M tiles (half the operand traffic per MFMA), every V element is read and split by
//     exactly one wave; with ONE xi per wave (16 waves = four per SIMD) that is 4 accumulators = 64 of a wave's 128 registers;
//   * the input transform has no multiplications (32 additions per tile and channel), the output transform reads each product once;
//   * 4 MACs per output instead of 2.25 -- 1.8x the MFMAs of F(4x4), which the bf16 pipe has to spare: per 8-channel chunk a SIMD
//     issues 48 MFMAs (1 536 cycles) against ~300 VALU instructions, 49 KB of B operands per CU (32 B/clk) and ~110 KB of LDS traffic;
//   * rounding: F(2x2)'s transforms are additions and halvings, its error is ~10x below F(4x4)'s.
// K is walked in chunks of 8 input channels; a lane (tile or co = lane & 31, k4 = lane >> 5) holds the 4 channels of its k4 group as
// bf16 pairs in a window of six dwords  A = [Vm01 Vm23 | Vh01 Vh23 | Vl01 Vl23],  B = [Um01 Um23 | Uh01 Uh23 | Ul01 Ul23]  and the
// three MFMAs of a product are  A[0:3] B[0:3] = Vm Um + Vh Uh,  A[2:5] B[0:3] = Vh Um + Vl Uh,  A[0:3] B[2:5] = Vm Uh + Vh Ul.
// LDS (154 112 B):  raw [2 buf][k4][pair][18 rows][20] f32x2 (columns permuted even | odd)  |  V [2 buf][xi 16][k4][pair][64 tiles]
// f32x2 (fp32; split by the wave that reads it)  |  X [xi 16][16 tiles][64 co] fp32, the exchange buffer of the output stage.
#include "b2f_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace b2f {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace wino2s {
constexpr int TH = 16, TW = 16;             // output pixels per block = 8 x 8 tiles of 2 x 2
constexpr int PH = TH + 2, PW = TW + 2;     // 18 x 18 input patch
constexpr int RWP = 20;                     // f32x2 per patch row of a pair plane; RWP = 4 (mod 16): the four tile rows of a 32-lane
                                            // ds_read_b64 group land 64 B apart in bank space
constexpr int PL2 = PH * RWP;               // 360 f32x2 per (k4, pair) plane
constexpr int RAW_F2 = 4 * PL2;             // f32x2 per raw buffer
constexpr int V_F2 = 16 * 2 * 2 * 64;       // f32x2 per V buffer
constexpr int X_FLOATS = 16 * 16 * 64;
constexpr int LDS_BYTES = 8 * (2 * RAW_F2 + 2 * V_F2) + 4 * X_FLOATS;   // 23 040 + 65 536 + 65 536 = 154 112
constexpr int U4_BYTES = 16 * 2 * 64 * 16;  // (Um Um Uh Uh) plane of one (n-block, chunk)
constexpr int UC_BYTES = 2 * U4_BYTES;      // + the (Uh Uh Ul Ul) plane: Uh is stored twice so that both B windows arrive as whole
                                            // register quads (a wave is bound by the number of instructions it issues, 5.5 cycles
                                            // each: two copies per window cost more than 8 more bytes per lane)
constexpr int NITEM = 2 * PH * PW;          // 648 (pixel, k4) staging items per chunk
__device__ __host__ constexpr int colperm(int p) { return (p & 1) * 9 + (p >> 1); }
}  // namespace wino2s

__device__ __forceinline__ unsigned w2s_pk(float a, float b)
{
    // one v_cvt_pk_bf16_f32; NOT inline asm: the compiler must see the instruction to keep the VALU-write -> MFMA-read
    // wait states (an asm statement two instructions ahead of the MFMA that read its result gave garbage)
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));
}
// fp32 quad -> window [m01 m23 | h01 h23 | l01 l23] of bf16 pairs (round to nearest even; x = h + m + l exactly); scalar fp32
// subtractions on purpose (packed fp32 ops do not overlap the bf16 MFMAs: tools/mfma_bf16_chain.hip; -fno-slp-vectorize)
__device__ __forceinline__ void w2s_split(const f32x4 v, unsigned (&w)[6])
{
    const unsigned h01 = w2s_pk(v[0], v[1]), h23 = w2s_pk(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, h01 << 16), r1 = v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h23 << 16), r3 = v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u);
    const unsigned m01 = w2s_pk(r0, r1), m23 = w2s_pk(r2, r3);
    const float l0 = r0 - __builtin_bit_cast(float, m01 << 16), l1 = r1 - __builtin_bit_cast(float, m01 & 0xffff0000u);
    const float l2 = r2 - __builtin_bit_cast(float, m23 << 16), l3 = r3 - __builtin_bit_cast(float, m23 & 0xffff0000u);
    w[0] = m01; w[1] = m23; w[2] = h01; w[3] = h23;
    w[4] = w2s_pk(l0, l1);
    w[5] = w2s_pk(l2, l3);
}

#ifndef B2F_W2S_ABLATE
#define B2F_W2S_ABLATE 0     // profiling only (wrong results): 1 no input transform, 2 no raw staging, 4 no B loads, 8 no MFMAs, 16 no split
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// 1 024 threads = FOUR waves per SIMD: wave w owns ONE xi (= w) for both M tiles and both N tiles -- 4 accumulators = 64 of its 128
// registers.  The 8-wave forms of this kernel (two xi per wave, two waves per SIMD, with and without a deep software pipeline) ran
// at the fp32 kernel's speed with the matrix pipe 37 % busy: an in-order wave exposes every latency its one partner does not cover,
// and the younger wave of a SIMD loses every arbitration (profiles/r04_wino2s_notes.txt (3), (4)).  With four waves the hardware
// does the overlapping: per chunk a wave reads its A operand, splits it, issues 12 MFMAs and a 16th of the transform, plainly in
// that order.
//   transform thread = (tile = lane, k4, channel pair, row a of the 4 x 4) = 64 x 16 waves; staging: one (pixel, k4) item per thread
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void conv3x3_wino2s(const ConvLaunch p)
{
    using namespace wino2s;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x2 *R2 = reinterpret_cast<f32x2 *>(smem);                 // [2][RAW_F2]
    f32x2 *V2 = R2 + 2 * RAW_F2;                                  // [2][V_F2]
    float *X = reinterpret_cast<float *>(V2 + 2 * V_F2);          // [16][16][64]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..15 = xi
    const int m = lane & 31, half = lane >> 5;

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = bid % p.nblk + p.nb0;
    bid /= p.nblk;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;
    const int ox0 = tx_i * TW, oy0 = ty_i * TH;
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);

    // ---- staging: thread tid < 648 = (pixel tid >> 1 of the 18 x 18 patch, k4 = tid & 1) ----
    unsigned s_off;
    int s_slot;
    const bool s_on = tid < NITEM;
    {
        const int pix = min(tid, NITEM - 1) >> 1;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
        const bool ok = s_on && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        s_off = ok ? ((unsigned)(gy * p.W + gx) * (unsigned)p.seg[0].pix_stride + (tid & 1) * 4) * 4u : 0xfffffff0u;
        s_slot = (tid & 1) * 2 * PL2 + py * RWP + colperm(px);
    }
    const __amdgpu_buffer_rsrc_t r_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.seg[0].ptr + (size_t)img * p.seg[0].img_stride), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.seg[1].ptr + (size_t)img * p.seg[1].img_stride), 0, 0x7fffffff, 0x00020000);
    f32x4 sr[2];
#define W2S_LOAD_RAW(c_, k_)                                                                           \
    do {                                                                                            \
        const int c__ = (c_);                                                                       \
        const bool s1 = c__ >= p.seg[0].nchunks;                                                    \
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                       \
        const int cc = s1 ? c__ - p.seg[0].nchunks : c__;                                           \
        if (!(B2F_W2S_ABLATE & 2))                                                                  \
            sr[k_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_rsrc1 : r_rsrc0, (int)s_off, (int)(cc * cstr * 4), 0)); \
    } while (0)
#define W2S_WRITE_RAW(buf_, k_)                                                                       \
    do {                                                                                            \
        f32x2 *r__ = R2 + (buf_) * RAW_F2 + s_slot;                                                 \
        if (s_on && !(B2F_W2S_ABLATE & 2)) {                                                        \
            r__[0] = __builtin_shufflevector(sr[k_], sr[k_], 0, 1);                                 \
            r__[PL2] = __builtin_shufflevector(sr[k_], sr[k_], 2, 3);                               \
        }                                                                                           \
    } while (0)

    // ---- input transform: lane = tile (ty = lane >> 3, tx = lane & 7), wave = (k4 = wave & 1, pair = (wave >> 1) & 1, row a = wave >> 2):
    //   B^T row a of the patch rows:  a = 0: d0 - d2,  a = 1: d1 + d2,  a = 2: d2 - d1,  a = 3: d1 - d3   = x1 + sg x2 of two patch rows
    const int t_k4 = wave & 1, t_pair = (wave >> 1) & 1, t_a = wave >> 2;
    const int t_r1 = t_a == 0 ? 0 : t_a == 2 ? 2 : 1, t_r2 = t_a == 0 ? 2 : t_a == 1 ? 2 : t_a == 2 ? 1 : 3;
    const float t_sg = t_a == 1 ? 1.f : -1.f;
    const int t_base = (t_k4 * 2 + t_pair) * PL2 + (2 * (lane >> 3) + t_r1) * RWP + (lane & 7);
    const int t_row2 = (t_r2 - t_r1) * RWP;
    const int t_dst = (((4 * t_a) * 2 + t_k4) * 2 + t_pair) * 64 + lane;                       // V[xi = 4 a][k4][pair][tile]; xi + 1 -> + 256
#define W2S_TRANSFORM(rbuf_, vbuf_)                                                                 \
    do {                                                                                            \
        if (!(B2F_W2S_ABLATE & 1)) {                                                                \
            const f32x2 *rp__ = R2 + (rbuf_) * RAW_F2 + t_base;                                     \
            float rx__[4], ry__[4];     /* scalar on purpose: packed fp32 ops hold the VALU ~7 cycles and do not overlap the MFMAs */ \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
                const int cj = (j & 1) * 9 + (j >> 1);                                              \
                const f32x2 p1 = rp__[cj], p2 = rp__[t_row2 + cj];                                  \
                rx__[j] = __builtin_fmaf(t_sg, p2[0], p1[0]);                                       \
                ry__[j] = __builtin_fmaf(t_sg, p2[1], p1[1]);                                       \
            }                                                                                       \
            f32x2 *v__ = V2 + (vbuf_) * V_F2 + t_dst;                                               \
            v__[0] = f32x2{rx__[0] - rx__[2], ry__[0] - ry__[2]};                                   \
            v__[256] = f32x2{rx__[1] + rx__[2], ry__[1] + ry__[2]};                                 \
            v__[512] = f32x2{rx__[2] - rx__[1], ry__[2] - ry__[1]};                                 \
            v__[768] = f32x2{rx__[1] - rx__[3], ry__[1] - ry__[3]};                                 \
        }                                                                                           \
    } while (0)

    // ---- GEMM side: wave = xi, M tiles mt = 0, 1 (tiles 32 mt + m), N tiles nt = 0, 1 ----
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const int a_off = ((wave * 2 + half) * 2) * 64 + m;          // f32x2 index of V[xi][k4 = half][pair 0][tile m]; pair 1 -> + 64, mt -> + 32
    const unsigned b_off = ((wave * 2 + half) * 64 + m) * 16u;   // bytes: (Um Um Uh Uh) of [xi][k4][co m]; nt -> + 512; (Uh Uh Ul Ul): + U4_BYTES
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wpk_split2) + (size_t)nb * nchunks * UC_BYTES), 0, 0x7fffffff, 0x00020000);
    f32x4 av[2];
    u32x4 bq[2], bl[2];
#define W2S_A_READ(vbuf_)                                                                           \
    do {                                                                                            \
        const f32x2 *q__ = V2 + (vbuf_) * V_F2 + a_off;                                             \
        av[0] = __builtin_shufflevector(q__[0], q__[64], 0, 1, 2, 3);                               \
        av[1] = __builtin_shufflevector(q__[32], q__[96], 0, 1, 2, 3);                              \
    } while (0)
#define W2S_LOAD_UQ(c_)                                                                             \
    do {                                                                                            \
        if (!(B2F_W2S_ABLATE & 4)) {                                                                \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                        \
                bq[nt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off + nt * 512), (c_) * UC_BYTES, 0)); \
        }                                                                                           \
    } while (0)
#define W2S_LOAD_UL(c_)                                                                             \
    do {                                                                                            \
        if (!(B2F_W2S_ABLATE & 4)) {                                                                \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                        \
                bl[nt] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)(b_off + U4_BYTES + nt * 512), (c_) * UC_BYTES, 0)); \
        }                                                                                           \
    } while (0)
#define W2S_MF(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)

    // ---- prologue: raw(0), raw(1) -> LDS; Tr(0) -> V[0]; raw(2), raw(3) in flight; B and A of chunk 0 ----
    W2S_LOAD_RAW(0, 0);
    W2S_LOAD_RAW(min(1, nchunks - 1), 1);
    W2S_WRITE_RAW(0, 0);
    W2S_LOAD_RAW(min(2, nchunks - 1), 0);
    W2S_WRITE_RAW(1, 1);
    W2S_LOAD_RAW(min(3, nchunks - 1), 1);
    W2S_LOAD_UQ(0);
    W2S_LOAD_UL(0);
    __syncthreads();
    W2S_TRANSFORM(0, 0);
    __syncthreads();
    W2S_A_READ(0);

    // Iteration c (unrolled by two: buffer parities and the staging register set are compile-time): multiply V[c & 1]; transform
    // raw[(c + 1) & 1] -> V[(c + 1) & 1]; raw(c + 2), requested two iterations ago, -> raw[c & 1] (consumed by Tr(c) in iteration
    // c - 1); request raw(c + 4) (HBM latency under load is longer than one iteration); ONE barrier; then the A operand of c + 1.
    auto iteration = [&](auto pcv, const int c) __attribute__((always_inline)) {
        constexpr int PC_ = decltype(pcv)::value;
        u32x4 amh[2], ahl[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            unsigned w[6];
            if (!(B2F_W2S_ABLATE & 16)) w2s_split(av[q], w);
            else {
#pragma unroll
                for (int k = 0; k < 6; ++k) w[k] = __builtin_bit_cast(unsigned, av[q][k & 3]);
            }
            amh[q] = u32x4{w[0], w[1], w[2], w[3]};
            ahl[q] = u32x4{w[2], w[3], w[4], w[5]};
        }
        // the B windows of chunk c + 1 are requested as soon as their registers are free (after MFMA 8 and 12): their latency
        // runs under the rest of this iteration instead of after the barrier, where all 16 waves would wait for it together
        const int cn = min(c + 1, nchunks - 1);
        if (!(B2F_W2S_ABLATE & 8)) {
            W2S_MF(acc[0][0], amh[0], bq[0]);
            W2S_MF(acc[0][1], amh[0], bq[1]);
            W2S_MF(acc[1][0], amh[1], bq[0]);
            W2S_MF(acc[1][1], amh[1], bq[1]);
            W2S_MF(acc[0][0], ahl[0], bq[0]);
            W2S_MF(acc[0][1], ahl[0], bq[1]);
            W2S_MF(acc[1][0], ahl[1], bq[0]);
            W2S_MF(acc[1][1], ahl[1], bq[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        W2S_LOAD_UQ(cn);
        __builtin_amdgcn_sched_barrier(0);
        if (!(B2F_W2S_ABLATE & 8)) {
            W2S_MF(acc[0][0], amh[0], bl[0]);
            W2S_MF(acc[0][1], amh[0], bl[1]);
            W2S_MF(acc[1][0], amh[1], bl[0]);
            W2S_MF(acc[1][1], amh[1], bl[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        W2S_LOAD_UL(cn);
        W2S_TRANSFORM(PC_ ^ 1, PC_ ^ 1);
        W2S_WRITE_RAW(PC_, PC_);
        W2S_LOAD_RAW(min(c + 4, nchunks - 1), PC_);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        W2S_A_READ(PC_ ^ 1);
    };
    for (int c = 0; c < nchunks; c += 2) {
        iteration(std::integral_constant<int, 0>{}, c);
        if (c + 1 < nchunks) iteration(std::integral_constant<int, 1>{}, c + 1);
    }

    // ---- output: four passes (M tile mt, row half qh) = 16 tiles (two tile rows: 4 x 16 output pixels) x 64 outputs x 16 xi through
    // X[xi][tile 16][co ^ (tile column << 3)] (a dump instruction -- 32 consecutive co of a tile per lane half, tile columns c and c + 4
    // -- and a 16-lane group of the reads -- two 4-channel groups x eight tile columns -- both touch every bank once).  Dump: every
    // wave writes its xi plane; then item = (tile 16, 4 channels, output row i of the 2 x 2) on threads 0..511: the 12 products of rows
    // a = i .. i + 2, A^T M A, bias, LeakyReLU, two 16-byte stores; a wave's two stores together cover four 512-byte row segments.
    float *ob = p.out + (size_t)img * p.out_img_stride;
    const int o_w = wave & 7;
    const int o_cq = 2 * o_w + (lane & 1), o_tx = (lane >> 1) & 7, o_tr = (lane >> 4) & 1, o_i = lane >> 5;
    const int o_tl = o_tx + 8 * o_tr;
    const int co0 = nb * 64 + 4 * o_cq;
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + co0);
    const bool col_ok = co0 < p.cout;
    const float *xa = X + o_tl * 64 + ((4 * o_cq) ^ (o_tx << 3));
    const float sgn = o_i ? -1.f : 1.f;                                        // A^T row i: (1 1 1 0) or (0 1 -1 -1) over a = i .. i + 2
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int mt = pass >> 1, qh = pass & 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // X free (previous pass read)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int tl = (r8 & 3) + 4 * half + 8 * (r8 >> 2);
                X[(wave * 16 + tl) * 64 + ((nt * 32 + m) ^ ((tl & 7) << 3))] = acc[mt][nt][8 * qh + r8];
            }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (wave < 8) {
            // T[a][j] = sum_b M[a][b] A[b][j]:  j = 0: M0 + M1 + M2,  j = 1: M1 - M2 - M3;   y[i][j] = T[i][j] + sgn (T[i+1][j] + T[i+2][j])
            f32x4 y0 = {0.f, 0.f, 0.f, 0.f}, y1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float *xr = xa + (4 * (o_i + k)) * 1024;
                const f32x4 m0 = *reinterpret_cast<const f32x4 *>(xr), m1 = *reinterpret_cast<const f32x4 *>(xr + 1024);
                const f32x4 m2 = *reinterpret_cast<const f32x4 *>(xr + 2048), m3 = *reinterpret_cast<const f32x4 *>(xr + 3072);
                const f32x4 t0 = (m0 + m1) + m2, t1 = (m1 - m2) - m3;
                if (k == 0) { y0 = t0; y1 = t1; }
                else { y0 = y0 + sgn * t0; y1 = y1 + sgn * t1; }
            }
            const int oy = oy0 + 8 * mt + 4 * qh + 2 * o_tr + o_i, ox = ox0 + 2 * o_tx;
            float *o = ob + (size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)(oy * p.Wo + ox) * p.out_pix_stride + (co0 & 7);
            f32x4 v0 = y0 + bias, v1 = y1 + bias;
            if (p.leaky) { v0 = __builtin_elementwise_max(v0, 0.2f * v0); v1 = __builtin_elementwise_max(v1, 0.2f * v1); }
            if (col_ok && oy < p.Ho && ox < p.Wo) *reinterpret_cast<f32x4 *>(o) = v0;
            if (col_ok && oy < p.Ho && ox + 1 < p.Wo) *reinterpret_cast<f32x4 *>(o + p.out_pix_stride) = v1;
        }
    }
}

bool wino2s_supported(const ConvLaunch &p)
{
    if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo || !p.wpk_split2) return false;
    if (p.nseg > 1 && p.seg[1].pix_stride != p.seg[0].pix_stride) return false;
    if (((p.out_pix_stride | (int)p.out_chunk_stride) & 3) != 0 || (p.cout & 3) != 0) return false;   // 16-byte stores
    for (int i = 0; i < p.nseg; ++i)
        if ((double)p.seg[i].nchunks * (double)p.seg[i].chunk_stride * 4.0 >= 2147483648.0) return false;
    return (double)p.H * p.W * p.seg[0].pix_stride * 4.0 < 2147483648.0;
}

// n-blocks [nb0, nb0 + nblk) of 64 outputs each
hipError_t launch_conv3x3_wino2s(const ConvLaunch &p, int nb0, int nblk, hipStream_t s)
{
    using namespace wino2s;
    static bool attr_done_dev[64] = {false};
    bool &attr_done = attr_done_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wino2s), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    ConvLaunch q = p;
    q.nb0 = nb0;
    q.nblk = nblk;
    q.trace = nullptr;
    const int tiles = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH);
    hipLaunchKernelGGL(conv3x3_wino2s, dim3((unsigned)(tiles * p.nimg * nblk)), dim3(1024), LDS_BYTES, s, q);
    return hipGetLastError();
}

size_t wino2s_wpk_floats(int cin_chunks, int nblk) { return (size_t)nblk * cin_chunks * (wino2s::UC_BYTES / 4); }

static inline unsigned short w2s_bf16_rne(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float w2s_bf16_f32(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// U = G g G^T in double, rounded once to fp32, then split exactly into three bf16 terms; packed
// [nblk][chunk]{ [xi 16][k4 2][co 64] x (Um01 Um23 Uh01 Uh23) | [xi][k4][co] x (Uh01 Uh23 Ul01 Ul23) }
void wino2s_pack_weights(const float *w, int Co, int Ci, const int *cin_map, int cin_chunks, int nblk, float *wpk)
{
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> U((size_t)Co * Ci * 16);
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci) {
            const float *gk = w + ((size_t)co * Ci + ci) * 9;
            double t[4][3];
            for (int a = 0; a < 4; ++a)
                for (int v = 0; v < 3; ++v) t[a][v] = G[a][0] * gk[0 * 3 + v] + G[a][1] * gk[1 * 3 + v] + G[a][2] * gk[2 * 3 + v];
            for (int a = 0; a < 4; ++a)
                for (int bq = 0; bq < 4; ++bq)
                    U[((size_t)co * Ci + ci) * 16 + a * 4 + bq] = (float)(t[a][0] * G[bq][0] + t[a][1] * G[bq][1] + t[a][2] * G[bq][2]);
        }
    unsigned short *out = reinterpret_cast<unsigned short *>(wpk);
    const size_t uc = wino2s::UC_BYTES / 2, u4 = wino2s::U4_BYTES / 2;   // in bf16 units
    for (int nbk = 0; nbk < nblk; ++nbk)
        for (int c = 0; c < cin_chunks; ++c) {
            unsigned short *blk = out + ((size_t)nbk * cin_chunks + c) * uc;
            for (int xi = 0; xi < 16; ++xi)
                for (int h = 0; h < 2; ++h)
                    for (int nn = 0; nn < 64; ++nn)
                        for (int j = 0; j < 4; ++j) {
                            const int co = nbk * 64 + nn;
                            const int k = c * kCK + h * 4 + j;
                            const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                            float v = 0.f;
                            if (co < Co && ci >= 0) v = U[((size_t)co * Ci + ci) * 16 + xi];
                            const unsigned short hh = w2s_bf16_rne(v);
                            const float r1 = v - w2s_bf16_f32(hh);
                            const unsigned short mm = w2s_bf16_rne(r1);
                            const float r2 = r1 - w2s_bf16_f32(mm);
                            const unsigned short ll = w2s_bf16_rne(r2);
                            const size_t ln = (size_t)(xi * 2 + h) * 64 + nn;
                            blk[ln * 8 + j] = mm;            // plane 0: (Um01 Um23 Uh01 Uh23)
                            blk[ln * 8 + 4 + j] = hh;
                            blk[u4 + ln * 8 + j] = hh;       // plane 1: (Uh01 Uh23 Ul01 Ul23)
                            blk[u4 + ln * 8 + 4 + j] = ll;
                        }
        }
}

}  // namespace b2f
