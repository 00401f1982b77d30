// Winograd F(4x4, 3x3) convolution on the BF16 matrix pipe with exactly split fp32 operands ("bf16x6") for gfx950 (MI355X):
// the wide stride-1 nn.SpatialConvolution(Ci,Co,3,3,1,1,1,1) [+ LeakyReLU(0.2)] layers of /root/reference/models/pwc.lua:62,78-82.
// Same interface, tiling, LDS layout, input / output transforms and persistent tile walk as conv3x3_wino4p<2> (b2f_wino4.hip);
// what differs is how the 36 GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] U_xi[ci][co]  are executed.
//
// Why: on gfx950 the fp32 MFMA runs at the fp32 VECTOR rate on the vector FMA hardware (tools/mfma_overlap.hip: an fp32 MFMA
// and VALU work of the same or of another wave add up, they never overlap), so the fp32 kernel pays MFMA time PLUS transform
// time.  The bf16 MFMA is its own pipe, 16x faster per MAC, and VALU work issues beside it.
//
// How fp32 accuracy is kept: every fp32 operand is split into three bf16 terms, x = xh + xm + xl with
//     xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm)              (round to nearest; both subtractions are exact,
// and xl is exactly representable: 3 x 8 significand bits cover fp32's 24), and of the nine term products the six of order
// <= 2^-16 are kept:  v u ~= vh uh + (vh um + vm uh) + (vh ul + vl uh + vm um).  The dropped ones (vm ul, vl um, vl ul) are
// below 2^-25 |v u|, under half an fp32 ulp of the product; the bf16 MFMA forms each product exactly and accumulates in fp32,
// so the result carries fp32-level error (tests/test_gpu_parity.py compares this kernel and the fp32 kernel with an fp64
// convolution: same error bars).  Six products at 1/16 of the fp32 MFMA's cost each = 2.67x less matrix-pipe time.
//
// The six products as THREE v_mfma_f32_32x32x16_bf16 per (xi, N tile) with no duplicated operand bytes: a lane (tile or co =
// lane & 31, k4 = lane >> 5) holds the 4 input channels of its k4 group as bf16 pairs in a WINDOW of six dwords
//     A = [Vm01 Vm23 | Vh01 Vh23 | Vl01 Vl23]        B = [Um01 Um23 | Uh01 Uh23 | Ul01 Ul23]
// and the K = 16 of one MFMA are (lane half = k4 group) x (two terms x 4 channels) = four consecutive dwords of a window:
//     A[0:3] B[0:3] = Vm Um + Vh Uh      A[2:5] B[0:3] = Vh Um + Vl Uh      A[0:3] B[2:5] = Vm Uh + Vh Ul.
// V stays fp32 in LDS exactly as in the fp32 kernel (same transform code, same 16-byte A read per xi); the wave that reads it
// splits it (18 VALU per xi: 6 v_cvt_pk_bf16_f32, 8 shifts / ands, 4 packed subtractions).  U is split once on the host and
// packed [n-block][chunk]{ [xi 36][k4 2][co 64] x (Um Um Uh Uh) | [xi][k4][co] x (Ul Ul) }: 24 bytes per lane and xi, one
// dwordx4 + one dwordx2 buffer load.
//
// Block = 512 threads = two waves per SIMD, wave w = g + 4 n owns xi = 9g .. 9g+8 of N tile n exactly as in conv3x3_wino4p<2>
// (9 accumulators): while one wave of a SIMD waits for memory or LDS the other one multiplies.  (The first form of this kernel
// ran ONE wave per SIMD with both N tiles and the 512-register budget: every V element split once, but every s_waitcnt idled
// the SIMD and 18 accumulators do not fit the 256-register accumulator file -- profiles/r04_wino4s_notes.txt (1).)
// One barrier per 8-channel chunk, B operands two xi steps ahead (three-slot ring), fp32 A values one step ahead.
#include "b2f_internal.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace b2f {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace wino4s {
constexpr int TH = 16, TW = 32;             // output pixels per block
constexpr int PH = TH + 2, PW = TW + 2;     // 18 x 34 input patch
constexpr int RW = 38;                      // float4 per patch row in LDS (as in b2f_wino4.hip)
constexpr int RAW_P = PH * RW;              // 684 float4 per k4 plane
constexpr int RAW_F4 = 2 * RAW_P;           // per raw buffer
constexpr int V_F4 = 36 * 2 * 32;           // per V buffer
constexpr int XQ_F4 = 36 * 8 * 64 / 4;      // exchange buffer of one tile row (8 tiles x 64 co x 36 xi)
constexpr int LDS_BYTES = 16 * (2 * RAW_F4 + XQ_F4 + V_F4);   // [raw 0 | raw 1 | V 0 | gap | V 1] = 154 368
constexpr int U4_BYTES = 36 * 2 * 64 * 16;  // (Um Um Uh Uh) plane of one (n-block, chunk)
constexpr int U2_BYTES = 36 * 2 * 64 * 8;   // (Ul Ul) plane
constexpr int UC_BYTES = U4_BYTES + U2_BYTES;
constexpr int NSTG = 2 * 6 * PW;            // 408 staging threads: (patch row mod 6, patch column, k4), three rows each
__device__ __host__ constexpr int colpos(int p) { return (p & 3) * 9 + (p >> 2); }
}  // namespace wino4s

#define W4S_FMA(a_, b_, c_) __builtin_elementwise_fma((a_), (b_), (c_))
#define W4S_F(a_, b_, c_) __builtin_fmaf((a_), (b_), (c_))

// fp32 quad -> window [m01 m23 | h01 h23 | l01 l23] of bf16 pairs (round to nearest even; x = h + m + l exactly).
// Scalar fp32 subtractions on purpose: a packed fp32 op next to bf16 MFMAs costs ~7 cycles of SIMD time and does not overlap
// them (tools/mfma_bf16_chain.hip); this file is compiled with -fno-slp-vectorize so that the compiler does not re-pack them.
__device__ __forceinline__ unsigned w4s_pk(float a, float b)
{
    // one v_cvt_pk_bf16_f32; NOT inline asm: the compiler must see the instruction to keep the VALU-write -> MFMA-read
    // wait states (an asm statement two instructions ahead of the MFMA that read its result gave garbage)
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));
}
__device__ __forceinline__ void w4s_split(const f32x4 v, unsigned (&w)[6])
{
    const unsigned h01 = w4s_pk(v[0], v[1]), h23 = w4s_pk(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, h01 << 16), r1 = v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h23 << 16), r3 = v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u);
    const unsigned m01 = w4s_pk(r0, r1), m23 = w4s_pk(r2, r3);
    const float l0 = r0 - __builtin_bit_cast(float, m01 << 16), l1 = r1 - __builtin_bit_cast(float, m01 & 0xffff0000u);
    const float l2 = r2 - __builtin_bit_cast(float, m23 << 16), l3 = r3 - __builtin_bit_cast(float, m23 & 0xffff0000u);
    w[0] = m01; w[1] = m23; w[2] = h01; w[3] = h23;
    w[4] = w4s_pk(l0, l1);
    w[5] = w4s_pk(l2, l3);
}

#ifndef W4S_ACC_REG
#define W4S_ACC_REG(x_) "v"(x_)      // register class the accumulators live in ("a" if the compiler keeps them in the accumulator file)
#endif
#ifndef B2F_W4S_ABLATE
#define B2F_W4S_ABLATE 0     // profiling only (wrong results): 1 no input transform, 2 no raw staging, 4 no B loads, 8 no MFMAs, 16 no split
#endif

__global__ __launch_bounds__(512) void conv3x3_wino4s(const ConvLaunch p)
{
    using namespace wino4s;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VSTRIDE = XQ_F4;                               // V 1 starts one exchange buffer after V 0
    f32x4 *Rb = reinterpret_cast<f32x4 *>(smem);                 // [2][RAW_F4]
    f32x4 *Vb = Rb + 2 * RAW_F4;                                  // V 0 | gap | V 1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave & 3, n = wave >> 2;

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int total = tiles_x * tiles_y * p.nimg * p.nblk;
    const int G = gridDim.x;
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    if ((int)blockIdx.x >= total) return;

    // ---- staging of the raw patch (as conv3x3_wino4p): thread tid < 408 = (patch row r6 < 6, patch column px < 34, k4 = tid & 1)
    // stages the three pixels (r6 + 6 i, px): one LDS slot and one byte offset per thread; the tile enters through the base of the
    // buffer resource and three 64-bit lane masks (lanes whose pixel lies outside the image load at offset -16, which the range
    // check of the buffer load turns into the zero padding of the convolution)
    const bool s_act = tid < NSTG;
    int s_slot;
    unsigned l_off;
    const int rowblk = 6 * p.W * p.seg[0].pix_stride * 4;        // bytes between the items of a thread
    typedef unsigned long long u64;
    u64 mk[3], mk_n[3];                                           // load side's tile / the block's next tile
    __amdgpu_buffer_rsrc_t r_rsrc0, r_rsrc1;
    int cur_nb, cur_img, cur_ox0, cur_oy0;
    int nxt_nb, nxt_img, nxt_ox0, nxt_oy0;
    bool has_next;
    int lc = 0;                                                   // load side of the pipeline: next chunk of its tile
#define W4S_DECODE(v_, nb_, img_, ox0_, oy0_)                                                       \
    do {                                                                                            \
        int bid__ = xcd_remap((v_), total);                                                         \
        nb_ = bid__ % p.nblk + p.nb0;                                                               \
        bid__ /= p.nblk;                                                                            \
        ox0_ = (bid__ % tiles_x) * TW;                                                              \
        bid__ /= tiles_x;                                                                           \
        oy0_ = (bid__ % tiles_y) * TH;                                                              \
        img_ = bid__ / tiles_y;                                                                     \
    } while (0)
    // lane masks of a tile, from the hardware lane id (no register held between tiles)
#define W4S_MASKS(ox0_, oy0_, out_)                                                                 \
    do {                                                                                            \
        int z__ = 0;                                                                                \
        asm volatile("" : "+v"(z__));                                                               \
        const int ot__ = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z__)); \
        const int pix__ = min(ot__, NSTG - 1) >> 1;                                                 \
        const int r6__ = pix__ / PW, px__ = pix__ - r6__ * PW;                                      \
        const int gx = (ox0_) - 1 + px__;                                                           \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                             \
            const int gy = (oy0_) - 1 + r6__ + 6 * i;                                               \
            out_[i] = __builtin_amdgcn_ballot_w64(ot__ < NSTG && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W); \
        }                                                                                           \
    } while (0)
#define W4S_RSRC(img_, ox0_, oy0_)                                                                  \
    do {                                                                                            \
        const long long o__ = ((long long)((oy0_) - 1) * p.W + ((ox0_) - 1)) * p.seg[0].pix_stride; \
        r_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[0].ptr) + ((long long)(img_) * p.seg[0].img_stride + o__), 0, 0x7fffffff, 0x00020000); \
        r_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[1].ptr) + ((long long)(img_) * p.seg[1].img_stride + o__), 0, 0x7fffffff, 0x00020000); \
    } while (0)
    f32x4 sr[3];
    // next chunk of the load stream -> sr; after a tile's last chunk the stream moves on to the block's next tile (and keeps
    // re-reading the very last chunk when there is none: harmless)
#define W4S_LOAD_STREAM()                                                                           \
    do {                                                                                            \
        const bool s1 = lc >= p.seg[0].nchunks;                                                     \
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                       \
        const int cc = s1 ? lc - p.seg[0].nchunks : lc;                                             \
        const int so = (int)(cc * cstr * 4);                                                        \
        if (!(B2F_W4S_ABLATE & 2)) {                                                                \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                         \
                unsigned vo__;                                                                      \
                asm("v_cndmask_b32_e64 %0, -16, %1, %2" : "=v"(vo__) : "v"(l_off), "s"(mk[i]));     \
                sr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_rsrc1 : r_rsrc0, (int)vo__, so + i * rowblk, 0)); \
            }                                                                                       \
        }                                                                                           \
        if (++lc == nchunks) {                                                                      \
            if (has_next) {                                                                         \
                lc = 0;                                                                             \
                mk[0] = mk_n[0]; mk[1] = mk_n[1]; mk[2] = mk_n[2];                                  \
                W4S_RSRC(nxt_img, nxt_ox0, nxt_oy0);                                                \
            } else {                                                                                \
                lc = nchunks - 1;                                                                   \
            }                                                                                       \
        }                                                                                           \
    } while (0)
#define W4S_WRITE_RAW(buf_)                                                                         \
    do {                                                                                            \
        f32x4 *r__ = Rb + (buf_) * RAW_F4 + s_slot;                                                 \
        if (s_act && !(B2F_W4S_ABLATE & 2)) { r__[0] = sr[0]; r__[6 * RW] = sr[1]; r__[12 * RW] = sr[2]; } \
    } while (0)

    // ---- input transform V = B^T d B of a chunk: the arithmetic and the roles of b2f_wino4.hip, element for element.
    // Thread = (tile t = lane & 31, k4 = lane >> 5, channel pair th = wave & 1 of the k4 group's four); wave >> 1 selects the rows:
    //   waves 0, 1: rows 1, 2   P = d4 - 4 d2, Q = d3 - 4 d1, r1 = P + Q,  r2 = P - Q
    //   waves 2, 3: rows 3, 4   P = d4 -   d2, Q = d3 -   d1, r3 = P + 2Q, r4 = P - 2Q
    //   waves 4, 5: row 0       P = d4 - 5 d2, Q = d0,        r0 = P + 4Q
    //   waves 6, 7: row 5       P = d5 - 5 d3, Q = d1,        r5 = P + 4Q
    // one instruction stream  P = fma(ca, x1, x2), Q = fma(cb, x3, x4), rA = fma(cs, Q, P), rB = fma(-cs, Q, P)  with wave-uniform
    // rows and coefficients, then the 6-point column pass of each produced row.  Slice s < 6: column s of the 6-wide window;
    // slice 6 / 7: column pass + LDS writes of the first / second produced row.
    const int t_role = wave >> 1;
    const int th = wave & 1;
    int rx1, rx2, rx3, rx4, t_orow;
    float t_ca, t_cb, t_cs;
    switch (t_role) {
    case 0: rx1 = 2; rx2 = 4; rx3 = 1; rx4 = 3; t_ca = -4.f; t_cb = -4.f; t_cs = 1.f; t_orow = 1; break;
    case 1: rx1 = 2; rx2 = 4; rx3 = 1; rx4 = 3; t_ca = -1.f; t_cb = -1.f; t_cs = 2.f; t_orow = 3; break;
    case 2: rx1 = 2; rx2 = 4; rx3 = 0; rx4 = 0; t_ca = -5.f; t_cb = 0.f; t_cs = 4.f; t_orow = 0; break;
    default: rx1 = 3; rx2 = 5; rx3 = 1; rx4 = 1; t_ca = -5.f; t_cb = 0.f; t_cs = 4.f; t_orow = 5; break;
    }
    const bool t_two = t_role < 2;
    int t_row[4], t_dst;
    f32x4 R[6], d[2];
#define W4S_T_READ(s_, rbuf_)                                                                       \
    do {                                                                                            \
        if ((s_) < 6 && !(B2F_W4S_ABLATE & 1)) {                                                    \
            const f32x2 *rp = reinterpret_cast<const f32x2 *>(Rb + (rbuf_) * RAW_F4 + colpos(s_)) + th; \
            const f32x2 x1 = rp[2 * t_row[0]], x2 = rp[2 * t_row[1]], x3 = rp[2 * t_row[2]], x4 = rp[2 * t_row[3]]; \
            d[0] = __builtin_shufflevector(x1, x2, 0, 1, 2, 3);                                     \
            d[1] = __builtin_shufflevector(x3, x4, 0, 1, 2, 3);                                     \
        }                                                                                           \
    } while (0)
    // packed ops as inline asm (pins the slice where it stands; the bf16 MFMA does co-issue with them)
#define W4S_T_FMA(s_, vbuf_)                                                                        \
    do {                                                                                            \
        if (B2F_W4S_ABLATE & 1) {                                                                   \
        } else if ((s_) < 6) {                                                                      \
            const f32x2 ca2 = {t_ca, t_ca}, cb2 = {t_cb, t_cb}, cs2 = {t_cs, t_cs}, cn2 = {-t_cs, -t_cs}; \
            const f32x2 x1 = __builtin_shufflevector(d[0], d[0], 0, 1), x2 = __builtin_shufflevector(d[0], d[0], 2, 3); \
            const f32x2 x3 = __builtin_shufflevector(d[1], d[1], 0, 1), x4 = __builtin_shufflevector(d[1], d[1], 2, 3); \
            /* (scalar fp32 FMAs, same values as the fp32 kernel's packed ones) */                  \
            const f32x2 P = {W4S_F(ca2[0], x1[0], x2[0]), W4S_F(ca2[1], x1[1], x2[1])};             \
            const f32x2 Q = {W4S_F(cb2[0], x3[0], x4[0]), W4S_F(cb2[1], x3[1], x4[1])};             \
            const f32x2 oa = {W4S_F(cs2[0], Q[0], P[0]), W4S_F(cs2[1], Q[1], P[1])};                \
            const f32x2 ob = {W4S_F(cn2[0], Q[0], P[0]), W4S_F(cn2[1], Q[1], P[1])};                \
            R[(s_) < 6 ? (s_) : 0] = __builtin_shufflevector(oa, ob, 0, 1, 2, 3);                   \
        } else if ((s_) == 6) {                                                                     \
            W4S_T_COLPASS(0, (vbuf_));                                                              \
        } else if (t_two) {                                                                         \
            W4S_T_COLPASS(1, (vbuf_));                                                              \
        }                                                                                           \
    } while (0)
#define W4S_T_COLPASS(hh_, vbuf_)                                                                   \
    do {                                                                                            \
        const f32x2 k4v = {4.f, 4.f}, k5v = {-5.f, -5.f}, km4 = {-4.f, -4.f}, k2v = {2.f, 2.f}, km2 = {-2.f, -2.f}; \
        f32x2 r0 = (hh_) ? __builtin_shufflevector(R[0], R[0], 2, 3) : __builtin_shufflevector(R[0], R[0], 0, 1); \
        f32x2 r1 = (hh_) ? __builtin_shufflevector(R[1], R[1], 2, 3) : __builtin_shufflevector(R[1], R[1], 0, 1); \
        f32x2 r2 = (hh_) ? __builtin_shufflevector(R[2], R[2], 2, 3) : __builtin_shufflevector(R[2], R[2], 0, 1); \
        f32x2 r3 = (hh_) ? __builtin_shufflevector(R[3], R[3], 2, 3) : __builtin_shufflevector(R[3], R[3], 0, 1); \
        f32x2 r4 = (hh_) ? __builtin_shufflevector(R[4], R[4], 2, 3) : __builtin_shufflevector(R[4], R[4], 0, 1); \
        f32x2 r5 = (hh_) ? __builtin_shufflevector(R[5], R[5], 2, 3) : __builtin_shufflevector(R[5], R[5], 0, 1); \
        f32x2 t0, t1, pq, qq, uu, vv, o1, o2, o3, o4;                                               \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                             \
            t0[e] = W4S_F(k5v[e], r2[e], W4S_F(k4v[e], r0[e], r4[e]));                              \
            pq[e] = W4S_F(km4[e], r2[e], r4[e]); qq[e] = W4S_F(km4[e], r1[e], r3[e]);               \
            uu[e] = r4[e] - r2[e]; vv[e] = r3[e] - r1[e];                                           \
            t1[e] = W4S_F(k5v[e], r3[e], W4S_F(k4v[e], r1[e], r5[e]));                              \
            o1[e] = pq[e] + qq[e]; o2[e] = pq[e] - qq[e];                                           \
            o3[e] = W4S_F(k2v[e], vv[e], uu[e]); o4[e] = W4S_F(km2[e], vv[e], uu[e]);               \
        }                                                                                           \
        f32x2 *v = reinterpret_cast<f32x2 *>(Vb + (vbuf_) * VSTRIDE + t_dst + 384 * (hh_)) + th;    \
        v[0] = t0; v[2 * 64] = o1; v[2 * 128] = o2; v[2 * 192] = o3; v[2 * 256] = o4; v[2 * 320] = t1; \
    } while (0)

    // ---- GEMM side: wave (g, n) owns xi = 9 g + x, x = 0..8, of N tile n ----
    f32x16 acc[9];
    int a_off;                                                    // float4 index of V[xi = 9g][k4 = lane >> 5][tile lane & 31]; xi + 1 -> + 64
    unsigned b4_off, b2_off;                                      // bytes: (Um Um Uh Uh) of [xi = 9g][k4][co]; xi + 1 -> + 2048 | (Ul Ul): xi + 1 -> + 1024
    f32x4 av[3];                                                  // fp32 A values, read one step ahead
    unsigned wa[6];                                               // the step's split A window
#ifndef W4S_LA
#define W4S_LA 3                                                  // B operands this many xi steps ahead (ring of 6 slots: the pattern repeats every two chunks)
#endif
    u32x4 bq[6];
    u32x2 bl[6];
    __amdgpu_buffer_rsrc_t w_rsrc;
#define W4S_LOAD_U(slot_, c_, x_)                                                                   \
    do {                                                                                            \
        if (!(B2F_W4S_ABLATE & 4)) {                                                                \
            /* ablate 32: every chunk reads chunk 0's weights (always cache-hot: is B latency a matter of L2 misses?) */ \
            bq[slot_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)b4_off, (int)(((B2F_W4S_ABLATE & 32) ? 0 : (c_)) * UC_BYTES + (x_) * 2048), 0)); \
            bl[slot_] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(w_rsrc, (int)b2_off, (int)(((B2F_W4S_ABLATE & 32) ? 0 : (c_)) * UC_BYTES + (x_) * 1024), 0)); \
        }                                                                                           \
    } while (0)
    // Anti-phase (W4S_ANTIPHASE): the two waves of a SIMD are w and w + 4, i.e. N tile 0 and N tile 1 of the same xi group, and run
    // the same instruction stream from the same barrier -- in phase, so both want the matrix pipe at the same time and both do
    // their VALU work at the same time.  The n = 0 waves split the A operand of a step right before its MFMAs, the n = 1 waves
    // split it at the end of the PREVIOUS step: one wave of a SIMD is in its VALU stretch while the other is in its MFMAs.
#ifndef W4S_STAGGER
#define W4S_STAGGER 0
#endif
#ifndef W4S_ANTIPHASE
#define W4S_ANTIPHASE 0      /* measured: the two code paths cost 269 spilled registers -- see W4S_STAGGER for the branch-free way */
#endif
#define W4S_SPLIT_A(i_)                                                                             \
    do {                                                                                            \
        if (!(B2F_W4S_ABLATE & 16)) w4s_split(av[(i_) % 3], wa);                                    \
        else { _Pragma("unroll") for (int k = 0; k < 6; ++k) wa[k] = __builtin_bit_cast(unsigned, av[(i_) % 3][k & 3]); } \
    } while (0)
#define W4S_MFMA(x_, slot_)                                                                         \
    do {                                                                                            \
        if (!W4S_ANTIPHASE || n == 0) W4S_SPLIT_A(x_);                                              \
        if (!(B2F_W4S_ABLATE & 8)) {                                                                \
            u32x4 a_mh = {wa[0], wa[1], wa[2], wa[3]};                                              \
            u32x4 a_hl = {wa[2], wa[3], wa[4], wa[5]};                                              \
            u32x4 b_hl = {bq[slot_][2], bq[slot_][3], bl[slot_][0], bl[slot_][1]};                  \
            /* the three MFMAs of a step chain through one accumulator: strictly back to back (the pipe forwards the */ \
            /* accumulator; ONE other instruction between two of them costs ~43 cycles, MI355X_MICROARCH.md) -- the */ \
            /* operand copies are made first, the partner wave of the SIMD issues its VALU work meanwhile */ \
            asm volatile("" : "+v"(a_mh), "+v"(a_hl), "+v"(b_hl));                                  \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            acc[x_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, bq[slot_]), acc[x_], 0, 0, 0); \
            acc[x_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hl), __builtin_bit_cast(bf16x8, bq[slot_]), acc[x_], 0, 0, 0); \
            acc[x_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, b_hl), acc[x_], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                      \
        }                                                                                           \
        if (W4S_ANTIPHASE && n != 0) W4S_SPLIT_A((x_) + 1);                                         \
    } while (0)
    // the per-lane constants of the main loop, recomputed from the hardware lane id at the start of every tile: held across the
    // output stage they would be spilled, and a spill reload waits (vmcnt counts in order) for the output stores
#define W4S_LANE_CONSTANTS()                                                                        \
    do {                                                                                            \
        int lz = 0;                                                                                 \
        asm volatile("" : "+v"(lz));                                                                \
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)lz)); \
        const int lm = ln & 31, lh = ln >> 5, ltid = wave * 64 + ln;                                \
        const int tb = lh * RAW_P + (4 * (lm >> 3)) * RW + (lm & 7);                                \
        t_row[0] = tb + rx1 * RW; t_row[1] = tb + rx2 * RW; t_row[2] = tb + rx3 * RW; t_row[3] = tb + rx4 * RW; \
        t_dst = (t_orow * 6 * 2 + lh) * 32 + lm;                                                    \
        a_off = (9 * g * 2 + lh) * 32 + lm;                                                         \
        b4_off = ((9 * g * 2 + lh) * 64 + n * 32 + lm) * 16u;                                       \
        b2_off = U4_BYTES + ((9 * g * 2 + lh) * 64 + n * 32 + lm) * 8u;                             \
        const int pix = min(ltid, NSTG - 1) >> 1;                                                   \
        const int r6 = pix / PW, px = pix - r6 * PW;                                                \
        s_slot = (ltid & 1) * RAW_P + r6 * RW + colpos(px);                                         \
        l_off = ((unsigned)(r6 * p.W + px) * (unsigned)p.seg[0].pix_stride + (ltid & 1) * 4) * 4u;  \
    } while (0)

    // ---- first tile: prologue as in the fp32 kernel ----
    W4S_LANE_CONSTANTS();
    W4S_DECODE((int)blockIdx.x, cur_nb, cur_img, cur_ox0, cur_oy0);
    W4S_MASKS(cur_ox0, cur_oy0, mk);
    W4S_RSRC(cur_img, cur_ox0, cur_oy0);
    has_next = false;                                                           // no switch inside the prologue (nchunks >= 4)
    nxt_nb = cur_nb; nxt_img = cur_img; nxt_ox0 = cur_ox0; nxt_oy0 = cur_oy0;
    mk_n[0] = mk[0]; mk_n[1] = mk[1]; mk_n[2] = mk[2];
    int par = 0;                                                                // parity (V / raw buffer) of the tile's chunk 0
    int v_cur = blockIdx.x;
    W4S_LOAD_STREAM(); W4S_WRITE_RAW(0);     // chunk 0
    W4S_LOAD_STREAM(); W4S_WRITE_RAW(1);     // chunk 1
    W4S_LOAD_STREAM();                       // chunk 2, stays in flight
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) { W4S_T_READ(s, 0); W4S_T_FMA(s, 0); }
    __syncthreads();

    w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wpk) + (size_t)cur_nb * nchunks * UC_BYTES), 0, 0x7fffffff, 0x00020000);
#define W4S_PRELOAD_U() do { _Pragma("unroll") for (int x = 0; x < W4S_LA; ++x) W4S_LOAD_U(x, 0, x); } while (0)
    W4S_PRELOAD_U();
    for (;;) {
        // ---- start of a tile: V[par] holds Tr(0), raw buffer par ^ 1 holds chunk 1, chunk 2 is in flight in sr, the first two
        // B operands are in flight (issued under the last output pass of the previous tile) ----
        has_next = v_cur + G < total;
        if (has_next) {
            W4S_DECODE(v_cur + G, nxt_nb, nxt_img, nxt_ox0, nxt_oy0);
            W4S_MASKS(nxt_ox0, nxt_oy0, mk_n);
        }
        W4S_LANE_CONSTANTS();
        W4S_T_READ(0, par ^ 1); W4S_T_FMA(0, 0);
        W4S_T_READ(1, par ^ 1); W4S_T_FMA(1, 0);
        W4S_T_READ(2, par ^ 1);
        av[0] = Vb[par * VSTRIDE + a_off];
        av[1] = Vb[par * VSTRIDE + a_off + 64];
        if (W4S_ANTIPHASE && n != 0) W4S_SPLIT_A(0);
#pragma unroll
        for (int x = 0; x < 9; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

        // One chunk of the software pipeline (the fp32 kernel's, with the split + three bf16 MFMAs in place of four fp32 MFMAs):
        //   iteration c, xi steps 0..5 : B of step x + 2, fp32 A of step x + 1; split + MFMAs of step x; slices 2..7 of Tr(c+1)
        //                xi step  6    : raw(c+2) -> LDS, A of xi 7, 8 fetched, BARRIER
        //                xi steps 7, 8 : A of xi 0, 1 of chunk c+1, slices 0, 1 of Tr(c+2); global loads of raw(c+3)
        // LAST_ = the tile's last chunk: no B operands of a following chunk (fetched under the last output pass instead)
    // Order inside a xi step (W4S_SCHED = 1, from the step trace of the fp32 kernel, profiles/r04_wino4_step_trace.txt: an
    // in-order wave that issues its loads, then multiplies, then runs its transform slice spends ~650 cycles per step on 96 cycles of
    // matrix pipe): all of the step's VALU work first -- the transform slice (rows read a step ago) and the split of the A operand
    // (read a step ago) -- then the three MFMAs with the step's memory instructions between them:
    //     slice VALU, split | MFMA 0 | rows of the next slice | MFMA 1 | B operand of step x + LA | MFMA 2 | A operand of step x + 1
    // While this wave is in its VALU stretch the other wave of the SIMD can be in its MFMAs (the bf16 pipe does not block the VALU).
#ifndef W4S_SCHED
#define W4S_SCHED 1
#endif
#define W4S_STEP_LOAD_U(PH_, x_, LAST_)                                                             \
    do {                                                                                            \
        if ((x_) + W4S_LA < 9) W4S_LOAD_U((9 * (PH_) + (x_) + W4S_LA) % 6, c, (x_) + W4S_LA);       \
        else if (!(LAST_)) W4S_LOAD_U((9 * (PH_) + (x_) + W4S_LA) % 6, c + 1, (x_) + W4S_LA - 9);   \
    } while (0)
#define W4S_STEP_A_READS(x_)                                                                        \
    do {                                                                                            \
        if ((x_) >= 1 && (x_) <= 6) av[((x_) + 1) % 3] = Vc[((x_) + 1) * 64];                       \
        if ((x_) == 6) av[8 % 3] = Vc[8 * 64];                                                      \
        if ((x_) == 7) av[0] = Vn[0];                                                               \
        if ((x_) == 8) av[1] = Vn[64];                                                              \
    } while (0)
#define W4S_CHUNK(PH_, c_, LAST_)                                                                   \
    do {                                                                                            \
        const int c = (c_);                                                                         \
        const int pc = (par + c) & 1;                                                               \
        const f32x4 *Vc = Vb + pc * VSTRIDE + a_off;                                                \
        const f32x4 *Vn = Vb + (pc ^ 1) * VSTRIDE + a_off;                                          \
        _Pragma("unroll") for (int x = 0; x < 9; ++x) {                                             \
            if (W4S_SCHED && !(B2F_W4S_ABLATE & 8)) {                                               \
                const int slot = (9 * (PH_) + x) % 6;                                               \
                if (x < 6) W4S_T_FMA(x + 2, pc ^ 1);                                                \
                else if (x == 7) W4S_T_FMA(0, 0);                                                   \
                else if (x == 8) W4S_T_FMA(1, 0);                                                   \
                W4S_SPLIT_A(x);                                                                     \
                u32x4 a_mh = {wa[0], wa[1], wa[2], wa[3]};                                          \
                u32x4 a_hl = {wa[2], wa[3], wa[4], wa[5]};                                          \
                u32x4 b_hl = {bq[slot][2], bq[slot][3], bl[slot][0], bl[slot][1]};                  \
                asm volatile("" : "+v"(a_mh), "+v"(a_hl), "+v"(b_hl));                              \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, bq[slot]), acc[x], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                if (x < 6) { if (x + 3 < 8) W4S_T_READ(x + 3, pc ^ 1); }                            \
                else if (x == 7) W4S_T_READ(1, pc);                                                 \
                else if (x == 8) W4S_T_READ(2, pc);                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hl), __builtin_bit_cast(bf16x8, bq[slot]), acc[x], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4S_STEP_LOAD_U(PH_, x, LAST_);                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, b_hl), acc[x], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4S_STEP_A_READS(x);                                                                \
                if (x == 6) {                                                                       \
                    W4S_WRITE_RAW(pc);                                                              \
                    __builtin_amdgcn_sched_barrier(0);                                              \
                    __syncthreads();                                                                \
                    if (W4S_STAGGER > 0 && n != 0) __builtin_amdgcn_s_sleep(W4S_STAGGER);            \
                    W4S_T_READ(0, pc);                                                              \
                    if (!(LAST_)) W4S_LOAD_STREAM();   /* the last chunk's is issued in the output stage */ \
                }                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
            } else {                                                                                \
            W4S_STEP_LOAD_U(PH_, x, LAST_);                                                         \
            W4S_STEP_A_READS(x);                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            W4S_MFMA(x, (9 * (PH_) + x) % 6);                                                       \
            if (x < 6) {                                                                            \
                W4S_T_FMA(x + 2, pc ^ 1);                                                           \
                if (x + 3 < 8) W4S_T_READ(x + 3, pc ^ 1);                                           \
            } else if (x == 6) {                                                                    \
                W4S_WRITE_RAW(pc);                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                __syncthreads();                                                                    \
                /* the two waves of a SIMD (w, w + 4) leave the barrier together and run the same stream in phase: both */ \
                /* want the matrix pipe at once, both do their VALU stretch at once.  The n = 1 waves sleep W4S_STAGGER */ \
                /* x 64 cycles here, once per chunk, so that one wave's MFMAs meet the other's VALU work */ \
                if (W4S_STAGGER > 0 && n != 0) __builtin_amdgcn_s_sleep(W4S_STAGGER);                \
                W4S_T_READ(0, pc);                                                                  \
                if (!(LAST_)) W4S_LOAD_STREAM();   /* the last chunk's is issued in the output stage */ \
            } else if (x == 7) {                                                                    \
                W4S_T_FMA(0, 0); W4S_T_READ(1, pc);                                                 \
            } else {                                                                                \
                W4S_T_FMA(1, 0); W4S_T_READ(2, pc);                                                 \
            }                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            }                                                                                       \
        }                                                                                           \
    } while (0)
        {
            int c2 = 0;
            for (; c2 + 2 < nchunks; c2 += 2) {
                W4S_CHUNK(0, c2, false);
                W4S_CHUNK(1, c2 + 1, false);
            }
            const bool even = c2 + 2 == nchunks;
            if (even) W4S_CHUNK(0, c2, false);
            if (even) W4S_CHUNK(1, c2 + 1, true);
            if (!even) W4S_CHUNK(0, c2, true);
        }
#undef W4S_CHUNK

        // ---- output: four passes (tile rows) through the exchange buffer = dead V buffer + gap, exactly as conv3x3_wino4p<2> ----
        const int pl = (par + nchunks - 1) & 1;                                 // V[pl] is dead, V[pl ^ 1] holds the next tile's Tr(0)
        int oz = 0;
        asm volatile("" : "+v"(oz));
        const int olane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)oz));
        const int om = olane & 31, ohalf = olane >> 5;
        unsigned dump_rel[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t8 = e + 4 * ohalf;                                       // tile column inside the tile row
            dump_rel[e] = 4u * (unsigned)((9 * g * 8 + t8) * 64 + ((n * 32 + om) ^ (t8 << 3)));
        }
        const int o_tx = olane >> 3, o_j = (olane >> 1) & 3, o_cq = 2 * wave + (olane & 1);
        const int o_rel = o_tx * 64 + ((4 * o_cq) ^ (o_tx << 3));               // float index inside a xi plane (512 floats)
        const float o_sg = (o_j & 1) ? -1.f : 1.f;
        const float o_kq = o_j == 0 ? 1.f : o_j == 1 ? 2.f : o_j == 2 ? 4.f : 8.f;
        const float o_k0 = o_j == 0 ? 1.f : 0.f, o_k3 = o_j == 3 ? 1.f : 0.f;
        const int o_xe = o_j == 3 ? 5 * 512 : 0;                                // M5 for j = 3, M0 otherwise (weight 0 for j = 1, 2)
        float *X = reinterpret_cast<float *>(Vb + pl * V_F4);
        const unsigned xbase = static_cast<unsigned>(reinterpret_cast<size_t>(X));
        float *ob = p.out + (size_t)cur_img * p.out_img_stride;
        const int co0 = cur_nb * 64 + 4 * o_cq;
        // bias of this wave's 8 channels through the scalar cache (a vector load here would wait, in order, behind the
        // raw-patch loads of the next tile that are in flight)
        typedef const __attribute__((address_space(4))) float cfloat;
        cfloat *bp = (cfloat *)(p.bias + cur_nb * 64 + 8 * wave);
        float bb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bb[k] = bp[k];
            asm volatile("" : "+s"(bb[k]));       // keeps the eight loads scalar
        }
        const f32x4 bias = (olane & 1) ? f32x4{bb[4], bb[5], bb[6], bb[7]} : f32x4{bb[0], bb[1], bb[2], bb[3]};
        const bool col_ok = co0 < p.cout;
        const int ox = cur_ox0 + 4 * o_tx + o_j;
        float *obase = ob + (size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)(cur_oy0 * p.Wo + ox) * p.out_pix_stride + (co0 & 7);
        const float *xa = X + o_rel;
#define W4S_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define W4S_PASS(q_, EXTRA_)                                                                        \
    do {                                                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
            const unsigned da = xbase + dump_rel[e];                                                \
            _Pragma("unroll") for (int x = 0; x < 8; x += 2)                                        \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4"                   \
                             :: "v"(da), W4S_ACC_REG(acc[x][4 * (q_) + e]), W4S_ACC_REG(acc[x + 1][4 * (q_) + e]), "n"(x * 8), "n"((x + 1) * 8) : "memory"); \
            asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(da), W4S_ACC_REG(acc[8][4 * (q_) + e]), "n"(8 * 8 * 256) : "memory"); \
        }                                                                                           \
        W4S_LDS_BARRIER();                                                                          \
        EXTRA_                                                                                      \
        {                                                                                           \
            const f32x4 sg4 = {o_sg, o_sg, o_sg, o_sg}, kq4 = {o_kq, o_kq, o_kq, o_kq};             \
            const f32x4 k04 = {o_k0, o_k0, o_k0, o_k0}, k34 = {o_k3, o_k3, o_k3, o_k3};             \
            const f32x4 k2 = {2.f, 2.f, 2.f, 2.f}, k4 = {4.f, 4.f, 4.f, 4.f}, k8 = {8.f, 8.f, 8.f, 8.f}; \
            f32x4 T[6];                                                                             \
            _Pragma("unroll") for (int a = 0; a < 6; ++a) {                                         \
                const float *xr6 = xa + (6 * a) * 512;                                              \
                const f32x4 m1 = *reinterpret_cast<const f32x4 *>(xr6 + 1 * 512), m2 = *reinterpret_cast<const f32x4 *>(xr6 + 2 * 512); \
                const f32x4 m3 = *reinterpret_cast<const f32x4 *>(xr6 + 3 * 512), m4 = *reinterpret_cast<const f32x4 *>(xr6 + 4 * 512); \
                const f32x4 me = *reinterpret_cast<const f32x4 *>(xr6 + o_xe);                      \
                const f32x4 e1 = W4S_FMA(sg4, m2, m1), e2 = W4S_FMA(sg4, m4, m3);                   \
                /* the fp32 kernel's output transform, element for element: */                      \
                /* j=0 (M0 + s1) + s2, j=1 fma(2, d2, d1), j=2 fma(4, s2, s1), j=3 fma(8, d2, d1) + M5 */ \
                T[a] = W4S_FMA(k34, me, W4S_FMA(kq4, e2, W4S_FMA(k04, me, e1)));                    \
            }                                                                                       \
            const f32x4 s1 = T[1] + T[2], d1 = T[1] - T[2], s2 = T[3] + T[4], d2 = T[3] - T[4];     \
            f32x4 y[4];                                                                             \
            y[0] = T[0] + s1 + s2;                                                                  \
            y[1] = W4S_FMA(k2, d2, d1);                                                             \
            y[2] = W4S_FMA(k4, s2, s1);                                                             \
            y[3] = W4S_FMA(k8, d2, d1) + T[5];                                                      \
            const int oy = cur_oy0 + 4 * (q_);                                                      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                         \
                f32x4 v = y[i] + bias;                                                              \
                if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);                            \
                if (col_ok && oy + i < p.Ho && ox < p.Wo)                                           \
                    *reinterpret_cast<f32x4 *>(obase + (size_t)((4 * (q_) + i) * p.Wo) * p.out_pix_stride) = v; \
            }                                                                                       \
        }                                                                                           \
        W4S_LDS_BARRIER();                                                                          \
    } while (0)
        // the stream load the last chunk skipped is issued after the first dump; the first B operands of the block's next tile
        // are fetched under the last pass (without a next tile the loads re-read this tile's and are dropped)
        W4S_PASS(0, W4S_LOAD_STREAM(););
        W4S_PASS(1, );
        W4S_PASS(2, );
        W4S_PASS(3,
                 w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                     const_cast<char *>(reinterpret_cast<const char *>(p.wpk) + (size_t)nxt_nb * nchunks * UC_BYTES), 0, 0x7fffffff, 0x00020000);
                 W4S_PRELOAD_U(););
#undef W4S_PASS
#undef W4S_LDS_BARRIER

        // ---- next tile of this block: the one whose Tr(0) the last iteration left in V[pl ^ 1] ----
        par = pl ^ 1;
        if (!has_next) break;
        v_cur += G;
        cur_nb = nxt_nb; cur_img = nxt_img; cur_ox0 = nxt_ox0; cur_oy0 = nxt_oy0;
    }
}

// the prologue of a block walks three chunks ahead of the tile switch: layers from 32 input channels
bool wino4s_supported(const ConvLaunch &p)
{
    return p.wpk_split != nullptr && p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0) >= 4;
}

hipError_t launch_conv3x3_wino4s(const ConvLaunch &p, int nb0, int nblk, hipStream_t s)
{
    using namespace wino4s;
    static bool attr_done_dev[64] = {false};
    static int n_cu_dev[64] = {0};
    bool &attr_done = attr_done_dev[attr_slot()];
    int &n_cu = n_cu_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wino4s), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
        n_cu &= ~7;                     // the XCD remap of the virtual block index needs a multiple of 8
        if (n_cu < 8) n_cu = 8;
    }
    ConvLaunch q = p;
    q.nb0 = nb0;
    q.nblk = nblk;
    q.trace = nullptr;
    q.wpk = reinterpret_cast<const float *>(p.wpk_split);
    const int tiles = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH);
    const int total = tiles * p.nimg * nblk;
    const int pcap = p.w4_persist > 1 ? p.w4_persist : n_cu;
    const int pgrid = pcap < total ? pcap : total;
    hipLaunchKernelGGL(conv3x3_wino4s, dim3((unsigned)pgrid), dim3(512), LDS_BYTES, s, q);
    return hipGetLastError();
}

size_t wino4s_wpk_floats(int cin_chunks, int nblk) { return (size_t)nblk * cin_chunks * (wino4s::UC_BYTES / 4); }

static inline unsigned short bf16_rne(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf16_f32(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// U = G g G^T in double, rounded once to fp32 (the values the fp32 kernel multiplies with), then split exactly into three
// bf16 terms; packed [nblk][chunk]{ [xi 36][k4 2][co 64] x (Um01 Um23 Uh01 Uh23) | [xi][k4][co] x (Ul01 Ul23) }
void wino4s_pack_weights(const float *w, int Co, int Ci, const int *cin_map, int cin_chunks, int nblk, float *wpk)
{
    static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    std::vector<float> U((size_t)Co * Ci * 36);
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci) {
            const float *gk = w + ((size_t)co * Ci + ci) * 9;
            double t[6][3];
            for (int a = 0; a < 6; ++a)
                for (int v = 0; v < 3; ++v) t[a][v] = G[a][0] * gk[0 * 3 + v] + G[a][1] * gk[1 * 3 + v] + G[a][2] * gk[2 * 3 + v];
            for (int a = 0; a < 6; ++a)
                for (int bq = 0; bq < 6; ++bq)
                    U[((size_t)co * Ci + ci) * 36 + a * 6 + bq] = (float)(t[a][0] * G[bq][0] + t[a][1] * G[bq][1] + t[a][2] * G[bq][2]);
        }
    unsigned short *out = reinterpret_cast<unsigned short *>(wpk);
    const size_t uc = wino4s::UC_BYTES / 2, u4 = wino4s::U4_BYTES / 2;   // in bf16 units
    for (int nbk = 0; nbk < nblk; ++nbk)
        for (int c = 0; c < cin_chunks; ++c) {
            unsigned short *blk = out + ((size_t)nbk * cin_chunks + c) * uc;
            for (int xi = 0; xi < 36; ++xi)
                for (int h = 0; h < 2; ++h)
                    for (int nn = 0; nn < 64; ++nn)
                        for (int j = 0; j < 4; ++j) {
                            const int co = nbk * 64 + nn;
                            const int k = c * kCK + h * 4 + j;
                            const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                            float v = 0.f;
                            if (co < Co && ci >= 0) v = U[((size_t)co * Ci + ci) * 36 + xi];
                            const unsigned short hh = bf16_rne(v);
                            const float r1 = v - bf16_f32(hh);
                            const unsigned short mm = bf16_rne(r1);
                            const float r2 = r1 - bf16_f32(mm);
                            const unsigned short ll = bf16_rne(r2);
                            const size_t lane = (size_t)(xi * 2 + h) * 64 + nn;
                            blk[lane * 8 + j] = mm;          // dwords 0, 1: Um (channel pairs 01, 23; even channel in the low half)
                            blk[lane * 8 + 4 + j] = hh;      // dwords 2, 3: Uh
                            blk[u4 + lane * 4 + j] = ll;     // (Ul Ul) plane
                        }
        }
}

}  // namespace b2f
