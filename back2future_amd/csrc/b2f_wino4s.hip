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
// Block = 256 threads = ONE wave per SIMD with the 512-register budget: wave w owns xi = 9w .. 9w+8 for BOTH N tiles (18
// accumulators = 288 registers), so each V element is read and split exactly once, nothing is shared between the waves of a
// SIMD, and MFMAs and the VALU / LDS / memory instructions of the SAME wave interleave (the bf16 pipe takes an MFMA every 32
// cycles; what a wave issues in between is free).  One barrier per 8-channel chunk, as in the fp32 kernel.
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
constexpr int NSTG = 2 * 3 * PW;            // 204 staging threads: (patch row mod 3, patch column, k4), six rows each
__device__ __host__ constexpr int colpos(int p) { return (p & 3) * 9 + (p >> 2); }
}  // namespace wino4s

#define W4S_FMA(a_, b_, c_) __builtin_elementwise_fma((a_), (b_), (c_))

// fp32 quad -> window [m01 m23 | h01 h23 | l01 l23] of bf16 pairs (round to nearest even; x = h + m + l exactly)
__device__ __forceinline__ void w4s_split(const f32x4 v, unsigned (&w)[6])
{
    const f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
    const unsigned h01 = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, bf16x2));
    const unsigned h23 = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, bf16x2));
    const f32x2 hf01 = {__builtin_bit_cast(float, h01 << 16), __builtin_bit_cast(float, h01 & 0xffff0000u)};
    const f32x2 hf23 = {__builtin_bit_cast(float, h23 << 16), __builtin_bit_cast(float, h23 & 0xffff0000u)};
    const f32x2 r01 = v01 - hf01, r23 = v23 - hf23;
    const unsigned m01 = __builtin_bit_cast(unsigned, __builtin_convertvector(r01, bf16x2));
    const unsigned m23 = __builtin_bit_cast(unsigned, __builtin_convertvector(r23, bf16x2));
    const f32x2 mf01 = {__builtin_bit_cast(float, m01 << 16), __builtin_bit_cast(float, m01 & 0xffff0000u)};
    const f32x2 mf23 = {__builtin_bit_cast(float, m23 << 16), __builtin_bit_cast(float, m23 & 0xffff0000u)};
    const f32x2 l01 = r01 - mf01, l23 = r23 - mf23;
    w[0] = m01; w[1] = m23; w[2] = h01; w[3] = h23;
    w[4] = __builtin_bit_cast(unsigned, __builtin_convertvector(l01, bf16x2));
    w[5] = __builtin_bit_cast(unsigned, __builtin_convertvector(l23, bf16x2));
}

#ifndef B2F_W4S_ASM8
#define B2F_W4S_ASM8 1       // 0: the ninth accumulator through the builtin like the others (the compiler then swaps accumulators)
#endif
#ifndef B2F_W4S_ABLATE
#define B2F_W4S_ABLATE 0     // profiling only (wrong results): 1 no input transform, 2 no raw staging, 4 no B loads, 8 no MFMAs, 16 no split
#endif

__global__ __launch_bounds__(256) void conv3x3_wino4s(const ConvLaunch p)
{
    using namespace wino4s;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VSTRIDE = XQ_F4;                               // V 1 starts one exchange buffer after V 0
    f32x4 *Rb = reinterpret_cast<f32x4 *>(smem);                 // [2][RAW_F4]
    f32x4 *Vb = Rb + 2 * RAW_F4;                                  // V 0 | gap | V 1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, half = lane >> 5;

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int total = tiles_x * tiles_y * p.nimg * p.nblk;
    const int G = gridDim.x;
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    if ((int)blockIdx.x >= total) return;

    // ---- staging of the raw patch: thread tid < 204 = (patch row r3 < 3, patch column px < 34, k4 = tid & 1) stages the six
    // pixels (r3 + 3 i, px), i = 0..5: one LDS slot and one byte offset per thread, item i adds an immediate to the slot and
    // a scalar to the offset.  The tile enters through the base of the buffer resource and six 64-bit lane masks (lanes whose
    // pixel lies inside the image; the others load at offset -16, which the range check of the buffer load turns into the
    // zero padding of the convolution).
    const bool s_act = tid < NSTG;
    int s_slot;
    unsigned l_off;
    {
        const int pix = min(tid, NSTG - 1) >> 1;
        const int r3 = pix / PW, px = pix - r3 * PW;
        s_slot = (tid & 1) * RAW_P + r3 * RW + colpos(px);
        l_off = ((unsigned)(r3 * p.W + px) * (unsigned)p.seg[0].pix_stride + (tid & 1) * 4) * 4u;
    }
    const int rowblk = 3 * p.W * p.seg[0].pix_stride * 4;        // bytes between the items of a thread
    typedef unsigned long long u64;
    u64 mk[6], mk_n[6];                                           // load side's tile / the block's next tile
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
#define W4S_MASKS(ox0_, oy0_, out_)                                                                 \
    do {                                                                                            \
        const int pix__ = min(tid, NSTG - 1) >> 1;                                                  \
        const int r3__ = pix__ / PW, px__ = pix__ - r3__ * PW;                                      \
        const int gx = (ox0_) - 1 + px__;                                                           \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                             \
            const int gy = (oy0_) - 1 + r3__ + 3 * i;                                               \
            out_[i] = __builtin_amdgcn_ballot_w64(tid < NSTG && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W); \
        }                                                                                           \
    } while (0)
#define W4S_RSRC(img_, ox0_, oy0_)                                                                  \
    do {                                                                                            \
        const long long o__ = ((long long)((oy0_) - 1) * p.W + ((ox0_) - 1)) * p.seg[0].pix_stride; \
        r_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[0].ptr) + ((long long)(img_) * p.seg[0].img_stride + o__), 0, 0x7fffffff, 0x00020000); \
        r_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[1].ptr) + ((long long)(img_) * p.seg[1].img_stride + o__), 0, 0x7fffffff, 0x00020000); \
    } while (0)
    f32x4 sr[3];
    // The load stream: chunk after chunk of the block's tiles, each chunk in two halves (items 0..2, then 3..5) so that only
    // three staging registers are live at a time (24 were spilled -- behind vmcnt(0) -- in the first build of this kernel).
    // W4S_LOAD_STREAM(h): half h of the stream's current chunk -> sr; after the second half of a tile's last chunk the stream
    // moves on to the block's next tile (and keeps re-reading the very last chunk when there is none: harmless).
#define W4S_LOAD_STREAM(h_)                                                                         \
    do {                                                                                            \
        const bool s1 = lc >= p.seg[0].nchunks;                                                     \
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                       \
        const int cc = s1 ? lc - p.seg[0].nchunks : lc;                                             \
        const int so = (int)(cc * cstr * 4);                                                        \
        if (!(B2F_W4S_ABLATE & 2)) {                                                                \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                         \
                unsigned vo__;                                                                      \
                asm("v_cndmask_b32_e64 %0, -16, %1, %2" : "=v"(vo__) : "v"(l_off), "s"(mk[3 * (h_) + i])); \
                sr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_rsrc1 : r_rsrc0, (int)vo__, so + (3 * (h_) + i) * rowblk, 0)); \
            }                                                                                       \
        }                                                                                           \
        if ((h_) == 1 && ++lc == nchunks) {                                                         \
            if (has_next) {                                                                         \
                lc = 0;                                                                             \
                _Pragma("unroll") for (int i = 0; i < 6; ++i) mk[i] = mk_n[i];                      \
                W4S_RSRC(nxt_img, nxt_ox0, nxt_oy0);                                                \
            } else {                                                                                \
                lc = nchunks - 1;                                                                   \
            }                                                                                       \
        }                                                                                           \
    } while (0)
#define W4S_WRITE_RAW(buf_, h_)                                                                     \
    do {                                                                                            \
        f32x4 *r__ = Rb + (buf_) * RAW_F4 + s_slot;                                                 \
        if (s_act && !(B2F_W4S_ABLATE & 2)) {                                                       \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) r__[3 * (3 * (h_) + i) * RW] = sr[i];     \
        }                                                                                           \
    } while (0)

    // ---- input transform V = B^T d B of a chunk (the arithmetic of b2f_wino4.hip, element for element): lane = (tile = lane
    // & 31, k4 = lane >> 5), channel pair th = wave & 1 of the k4 group; each wave runs two roles one after the other:
    //   role A (two rows with shared sub-expressions)   waves 0, 1: rows 1, 2 = (d4 - 4 d2) +- (d3 - 4 d1)
    //                                                   waves 2, 3: rows 3, 4 = (d4 - d2) +- 2 (d3 - d1)
    //   role B (one row)                                waves 0, 1: row 0 = 4 d0 + (d4 - 5 d2);  waves 2, 3: row 5 = 4 d1 + (d5 - 5 d3)
    // 15 slices per chunk: A0..A5 = column s of the 6-wide window (4 row reads, 4 packed ops), A6 / A7 = 6-point column pass
    // of the first / second produced row + six 8-byte LDS writes, B0..B5 (3 row reads, 2 packed ops), B6 column pass.
    const int rp = wave >> 1;
    const int th = wave & 1;
    const float a_ca = rp ? -1.f : -4.f, a_cs = rp ? 2.f : 1.f;  // role A: P = fma(ca, d2, d4), Q = fma(ca, d1, d3), r = P +- cs Q
    const int a_orow = rp ? 3 : 1;
    const int b_r1 = rp ? 3 : 2, b_r2 = rp ? 5 : 4, b_r3 = rp ? 1 : 0, b_orow = rp ? 5 : 0;   // role B: P = fma(-5, d[r1], d[r2]), r = fma(4, d[r3], P)
    const int t_base = half * RAW_P + (4 * (m >> 3)) * RW + (m & 7);
    const int ta_row[4] = {t_base + 2 * RW, t_base + 4 * RW, t_base + 1 * RW, t_base + 3 * RW};
    const int tb_row[3] = {t_base + b_r1 * RW, t_base + b_r2 * RW, t_base + b_r3 * RW};
    const int ta_dst = (a_orow * 6 * 2 + half) * 32 + m;        // float4 index of V[xi = 6 row][k4][tile]; xi + 1 -> + 64, next row -> + 384
    const int tb_dst = (b_orow * 6 * 2 + half) * 32 + m;
    f32x4 RA[6];                                                // role A: (row a | row b) of tile column j, one channel pair
    f32x2 RB[6];
    f32x2 da[2][4], db[4][3];                                   // rows read one step ahead of their use
#define W4S_TA_READ(s_, rbuf_, k_)                                                                  \
    do {                                                                                            \
        const f32x2 *rp__ = reinterpret_cast<const f32x2 *>(Rb + (rbuf_) * RAW_F4 + colpos(s_)) + th; \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) da[k_][r] = rp__[2 * ta_row[r]];             \
    } while (0)
#define W4S_TA_FMA(s_, k_)                                                                          \
    do {                                                                                            \
        const f32x2 ca2 = {a_ca, a_ca}, cs2 = {a_cs, a_cs};                                         \
        const f32x2 P = W4S_FMA(ca2, da[k_][0], da[k_][1]), Q = W4S_FMA(ca2, da[k_][2], da[k_][3]); \
        const f32x2 oa = W4S_FMA(cs2, Q, P), ob = W4S_FMA(-cs2, Q, P);                              \
        RA[s_] = __builtin_shufflevector(oa, ob, 0, 1, 2, 3);                                       \
    } while (0)
#define W4S_TB_READ(s_, rbuf_, k_)                                                                  \
    do {                                                                                            \
        const f32x2 *rp__ = reinterpret_cast<const f32x2 *>(Rb + (rbuf_) * RAW_F4 + colpos(s_)) + th; \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) db[k_][r] = rp__[2 * tb_row[r]];             \
    } while (0)
#define W4S_TB_FMA(s_, k_)                                                                          \
    do {                                                                                            \
        const f32x2 k5 = {-5.f, -5.f}, k4c = {4.f, 4.f};                                            \
        const f32x2 P = W4S_FMA(k5, db[k_][0], db[k_][1]);                                          \
        RB[s_] = W4S_FMA(k4c, db[k_][2], P);                                                        \
    } while (0)
    // column pass of one produced row: V[a][.] = r B (12 packed ops), six 8-byte LDS writes
#define W4S_COLPASS(r0, r1, r2, r3, r4, r5, dst_, vbuf_)                                            \
    do {                                                                                            \
        const f32x2 k4v = {4.f, 4.f}, k5v = {-5.f, -5.f}, km4 = {-4.f, -4.f}, k2v = {2.f, 2.f}, km2 = {-2.f, -2.f}; \
        const f32x2 t0 = W4S_FMA(k5v, (r2), W4S_FMA(k4v, (r0), (r4)));                              \
        const f32x2 pq = W4S_FMA(km4, (r2), (r4)), qq = W4S_FMA(km4, (r1), (r3));                   \
        const f32x2 uu = (r4) - (r2), vv = (r3) - (r1);                                             \
        const f32x2 t1 = W4S_FMA(k5v, (r3), W4S_FMA(k4v, (r1), (r5)));                              \
        const f32x2 o1 = pq + qq, o2 = pq - qq;                                                     \
        const f32x2 o3 = W4S_FMA(k2v, vv, uu), o4 = W4S_FMA(km2, vv, uu);                           \
        f32x2 *v__ = reinterpret_cast<f32x2 *>(Vb + (vbuf_) * VSTRIDE + (dst_)) + th;               \
        v__[0] = t0; v__[2 * 64] = o1; v__[2 * 128] = o2; v__[2 * 192] = o3; v__[2 * 256] = o4; v__[2 * 320] = t1; \
    } while (0)
#define W4S_LO(x_) __builtin_shufflevector((x_), (x_), 0, 1)
#define W4S_HI(x_) __builtin_shufflevector((x_), (x_), 2, 3)
#define W4S_TA_COL(hh_, vbuf_)                                                                      \
    do {                                                                                            \
        if ((hh_) == 0) W4S_COLPASS(W4S_LO(RA[0]), W4S_LO(RA[1]), W4S_LO(RA[2]), W4S_LO(RA[3]), W4S_LO(RA[4]), W4S_LO(RA[5]), ta_dst, (vbuf_)); \
        else W4S_COLPASS(W4S_HI(RA[0]), W4S_HI(RA[1]), W4S_HI(RA[2]), W4S_HI(RA[3]), W4S_HI(RA[4]), W4S_HI(RA[5]), ta_dst + 384, (vbuf_)); \
    } while (0)
#define W4S_TB_COL(vbuf_) W4S_COLPASS(RB[0], RB[1], RB[2], RB[3], RB[4], RB[5], tb_dst, (vbuf_))
    // The transform of chunk k as a schedule over the xi steps of the two iterations before it (rd = raw buffer that holds
    // chunk k, vb = V buffer it goes to).  Reads are issued one step before the slice that uses them:
    //   iteration k - 2, after the barrier of step 6:  reads A0 A1 | step 7: A0 A1, reads A2 A3 | step 8: A2 A3, reads A4 A5
    //   iteration k - 1:  step 0: A4 A5 | 1: A6, reads B0..B3 | 2: A7 | 3: B0..B3, reads B4 B5 | 4: B4 B5 | 5: B6 | 6: raw write, barrier
#define W4S_TR_EARLY(x_, rd_)                                                                       \
    do {                                                                                            \
        if (!(B2F_W4S_ABLATE & 1)) {                                                                \
            if ((x_) == 6) { W4S_TA_READ(0, rd_, 0); W4S_TA_READ(1, rd_, 1); }                      \
            if ((x_) == 7) { W4S_TA_FMA(0, 0); W4S_TA_FMA(1, 1); W4S_TA_READ(2, rd_, 0); W4S_TA_READ(3, rd_, 1); } \
            if ((x_) == 8) { W4S_TA_FMA(2, 0); W4S_TA_FMA(3, 1); W4S_TA_READ(4, rd_, 0); W4S_TA_READ(5, rd_, 1); } \
        }                                                                                           \
    } while (0)
#define W4S_TR_LATE(x_, rd_, vb_)                                                                   \
    do {                                                                                            \
        if (!(B2F_W4S_ABLATE & 1)) {                                                                \
            if ((x_) == 0) { W4S_TA_FMA(4, 0); W4S_TA_FMA(5, 1); }                                  \
            if ((x_) == 1) { W4S_TA_COL(0, vb_); W4S_TB_READ(0, rd_, 0); W4S_TB_READ(1, rd_, 1); W4S_TB_READ(2, rd_, 2); W4S_TB_READ(3, rd_, 3); } \
            if ((x_) == 2) { W4S_TA_COL(1, vb_); }                                                  \
            if ((x_) == 3) { W4S_TB_FMA(0, 0); W4S_TB_FMA(1, 1); W4S_TB_FMA(2, 2); W4S_TB_FMA(3, 3); W4S_TB_READ(4, rd_, 0); W4S_TB_READ(5, rd_, 1); } \
            if ((x_) == 4) { W4S_TB_FMA(4, 0); W4S_TB_FMA(5, 1); }                                  \
            if ((x_) == 5) { W4S_TB_COL(vb_); }                                                     \
        }                                                                                           \
    } while (0)

    // ---- GEMM side: wave w owns xi = 9 w + x, x = 0..8, for both N tiles ----
    f32x16 acc[9][2];
    const int a_off = (9 * wave * 2 + half) * 32 + m;            // float4 index of V[xi = 9w][k4 = half][tile m]; xi + 1 -> + 64
    const unsigned b4_off = ((9 * wave * 2 + half) * 64 + m) * 16u;      // bytes: (Um Um Uh Uh) of [xi = 9w][k4][co m]; xi + 1 -> + 2048, N tile 1 -> + 512
    const unsigned b2_off = U4_BYTES + ((9 * wave * 2 + half) * 64 + m) * 8u;   // (Ul Ul); xi + 1 -> + 1024, N tile 1 -> + 256
    f32x4 avf[3];                                                 // fp32 A values, read two steps ahead
    unsigned wa[2][6];                                            // split A windows, one step ahead
    u32x4 bq[3][2];                                               // B ring: three slots, two steps (>= 12 MFMAs) ahead; 9 steps per chunk,
    u32x2 bl[3][2];                                               // so step x always uses slot x % 3
    __amdgpu_buffer_rsrc_t w_rsrc;
#define W4S_LOAD_U(slot_, c_, x_)                                                                   \
    do {                                                                                            \
        if (!(B2F_W4S_ABLATE & 4)) {                                                                \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                         \
                bq[slot_][n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)b4_off, (int)((c_) * UC_BYTES + (x_) * 2048 + n * 512), 0)); \
                bl[slot_][n] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(w_rsrc, (int)b2_off, (int)((c_) * UC_BYTES + (x_) * 1024 + n * 256), 0)); \
            }                                                                                       \
        }                                                                                           \
    } while (0)
    // 18 accumulators = 288 registers, the accumulator file holds 256: the ninth xi of a wave accumulates in ordinary VGPRs
    // (the "+v" form below); left to itself the compiler keeps swapping two accumulators through the accumulator file
#define W4S_MFMA_V(acc_, a_, b_) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc_) : "v"(a_), "v"(b_))
#define W4S_MFMA(x_, par_, slot_)                                                                   \
    do {                                                                                            \
        if (!(B2F_W4S_ABLATE & 8) && (x_) == 8 && B2F_W4S_ASM8) {                                   \
            const u32x4 a_mh = {wa[par_][0], wa[par_][1], wa[par_][2], wa[par_][3]};                \
            const u32x4 a_hl = {wa[par_][2], wa[par_][3], wa[par_][4], wa[par_][5]};                \
            const u32x4 b_hl0 = {bq[slot_][0][2], bq[slot_][0][3], bl[slot_][0][0], bl[slot_][0][1]}; \
            const u32x4 b_hl1 = {bq[slot_][1][2], bq[slot_][1][3], bl[slot_][1][0], bl[slot_][1][1]}; \
            W4S_MFMA_V(acc[8][0], a_mh, bq[slot_][0]); W4S_MFMA_V(acc[8][1], a_mh, bq[slot_][1]);   \
            W4S_MFMA_V(acc[8][0], a_hl, bq[slot_][0]); W4S_MFMA_V(acc[8][1], a_hl, bq[slot_][1]);   \
            W4S_MFMA_V(acc[8][0], a_mh, b_hl0); W4S_MFMA_V(acc[8][1], a_mh, b_hl1);                 \
        } else if (!(B2F_W4S_ABLATE & 8)) {                                                         \
            const u32x4 a_mh = {wa[par_][0], wa[par_][1], wa[par_][2], wa[par_][3]};                \
            const u32x4 a_hl = {wa[par_][2], wa[par_][3], wa[par_][4], wa[par_][5]};                \
            const u32x4 b_hl0 = {bq[slot_][0][2], bq[slot_][0][3], bl[slot_][0][0], bl[slot_][0][1]}; \
            const u32x4 b_hl1 = {bq[slot_][1][2], bq[slot_][1][3], bl[slot_][1][0], bl[slot_][1][1]}; \
            acc[x_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, bq[slot_][0]), acc[x_][0], 0, 0, 0); \
            acc[x_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, bq[slot_][1]), acc[x_][1], 0, 0, 0); \
            acc[x_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hl), __builtin_bit_cast(bf16x8, bq[slot_][0]), acc[x_][0], 0, 0, 0); \
            acc[x_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hl), __builtin_bit_cast(bf16x8, bq[slot_][1]), acc[x_][1], 0, 0, 0); \
            acc[x_][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, b_hl0), acc[x_][0], 0, 0, 0); \
            acc[x_][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, b_hl1), acc[x_][1], 0, 0, 0); \
        }                                                                                           \
    } while (0)
#define W4S_SPLIT(src_, par_)                                                                       \
    do {                                                                                            \
        if (!(B2F_W4S_ABLATE & 16)) w4s_split(avf[src_], wa[par_]);                                 \
        else { _Pragma("unroll") for (int k = 0; k < 6; ++k) wa[par_][k] = __builtin_bit_cast(unsigned, avf[src_][k & 3]); } \
    } while (0)

    // ---- first tile: prologue ----
    W4S_DECODE((int)blockIdx.x, cur_nb, cur_img, cur_ox0, cur_oy0);
    W4S_MASKS(cur_ox0, cur_oy0, mk);
    W4S_RSRC(cur_img, cur_ox0, cur_oy0);
    has_next = false;                                                           // no switch inside the prologue (nchunks >= 4)
    nxt_nb = cur_nb; nxt_img = cur_img; nxt_ox0 = cur_ox0; nxt_oy0 = cur_oy0;
#pragma unroll
    for (int i = 0; i < 6; ++i) mk_n[i] = mk[i];
    int par = 0;                                                                // parity (V / raw buffer) of the tile's chunk 0
    int v_cur = blockIdx.x;
    W4S_LOAD_STREAM(0); W4S_WRITE_RAW(0, 0); W4S_LOAD_STREAM(1); W4S_WRITE_RAW(0, 1);     // chunk 0
    W4S_LOAD_STREAM(0); W4S_WRITE_RAW(1, 0); W4S_LOAD_STREAM(1); W4S_WRITE_RAW(1, 1);     // chunk 1
    W4S_LOAD_STREAM(0);                      // first half of chunk 2, stays in flight
    __syncthreads();
    // Tr(0) -> V[0] in one go, then the early slices of Tr(1)
#pragma unroll
    for (int x = 6; x < 9; ++x) W4S_TR_EARLY(x, 0);
#pragma unroll
    for (int x = 0; x < 6; ++x) W4S_TR_LATE(x, 0, 0);
#pragma unroll
    for (int x = 6; x < 9; ++x) W4S_TR_EARLY(x, 1);
    __syncthreads();

    w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wpk) + (size_t)cur_nb * nchunks * UC_BYTES), 0, 0x7fffffff, 0x00020000);
    W4S_LOAD_U(0, 0, 0); W4S_LOAD_U(1, 0, 1);
    for (;;) {
        // ---- start of a tile: V[par] holds Tr(0), raw buffer par ^ 1 holds chunk 1 with the early slices of Tr(1) done (RA[0..3],
        // reads of A4 A5 in flight), the first half of chunk 2 is in flight in sr, the first two B operands are in flight ----
        has_next = v_cur + G < total;
        if (has_next) {
            W4S_DECODE(v_cur + G, nxt_nb, nxt_img, nxt_ox0, nxt_oy0);
            W4S_MASKS(nxt_ox0, nxt_oy0, mk_n);
        }
        avf[0] = Vb[par * VSTRIDE + a_off];
        avf[1] = Vb[par * VSTRIDE + a_off + 64];
        W4S_SPLIT(0, 0);
#pragma unroll
        for (int x = 0; x < 9; ++x)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][n][r] = 0.f;

        // One chunk of the software pipeline; PH_ = parity of the A window at step 0 (9 steps per chunk: the pattern repeats
        // every two chunks, the loop below is unrolled by two; an odd last chunk reuses phase 0), LAST_ = the tile's last chunk
        // (no B operands of a following chunk: they are fetched under the last output pass).
        //   step x: B of step x + 2 | fp32 A of step x + 2 | six MFMAs of step x, the split of step x + 1 and the transform
        //   slices of this step interleaved by the scheduler | (step 6: raw(c + 2) -> LDS, barrier, stream load of raw(c + 3))
#define W4S_CHUNK(PH_, c_, LAST_)                                                                   \
    do {                                                                                            \
        const int c = (c_);                                                                         \
        const int pc = (par + c) & 1;                                                               \
        const f32x4 *Vc = Vb + pc * VSTRIDE + a_off;                                                \
        const f32x4 *Vn = Vb + (pc ^ 1) * VSTRIDE + a_off;                                          \
        _Pragma("unroll") for (int x = 0; x < 9; ++x) {                                             \
            if (x + 2 < 9) W4S_LOAD_U((x + 2) % 3, c, x + 2);                                       \
            else if (!(LAST_)) W4S_LOAD_U((x + 2) % 3, c + 1, x + 2 - 9);                           \
            if (x <= 6) avf[(x + 2) % 3] = Vc[(x + 2) * 64];   /* every read of V[pc] is issued before the barrier of step 6 */ \
            else avf[(x + 2) % 3] = Vn[(x - 7) * 64];          /* steps 0, 1 of the next chunk, after it */ \
            W4S_MFMA(x, (9 * (PH_) + x) & 1, x % 3);                                                \
            W4S_SPLIT((x + 1) % 3, (9 * (PH_) + x + 1) & 1);                                        \
            if (x < 6) W4S_TR_LATE(x, pc ^ 1, pc ^ 1);                                              \
            /* raw(c + 2) -> raw buffer pc (free since the barrier of the previous iteration): first half (in flight since */ \
            /* that barrier) at step 1, then the second half is loaded and written at step 6, before this barrier */ \
            if (x == 1) { W4S_WRITE_RAW(pc, 0); W4S_LOAD_STREAM(1); }                               \
            if (x == 6) {                                                                           \
                W4S_WRITE_RAW(pc, 1);                                                               \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                __syncthreads();                                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                if (!(LAST_)) W4S_LOAD_STREAM(0);  /* the last chunk's is issued in the output stage */ \
            }                                                                                       \
            if (x >= 6) W4S_TR_EARLY(x, pc);                                                        \
            __builtin_amdgcn_sched_barrier(0);   /* the scheduler interleaves inside a step, never across steps */ \
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

        // ---- output: four passes (tile rows) through the exchange buffer = dead V buffer + gap (as conv3x3_wino4p<2>; a wave
        // dumps its nine xi planes for both N tiles, a thread transforms two items = (tile column, output column j, 4 channels)) ----
        const int pl = (par + nchunks - 1) & 1;                                 // V[pl] is dead, V[pl ^ 1] holds the next tile's Tr(0)
        unsigned dump_rel[4][2];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t8 = e + 4 * half;                                        // tile column inside the tile row
#pragma unroll
            for (int n = 0; n < 2; ++n) dump_rel[e][n] = 4u * (unsigned)((9 * wave * 8 + t8) * 64 + ((n * 32 + m) ^ (t8 << 3)));
        }
        const int o_tx = lane >> 3, o_j = (lane >> 1) & 3;
        const float o_sg = (o_j & 1) ? -1.f : 1.f;
        const float o_kq = o_j == 0 ? 1.f : o_j == 1 ? 2.f : o_j == 2 ? 4.f : 8.f;
        const float o_k0 = o_j == 0 ? 1.f : 0.f, o_k3 = o_j == 3 ? 1.f : 0.f;
        const int o_xe = o_j == 3 ? 5 * 512 : 0;                                // M5 for j = 3, M0 otherwise (weight 0 for j = 1, 2)
        float *X = reinterpret_cast<float *>(Vb + pl * V_F4);
        const unsigned xbase = static_cast<unsigned>(reinterpret_cast<size_t>(X));
        float *ob = p.out + (size_t)cur_img * p.out_img_stride;
        const int ox = cur_ox0 + 4 * o_tx + o_j;
        f32x4 bias[2];
        const float *xa[2];
        float *obase[2];
        bool col_ok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int o_cq = 2 * (2 * wave + i) + (lane & 1);
            const int co0 = cur_nb * 64 + 4 * o_cq;
            bias[i] = *reinterpret_cast<const f32x4 *>(p.bias + co0);
            xa[i] = X + o_tx * 64 + ((4 * o_cq) ^ (o_tx << 3));
            col_ok[i] = co0 < p.cout;
            obase[i] = ob + (size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)(cur_oy0 * p.Wo + ox) * p.out_pix_stride + (co0 & 7);
        }
#define W4S_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define W4S_PASS(q_, EXTRA_)                                                                        \
    do {                                                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                               \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                         \
                const unsigned da__ = xbase + dump_rel[e][n];                                       \
                _Pragma("unroll") for (int x = 0; x < 8; x += 2)                                    \
                    asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4"               \
                                 :: "v"(da__), "a"(acc[x][n][4 * (q_) + e]), "a"(acc[x + 1][n][4 * (q_) + e]), "n"(x * 8), "n"((x + 1) * 8) : "memory"); \
                asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(da__), "v"(acc[8][n][4 * (q_) + e]), "n"(8 * 8 * 256) : "memory"); \
            }                                                                                       \
        W4S_LDS_BARRIER();                                                                          \
        EXTRA_                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                             \
            const f32x4 sg4 = {o_sg, o_sg, o_sg, o_sg}, kq4 = {o_kq, o_kq, o_kq, o_kq};             \
            const f32x4 k04 = {o_k0, o_k0, o_k0, o_k0}, k34 = {o_k3, o_k3, o_k3, o_k3};             \
            const f32x4 k2 = {2.f, 2.f, 2.f, 2.f}, k4 = {4.f, 4.f, 4.f, 4.f}, k8 = {8.f, 8.f, 8.f, 8.f}; \
            f32x4 T[6];                                                                             \
            _Pragma("unroll") for (int a = 0; a < 6; ++a) {                                         \
                const float *xr6 = xa[i] + (6 * a) * 512;                                           \
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
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                         \
                f32x4 v = y[r] + bias[i];                                                           \
                if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);                            \
                if (col_ok[i] && oy + r < p.Ho && ox < p.Wo)                                        \
                    *reinterpret_cast<f32x4 *>(obase[i] + (size_t)((4 * (q_) + r) * p.Wo) * p.out_pix_stride) = v; \
            }                                                                                       \
        }                                                                                           \
        W4S_LDS_BARRIER();                                                                          \
    } while (0)
        // the stream load the last chunk skipped is issued after the first dump; the first B operands of the block's next tile
        // are fetched under the last pass (without a next tile the loads re-read this tile's and are dropped)
        W4S_PASS(0, W4S_LOAD_STREAM(0););
        W4S_PASS(1, );
        W4S_PASS(2, );
        W4S_PASS(3,
                 w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                     const_cast<char *>(reinterpret_cast<const char *>(p.wpk) + (size_t)nxt_nb * nchunks * UC_BYTES), 0, 0x7fffffff, 0x00020000);
                 W4S_LOAD_U(0, 0, 0); W4S_LOAD_U(1, 0, 1););
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
    hipLaunchKernelGGL(conv3x3_wino4s, dim3((unsigned)pgrid), dim3(256), LDS_BYTES, s, q);
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
