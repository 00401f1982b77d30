// Winograd F(4x4, 3x3) convolution on the fp32 MFMA for gfx950 (MI355X): the wide stride-1
// nn.SpatialConvolution(Ci,Co,3,3,1,1,1,1) [+ LeakyReLU(0.2)] layers (Co >= 64) of
// /root/reference/models/pwc.lua:62,78-82.  Same interface as b2f_wino.hip (chunk-planar in/out, up to two
// input K-segments, bias + LeakyReLU fused) with 4x fewer MFMA MACs than the direct form:
//
//   Y(4x4) = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        d = 6x6 input tile, g = 3x3 filter
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]                (Lavin & Gray 2015)
//
// 36 independent GEMMs M_xi[tile][co] = sum_ci V_xi[tile][ci] U_xi[ci][co] (xi = 6a + b) on
// v_mfma_f32_32x32x2_f32.  Why this form: on gfx950 nothing of the VALU kind overlaps an fp32 MFMA
// (tools/mfma_overlap.hip), so time ~ MFMA count + the rest; F(4x4) needs 2.25 MACs per output
// instead of 4 (F(2x2)) or 9 (direct), while its transforms cost about the same VALU work per
// output pixel as F(2x2)'s.  Arithmetic differs from the direct form by fp32 rounding of the
// transforms (coefficients up to 8): ~1e-6 relative on unit-scale data, end to end ~2e-7 absolute on
// the flow (tests hold 1e-3).
//
// Block = 512 threads (8 waves, one block per CU) -> 16 x 32 output pixels = 4 x 8 tiles = 32 tiles
// (one 32-row MFMA M tile), 64 output channels.  Wave w = g + 4 n owns xi = 9g .. 9g+8 for N tile n
// (9 accumulators = 144 VGPRs).  K is walked in chunks of 8 input channels:
//   raw  [2 buf][2 k4][18 rows][38] float4   input patch with halo, columns permuted (p & 3) * 9 + (p >> 2) so
//                                           that the 8 tiles of a row read consecutive float4 (no bank conflicts)
//   V    [2 buf][36 xi][2 k4][32 tiles] float4   transformed input (A operand)
//   U    per lane straight from global (B operand), two xi ahead
// One barrier per chunk (after xi 6, see the pipeline comment in the kernel).  Per xi the wave issues 4
// MFMAs, then -- in order, pinned with scheduling barriers -- one slice of the other work, so that every
// LDS / L2 latency sits under the MFMAs of the same wave: input transform (thread = (tile, k4, row a =
// wave): 8 row-slices of 3 fused multiply-adds on float4, then the 6-point column pass and 6 b128
// writes), staging of the raw patch (3 loads per thread, masked LDS writes 8 steps later).
// Output: accumulators -> LDS [xi][tile][co] per N tile (144 KB), one thread per (tile, 4 channels):
// A^T M A, bias, LeakyReLU, sixteen 16-byte stores.
#include "b2f_internal.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace b2f {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- hybrid steps (round 4): some of a wave's nine xi steps run on the BF16 matrix pipe with exactly split fp32 operands
// (b2f_wino4s.hip explains the split: x = xh + xm + xl, six of nine term products, fp32-level accuracy) instead of four fp32
// MFMAs.  A fp32 MFMA blocks the SIMD's matrix pipe AND its VALU for 64 cycles (tools/mfma_bf16_chain.hip mode 18: fp32 and
// bf16 MFMAs of two waves add up; r01: fp32 MFMA and VALU add up); three bf16 MFMAs hold the matrix pipe for 96 cycles and leave
// the VALU to the other wave, at the price of ~26 VALU instructions for the split.
__device__ __forceinline__ unsigned w4h_pk(float a, float b)
{
    // one v_cvt_pk_bf16_f32; NOT inline asm: the compiler must see the instruction to keep the VALU-write -> MFMA-read
    // wait states (an asm statement two instructions ahead of the MFMA that read its result gave garbage)
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));
}
// fp32 quad -> window [m01 m23 | h01 h23 | l01 l23] of bf16 pairs (round to nearest even; x = h + m + l exactly)
__device__ __forceinline__ void w4h_split(const f32x4 v, unsigned (&w)[6])
{
    const unsigned h01 = w4h_pk(v[0], v[1]), h23 = w4h_pk(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, h01 << 16), r1 = v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h23 << 16), r3 = v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u);
    const unsigned m01 = w4h_pk(r0, r1), m23 = w4h_pk(r2, r3);
    const float l0 = r0 - __builtin_bit_cast(float, m01 << 16), l1 = r1 - __builtin_bit_cast(float, m01 & 0xffff0000u);
    const float l2 = r2 - __builtin_bit_cast(float, m23 << 16), l3 = r3 - __builtin_bit_cast(float, m23 & 0xffff0000u);
    w[0] = m01; w[1] = m23; w[2] = h01; w[3] = h23;
    w[4] = w4h_pk(l0, l1);
    w[5] = w4h_pk(l2, l3);
}

namespace wino4 {
constexpr int TH = 16, TW = 32;             // output pixels per block
constexpr int PH = TH + 2, PW = TW + 2;     // 18 x 34 input patch
// LDS layouts (round 4): both the raw patch and V are stored as planes of channel PAIRS (f32x2), so that the 8-byte accesses
// of the input transform (a thread handles one channel pair) are bank-conflict free: a 32-lane group of a ds_read_b64 /
// ds_write_b64 covers 256 contiguous bytes.  Measured before (SQ_LDS_BANK_CONFLICT per ablation build, tools/w4_lds_conflicts.sh):
// 35 % of the kernel's LDS cycles were conflicts -- 49 % of them the transform's reads of the float4-per-pixel patch (tile rows
// t and t + 2 of a 32-lane group on the same banks), 36 % its writes of float4-per-tile V (tiles t and t + 16), 9 % the staging
// writes, 6 % the output stage, none the A operand reads or the accumulator dump (profiles/r04_wino4_notes.txt).
//   raw  [2 buf][k4 2][pair 2][18 rows][RWP] f32x2, columns permuted colpos(p); RWP = 2 (mod 8) puts the four tile rows of a
//        32-lane group 64 B apart in bank space
//   V    [2 buf][36 xi][k4 2][pair 2][32 tiles] f32x2; the A operand (4 channels of a tile) is one ds_read2_b64
constexpr int RWP = 42;                     // f32x2 per patch row of a pair plane
constexpr int PL2 = PH * RWP;               // 756 f32x2 per (k4, pair) plane
constexpr int RAW_F4 = 2 * PL2;             // float4 per raw buffer (4 planes)
constexpr int V_F4 = 36 * 2 * 32;           // per V buffer
constexpr int NITEM = 2 * PH * PW;          // 1224 (pixel, k4) staging items per chunk
constexpr int U_F4 = 36 * 2 * 64;           // float4 of packed weights per (n-block, chunk)
constexpr int STAGE_BYTES = 16 * (2 * V_F4 + 2 * RAW_F4);
constexpr int XCH_BYTES = 36 * 32 * 32 * 4;
constexpr int LDS_BYTES = STAGE_BYTES > XCH_BYTES ? STAGE_BYTES : XCH_BYTES;
constexpr int XQ_F4 = 36 * 8 * 64 / 4;                   // persistent kernel: exchange buffer of one tile row (8 tiles x 64 co x 36 xi)
constexpr int P_LDS_BYTES = 16 * (2 * RAW_F4 + XQ_F4 + V_F4);   // [raw 0 | raw 1 | V 0 | gap | V 1] = 158 976
constexpr int US4_BYTES = 36 * 2 * 64 * 16;  // split weights of one (n-block, chunk): (Um Um Uh Uh) plane, then the (Ul Ul) plane (b2f_wino4s.hip)
constexpr int USC_BYTES = US4_BYTES + 36 * 2 * 64 * 8;
__device__ __host__ constexpr int colpos(int p) { return (p & 3) * 9 + (p >> 2); }
}  // namespace wino4

#define W4_FMA(a_, b_, c_) __builtin_elementwise_fma((a_), (b_), (c_))
// float4 (4 channels of a pixel's k4 group) -> the two pair planes of a raw buffer; slot2_ = f32x2 index in pair plane 0
#define W4_RAW_STORE(buf4_, slot2_, v_)                                                             \
    do {                                                                                            \
        f32x2 *r2__ = reinterpret_cast<f32x2 *>(buf4_) + (slot2_);                                  \
        r2__[0] = __builtin_shufflevector((v_), (v_), 0, 1);                                        \
        r2__[wino4::PL2] = __builtin_shufflevector((v_), (v_), 2, 3);                               \
    } while (0)
// A operand: the 4 channels (two pair planes) of V[xi][k4][tile] as one ds_read2_b64; p2_ = f32x2 pointer at pair 0
#define W4_A_READ(p2_) __builtin_shufflevector((p2_)[0], (p2_)[32], 0, 1, 2, 3)

// Profiling only: -DB2F_WINO_TRACE=1 records clock64() at five points of every main-loop iteration of a
// few blocks (p.trace, set by the launcher when B2F_WINO_TRACE is in the environment).
#ifndef B2F_WINO_TRACE
#define B2F_WINO_TRACE 0
#endif
// Profiling only (results are wrong): -DB2F_WINO4_ABLATE=bits, 1 no input transform, 2 no raw staging, 4 no B loads,
// 8 no MFMAs, 16 no A operand reads; persistent two-N-tile kernel only: 64 no accumulator dump, 128 no output transform (LDS reads + stores),
// 256 transform without its LDS reads, 512 transform without its LDS writes
#ifndef B2F_WINO4_ABLATE
#define B2F_WINO4_ABLATE 0
#endif
#if B2F_WINO_TRACE
#define W4_T(k_) do { if (tr_on && lane == 0 && c < 32) tr_buf[(c * 5 + (k_))] = clock64(); } while (0)
#else
#define W4_T(k_) do {} while (0)
#endif

// Output stage of one N tile (32 channels), run by the 256 threads of the four waves that own it after they
// have dumped their accumulators to X[xi][tile row][co 32]: item = (tile, 4 channels), A^T M A, bias,
// LeakyReLU, sixteen 16-byte stores.
// TAG only labels the two inlined copies (asm comments): identical copies get merged back into one shared
// block by the compiler, and the accumulators of the second tile would be live (spilled) across it again.
template <int TAG>
__device__ __forceinline__ void wino4_output_tile(const float *X, const ConvLaunch &p, int idx, int co_base, int oy0, int ox0, float *ob)
{
    const int o_t = idx >> 3, o_cq = idx & 7;
    const int xr = o_t ^ ((o_t >> 2) & 1);
    const int co0 = co_base + 4 * o_cq;
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + co0);
    const float *xp = X + xr * 32 + 4 * o_cq;
    const f32x4 k2 = {2.f, 2.f, 2.f, 2.f}, k4 = {4.f, 4.f, 4.f, 4.f}, k8 = {8.f, 8.f, 8.f, 8.f};
    f32x4 T[6][4];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        f32x4 M[6];
        asm volatile("; output stage of N tile %0" ::"n"(TAG));
#pragma unroll
        for (int b = 0; b < 6; ++b) M[b] = *reinterpret_cast<const f32x4 *>(xp + (6 * a + b) * 1024);
        const f32x4 s1 = M[1] + M[2], d1 = M[1] - M[2], s2 = M[3] + M[4], d2 = M[3] - M[4];
        T[a][0] = M[0] + s1 + s2;
        T[a][1] = W4_FMA(k2, d2, d1);
        T[a][2] = W4_FMA(k4, s2, s1);
        T[a][3] = W4_FMA(k8, d2, d1) + M[5];
    }
    const int oy = oy0 + 4 * (o_t >> 3), ox = ox0 + 4 * (o_t & 7);
    // 16-byte stores; the launcher guarantees 4-aligned strides and cout % 4 == 0
    float *obase = ob + (size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)(oy * p.Wo + ox) * p.out_pix_stride + (co0 & 7);
    const bool col_ok = co0 < p.cout;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 s1 = T[1][j] + T[2][j], d1 = T[1][j] - T[2][j], s2 = T[3][j] + T[4][j], d2 = T[3][j] - T[4][j];
        f32x4 y[4];
        y[0] = T[0][j] + s1 + s2;
        y[1] = W4_FMA(k2, d2, d1);
        y[2] = W4_FMA(k4, s2, s1);
        y[3] = W4_FMA(k8, d2, d1) + T[5][j];
        asm volatile("; output stage of N tile %0" ::"n"(TAG));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = y[i] + bias;
            if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);   // == v > 0 ? v : 0.2 v
            if (col_ok && oy + i < p.Ho && ox + j < p.Wo && !((B2F_WINO4_ABLATE & 32) && v[0] != 12345.678f))
                *reinterpret_cast<f32x4 *>(obase + (size_t)(i * p.Wo + j) * p.out_pix_stride) = v;
        }
    }
}

// Half of the output stage of one N tile: item = (tile, 4 channels), output columns jp and jp + 2 only (the even
// columns need only the sums M1+M2, M3+M4 of a row, the odd ones only the differences), eight 16-byte stores.
// Used when all eight waves share one N tile (single-N-tile blocks).
__device__ __forceinline__ void wino4_output_half(const float *X, const ConvLaunch &p, int idx, int o_jp, int co_base, int oy0, int ox0,
                                                  float *ob)
{
    const int o_t = idx >> 3, o_cq = idx & 7;
    const int xr = o_t ^ ((o_t >> 2) & 1);
    const int co0 = co_base + 4 * o_cq;
    const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + co0);
    const float *xa = X + xr * 32 + 4 * o_cq;
    const float *xe = xa + (o_jp ? 5 : 0) * 1024;            // M5 for the odd columns, M0 for the even ones
    const float sg = o_jp ? -1.f : 1.f;
    const f32x4 sg4 = {sg, sg, sg, sg};
    const int oy = oy0 + 4 * (o_t >> 3), ox = ox0 + 4 * (o_t & 7);
    float *obase = ob + (size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)(oy * p.Wo + ox + o_jp) * p.out_pix_stride + (co0 & 7);
    const bool col_ok = co0 < p.cout;
    const f32x4 k2 = {2.f, 2.f, 2.f, 2.f}, k4 = {4.f, 4.f, 4.f, 4.f}, k8 = {8.f, 8.f, 8.f, 8.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        // column j = jp + 2q of T = M A:  j=0: M0 + s1 + s2,  j=1: d1 + 2 d2,  j=2: s1 + 4 s2,  j=3: d1 + 8 d2 + M5
        const float kq = q ? (o_jp ? 8.f : 4.f) : (o_jp ? 2.f : 1.f);
        const float ke = (q == 0) == (o_jp == 0) ? 1.f : 0.f;          // add M0 (j = 0) or M5 (j = 3)
        const f32x4 kq4 = {kq, kq, kq, kq}, ke4 = {ke, ke, ke, ke};
        f32x4 T[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const float *xr6 = xa + (6 * a) * 1024;
            const f32x4 m1 = *reinterpret_cast<const f32x4 *>(xr6 + 1 * 1024), m2 = *reinterpret_cast<const f32x4 *>(xr6 + 2 * 1024);
            const f32x4 m3 = *reinterpret_cast<const f32x4 *>(xr6 + 3 * 1024), m4 = *reinterpret_cast<const f32x4 *>(xr6 + 4 * 1024);
            const f32x4 me = *reinterpret_cast<const f32x4 *>(xe + (6 * a) * 1024);
            const f32x4 e1 = W4_FMA(sg4, m2, m1), e2 = W4_FMA(sg4, m4, m3);   // sums (even j) or differences (odd j)
            T[a] = W4_FMA(ke4, me, W4_FMA(kq4, e2, e1));
        }
        const f32x4 s1 = T[1] + T[2], d1 = T[1] - T[2], s2 = T[3] + T[4], d2 = T[3] - T[4];
        f32x4 y[4];
        y[0] = T[0] + s1 + s2;
        y[1] = W4_FMA(k2, d2, d1);
        y[2] = W4_FMA(k4, s2, s1);
        y[3] = W4_FMA(k8, d2, d1) + T[5];
        const int j = o_jp + 2 * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = y[i] + bias;
            if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);   // == v > 0 ? v : 0.2 v
            if (col_ok && oy + i < p.Ho && ox + j < p.Wo && !((B2F_WINO4_ABLATE & 32) && v[0] != 12345.678f))
                *reinterpret_cast<f32x4 *>(obase + (size_t)(i * p.Wo + 2 * q) * p.out_pix_stride) = v;
        }
    }
}

template <int NTV>
__global__ __launch_bounds__(512) void conv3x3_wino4(const ConvLaunch p)
{
    using namespace wino4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VSTRIDE = V_F4;
    f32x4 *Vb = reinterpret_cast<f32x4 *>(smem);                 // [2][V_F4]
    f32x4 *Rb = Vb + 2 * V_F4;                                    // [2][RAW_F4]

#if B2F_WINO_TRACE
    const long long t_start = clock64();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, half = lane >> 5;
    const int g = wave & 3, n = wave >> 2;

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    // 1-D grid, logical index = (image, tile row, tile column, n-block) with the n-block fastest, remapped so
    // that each XCD walks a contiguous range: the n-blocks of a tile and neighbouring tiles (which share the
    // raw patch resp. its halo) run on the same XCD at about the same time and find each other's lines in L2
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nb = bid % p.nblk + p.nb0;
    bid /= p.nblk;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;
    const int ox0 = tx_i * TW, oy0 = ty_i * TH;
    const int ix0 = ox0 - 1, iy0 = oy0 - 1;

    // ---- staging of the raw patch: item idx = tid + 512 i -> (pixel idx >> 1, k4 = tid & 1), 3 per thread.
    // Fixed over chunks: a 32-bit byte offset inside the (image, chunk) plane (both K segments have the
    // same pixel stride, checked by the launcher) and the LDS slot.  Items outside the image (zero padding)
    // or past the patch load offset 0 and never write LDS: padding slots are zeroed once here.
    unsigned s_off[3];
    int s_slot[3];
    bool s_ok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = tid + i * 512;
        const int pix = idx >> 1;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = iy0 + py, gx = ix0 + px;
        const bool in_patch = idx < NITEM;
        s_ok[i] = in_patch && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const unsigned gp = s_ok[i] ? (unsigned)(gy * p.W + gx) : 0u;
        s_off[i] = (gp * (unsigned)p.seg[0].pix_stride + (tid & 1) * 4) * 4u;
        s_slot[i] = (tid & 1) * 2 * PL2 + (in_patch ? py * RWP + colpos(px) : 0);    // f32x2 index of pair 0; pair 1 -> + PL2
        if (in_patch && !s_ok[i]) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            W4_RAW_STORE(Rb, s_slot[i], z4);
            W4_RAW_STORE(Rb + RAW_F4, s_slot[i], z4);
        }
    }

    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    const char *wsrc = reinterpret_cast<const char *>(p.wpk) + (size_t)nb * nchunks * U_F4 * 16;

    f32x4 sr[3];
    // buffer loads (scalar 128-bit resource per K segment based at this image, scalar chunk offset, one 32-bit
    // lane offset): no address arithmetic on the VALU, one address VGPR per load
    const __amdgpu_buffer_rsrc_t r_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.seg[0].ptr + (size_t)img * p.seg[0].img_stride), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.seg[1].ptr + (size_t)img * p.seg[1].img_stride), 0, 0x7fffffff, 0x00020000);
#define W4_LOAD_RAW(c_)                                                                             \
    do {                                                                                            \
        const int c__ = (c_);                                                                       \
        const bool s1 = c__ >= p.seg[0].nchunks;                                                    \
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                       \
        const int cc = s1 ? c__ - p.seg[0].nchunks : c__;                                           \
        const int so = (int)(cc * cstr * 4);                                                        \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                               \
            sr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_rsrc1 : r_rsrc0, (int)s_off[i], so, 0)); \
    } while (0)
#define W4_WRITE_RAW(buf_)                                                                          \
    do {                                                                                            \
        f32x4 *r = Rb + (buf_) * RAW_F4;                                                            \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) if (s_ok[i]) W4_RAW_STORE(r, s_slot[i], sr[i]); \
    } while (0)

#define W4_MAYBE_CONST const
    // ---- input transform V = B^T d B of a chunk: thread = (tile t = lane & 31, k4 = lane >> 5, channel pair th = wave & 1 of
    // the k4 group's four), wave >> 1 selects the rows of B^T d it produces, with the common sub-expressions of the row pairs:
    //   waves 0, 1: rows 1, 2   P = d4 - 4 d2, Q = d3 - 4 d1, r1 = P + Q,  r2 = P - Q
    //   waves 2, 3: rows 3, 4   P = d4 -   d2, Q = d3 -   d1, r3 = P + 2Q, r4 = P - 2Q
    //   waves 4, 5: row 0       P = d4 - 5 d2, Q = d0,        r0 = P + 4Q
    //   waves 6, 7: row 5       P = d5 - 5 d3, Q = d1,        r5 = P + 4Q
    // i.e. ONE instruction stream  P = fma(ca, x1, x2), Q = fma(cb, x3, x4), rA = fma(cs, Q, P), rB = fma(-cs, Q, P)  with
    // wave-uniform rows x1..x4 and coefficients (four packed ops per tile column for two rows, where one row at a time takes
    // four each), then the 6-point column pass of each produced row (12 packed ops; the single-row waves skip the second).
    // The waves w and w + 4 of a SIMD carry 48 + 36 packed ops per chunk where the row-per-wave form had 2 x 72 (rows 0..5 on
    // waves 0..5, waves 6, 7 redoing row 5): fp32 MFMAs and VALU share the issue pipe, so this is time (ablating 44 % of the old
    // transform's VALU: -4 % on the 200 -> 128 layer, profiles/r03_wino4_notes.txt).
    const int t_role = wave >> 1;
    const int th = wave & 1;                                     // channel pair of the float4: x, y | z, w
    int rx1, rx2, rx3, rx4, t_orow;
    float t_ca, t_cb, t_cs;
    switch (t_role) {
    case 0: rx1 = 2; rx2 = 4; rx3 = 1; rx4 = 3; t_ca = -4.f; t_cb = -4.f; t_cs = 1.f; t_orow = 1; break;
    case 1: rx1 = 2; rx2 = 4; rx3 = 1; rx4 = 3; t_ca = -1.f; t_cb = -1.f; t_cs = 2.f; t_orow = 3; break;
    case 2: rx1 = 2; rx2 = 4; rx3 = 0; rx4 = 0; t_ca = -5.f; t_cb = 0.f; t_cs = 4.f; t_orow = 0; break;
    default: rx1 = 3; rx2 = 5; rx3 = 1; rx4 = 1; t_ca = -5.f; t_cb = 0.f; t_cs = 4.f; t_orow = 5; break;
    }
    const bool t_two = t_role < 2;                               // two output rows (t_orow, t_orow + 1)
    const int t_tile = lane & 31, t_k4 = lane >> 5;
    const int t_base = (t_k4 * 2 + th) * PL2 + (4 * (t_tile >> 3)) * RWP + (t_tile & 7);   // f32x2 index in this thread's pair plane
    W4_MAYBE_CONST int t_row[4] = {t_base + rx1 * RWP, t_base + rx2 * RWP, t_base + rx3 * RWP, t_base + rx4 * RWP};
    W4_MAYBE_CONST int t_dst = ((t_orow * 6 * 2 + t_k4) * 2 + th) * 32 + t_tile;   // f32x2 index of V[xi = 6 t_orow][k4][pair th][t]; xi+1 -> +128, next row -> +768
    f32x4 R[6], d[NTV == 1 ? 4 : 2];   // R[j] = (row A of column j | row B); d = (x1 | x2), (x3 | x4) of a column
    // slice s < 6: column s of the 6-wide tile window (reads of its four rows, then the four packed ops); slice 6 / 7: column
    // pass + LDS writes of the first / second produced row into V buffer vbuf_
#define W4_T_READ(s_, rbuf_) W4_T_READ_D(s_, rbuf_, 0)
#define W4_T_READ_D(s_, rbuf_, db_)                                                                 \
    do {                                                                                            \
        if ((s_) < 6 && !(B2F_WINO4_ABLATE & 256)) {                                                \
            const f32x2 *rp = reinterpret_cast<const f32x2 *>(Rb + (rbuf_) * RAW_F4) + colpos(s_);  \
            const f32x2 x1 = rp[t_row[0]], x2 = rp[t_row[1]], x3 = rp[t_row[2]], x4 = rp[t_row[3]]; \
            d[(db_)] = __builtin_shufflevector(x1, x2, 0, 1, 2, 3);                                 \
            d[(db_) + 1] = __builtin_shufflevector(x3, x4, 0, 1, 2, 3);                             \
        }                                                                                           \
    } while (0)
    // Written as v_pk_* inline asm: hipcc unpacks packed fp32 ops that follow an MFMA into two scalar ones (it
    // assumes they co-issue with the MFMA; behind an fp32 MFMA they do not), and VALU instructions are what
    // this loop pays for.  The asm also pins the slice here.
#define W4_T_FMA(s_, vbuf_) W4_T_FMA_D(s_, 0, vbuf_)
#define W4_T_FMA_D(s_, db_, vbuf_)                                                                  \
    do {                                                                                            \
        if ((s_) < 6) {                                                                             \
            const f32x2 ca2 = {t_ca, t_ca}, cb2 = {t_cb, t_cb}, cs2 = {t_cs, t_cs}, cn2 = {-t_cs, -t_cs}; \
            const f32x2 x1 = __builtin_shufflevector(d[(db_)], d[(db_)], 0, 1), x2 = __builtin_shufflevector(d[(db_)], d[(db_)], 2, 3); \
            const f32x2 x3 = __builtin_shufflevector(d[(db_) + 1], d[(db_) + 1], 0, 1), x4 = __builtin_shufflevector(d[(db_) + 1], d[(db_) + 1], 2, 3); \
            f32x2 P, Q, oa, ob;                                                                     \
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(P) : "s"(ca2), "v"(x1), "v"(x2));     \
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(Q) : "s"(cb2), "v"(x3), "v"(x4));     \
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(oa) : "s"(cs2), "v"(Q), "v"(P));      \
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(ob) : "s"(cn2), "v"(Q), "v"(P));      \
            R[(s_) < 6 ? (s_) : 0] = __builtin_shufflevector(oa, ob, 0, 1, 2, 3);                   \
        } else if ((s_) == 6) {                                                                     \
            W4_T_COLPASS(0, (vbuf_));                                                               \
        } else if (t_two) {                                                                         \
            W4_T_COLPASS(1, (vbuf_));                                                               \
        }                                                                                           \
    } while (0)
    // column pass of produced row hh_ (0: first, 1: second): V[a][.] = R B (12 packed ops), six 8-byte LDS writes
#define W4_T_COLPASS(hh_, vbuf_)                                                                    \
    do {                                                                                            \
        const f32x2 k4v = {4.f, 4.f}, k5v = {-5.f, -5.f}, km4 = {-4.f, -4.f}, k2v = {2.f, 2.f}, km2 = {-2.f, -2.f}; \
        f32x2 r0 = (hh_) ? __builtin_shufflevector(R[0], R[0], 2, 3) : __builtin_shufflevector(R[0], R[0], 0, 1); \
        f32x2 r1 = (hh_) ? __builtin_shufflevector(R[1], R[1], 2, 3) : __builtin_shufflevector(R[1], R[1], 0, 1); \
        f32x2 r2 = (hh_) ? __builtin_shufflevector(R[2], R[2], 2, 3) : __builtin_shufflevector(R[2], R[2], 0, 1); \
        f32x2 r3 = (hh_) ? __builtin_shufflevector(R[3], R[3], 2, 3) : __builtin_shufflevector(R[3], R[3], 0, 1); \
        f32x2 r4 = (hh_) ? __builtin_shufflevector(R[4], R[4], 2, 3) : __builtin_shufflevector(R[4], R[4], 0, 1); \
        f32x2 r5 = (hh_) ? __builtin_shufflevector(R[5], R[5], 2, 3) : __builtin_shufflevector(R[5], R[5], 0, 1); \
        f32x2 t0, t1, pq, qq, uu, vv, o1, o2, o3, o4;                                               \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "s"(k4v), "v"(r0), "v"(r4));        \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t0) : "s"(k5v), "v"(r2));                 \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pq) : "s"(km4), "v"(r2), "v"(r4));        \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(qq) : "s"(km4), "v"(r1), "v"(r3));        \
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(uu) : "v"(r4), "v"(r2)); \
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(vv) : "v"(r3), "v"(r1)); \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "s"(k4v), "v"(r1), "v"(r5));        \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t1) : "s"(k5v), "v"(r3));                 \
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(o1) : "v"(pq), "v"(qq));                      \
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o2) : "v"(pq), "v"(qq)); \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o3) : "s"(k2v), "v"(vv), "v"(uu));        \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o4) : "s"(km2), "v"(vv), "v"(uu));        \
        f32x2 *v = reinterpret_cast<f32x2 *>(Vb + (vbuf_) * VSTRIDE) + t_dst + 768 * (hh_);         \
        if (!(B2F_WINO4_ABLATE & 512)) { v[0] = t0; v[128] = o1; v[2 * 128] = o2; v[3 * 128] = o3; v[4 * 128] = o4; v[5 * 128] = t1; } \
        else asm volatile("" :: "v"(t0), "v"(o1), "v"(o2), "v"(o3), "v"(o4), "v"(t1));              \
    } while (0)
#define W4_T_COLS(vbuf_) do {} while (0)      /* (the column passes are slices 6, 7 of W4_T_FMA now) */

    if constexpr (NTV == 2) {
    f32x16 acc[9];
#pragma unroll
    for (int x = 0; x < 9; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    const bool mf_on = n < NTV;                                          // this wave's N tile holds real channels
    const int a_off = (9 * g * 2 + half) * 2 * 32 + m;                   // f32x2 index of V[xi = 9g][k4 = half][pair 0][tile m]; pair 1 -> +32, xi+1 -> +128
    const unsigned b_off = ((9 * g * 2 + half) * 64 + n * 32 + m) * 16u; // bytes: U[xi = 9g][k4 = half][co]; xi+1 -> +2048
    f32x4 av[3], bv[6];
    // buffer loads: 128-bit resource (scalar), one 32-bit lane offset, scalar (chunk, xi) offset -- no per-load
    // address arithmetic on the VALU and one address VGPR instead of two
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wsrc), 0, 0x7fffffff, 0x00020000);
#define W4_LOAD_U(slot_, c_, x_)                                                                    \
    bv[slot_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)b_off, (int)((c_) * (U_F4 * 16) + (x_) * 2048), 0))

    // Software pipeline (Tr(k) = input transform of chunk k -> V[k & 1], raw(k) = its patch in raw buffer k & 1):
    //   iteration c, xi steps 0..5 : MFMAs of xi (A operand read one step ahead, B two steps ahead),
    //                                slices 2..7 of Tr(c+1)
    //                xi step  6    : MFMAs, column pass of Tr(c+1) -> V[(c+1) & 1], raw(c+2) -> LDS, A operands of
    //                                xi 7, 8 fetched, BARRIER
    //                xi steps 7, 8 : MFMAs, A operands of xi 0, 1 of chunk c+1, slices 0, 1 of Tr(c+2); global loads
    //                                of raw(c+3)
    // so nothing that follows the barrier depends on LDS data written just before it.
    // ---- prologue: raw(0), raw(1) -> LDS; Tr(0) -> V[0]; slices 0, 1 of Tr(1); raw(2) in flight; B of xi 0, 1 ----
    {
        f32x4 keep[3];
        W4_LOAD_RAW(min(1, nchunks - 1));
#pragma unroll
        for (int i = 0; i < 3; ++i) keep[i] = sr[i];
        W4_LOAD_RAW(0);
        if (NTV == 2 || mf_on) { W4_LOAD_U(0, 0, 0); W4_LOAD_U(1, 0, 1); W4_LOAD_U(2, 0, 2); W4_LOAD_U(3, 0, 3); W4_LOAD_U(4, 0, 4); }
        W4_WRITE_RAW(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) sr[i] = keep[i];
        W4_WRITE_RAW(1);
        W4_LOAD_RAW(min(2, nchunks - 1));
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) { W4_T_READ(s, 0); W4_T_FMA(s, 0); }
    W4_T_READ(0, 1); W4_T_FMA(0, 0);
    W4_T_READ(1, 1); W4_T_FMA(1, 0);
    W4_T_READ(2, 1);
    __syncthreads();
    av[0] = W4_A_READ(reinterpret_cast<const f32x2 *>(Vb) + a_off);
    av[1] = W4_A_READ(reinterpret_cast<const f32x2 *>(Vb) + a_off + 128);

#if B2F_WINO_TRACE
    const int tr_slot = blockIdx.x == 300 ? 0 : blockIdx.x == 301 ? 1 : blockIdx.x == 1200 ? 2 : blockIdx.x == 1456 ? 3 : -1;
    const bool tr_on = p.trace && tr_slot >= 0;
    long long *tr_buf = p.trace + (tr_on ? (tr_slot * 8 + wave) * 160 : 0);
#endif
#if B2F_WINO_TRACE
    if (tr_on && lane == 0) { tr_buf[150] = t_start; tr_buf[151] = clock64(); tr_buf[155] = wall_clock64(); }
#endif
    // One chunk of the software pipeline as a macro with the phase PH_ of the B ring as a compile-time
    // parameter: the ring has 6 slots and runs 5 xi ahead (9 xi per chunk, so the slot pattern repeats every
    // two chunks: the loop below is unrolled by two, an odd last chunk reuses phase 0).  Five xi of lookahead
    // = ~2500 cycles between a B load and its use: the B loads queued behind the raw-patch loads (HBM latency,
    // s_waitcnt vmcnt is in order) no longer stall the MFMAs of both waves of a SIMD at the start of a chunk.
#define W4_CHUNK(PH_, c_)                                                                           \
    do {                                                                                            \
        const int c = (c_);                                                                         \
        W4_T(0);                                                                                    \
        const int cn = min(c + 1, nchunks - 1);                                                     \
        const f32x2 *Vc = reinterpret_cast<const f32x2 *>(Vb + (c & 1) * V_F4) + a_off;             \
        const f32x2 *Vn = reinterpret_cast<const f32x2 *>(Vb + ((c + 1) & 1) * V_F4) + a_off;       \
        _Pragma("unroll") for (int x = 0; x < 9; ++x) {                                             \
            /* B operand five xi ahead (next chunk's for x >= 4); A operand one xi ahead (xi 8 two ahead, so */ \
            /* that every read of V[c & 1] is issued before the barrier; xi 0, 1 of the next chunk after it) */ \
            if (!(B2F_WINO4_ABLATE & 4) && (NTV == 2 || mf_on)) {                                   \
                if (x + 5 < 9) W4_LOAD_U((9 * (PH_) + x + 5) % 6, c, x + 5);                        \
                else W4_LOAD_U((9 * (PH_) + x + 5) % 6, cn, x + 5 - 9);                             \
            }                                                                                       \
            if (!(B2F_WINO4_ABLATE & 16)) {                                                         \
                if (x >= 1 && x <= 6) av[(x + 1) % 3] = W4_A_READ(Vc + (x + 1) * 128);              \
                if (x == 6) av[8 % 3] = W4_A_READ(Vc + 8 * 128);                                    \
                if (x == 7) av[0] = W4_A_READ(Vn);                                                  \
                if (x == 8) av[1] = W4_A_READ(Vn + 128);                                            \
            }                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if (!(B2F_WINO4_ABLATE & 8) && (NTV == 2 || mf_on)) {                                   \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
                    acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x % 3][j], bv[(9 * (PH_) + x) % 6][j], acc[x], 0, 0, 0); \
            }                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if (x < 6) {                                                                            \
                /* slice x + 2 of Tr(c+1) (its reads were issued one step ago), then the reads of the next slice */ \
                if (!(B2F_WINO4_ABLATE & 1)) {                                                      \
                    W4_T_FMA(x + 2, (c + 1) & 1);                                                   \
                    if (x + 3 < 8) W4_T_READ(x + 3, (c + 1) & 1);                                   \
                }                                                                                   \
            } else if (x == 6) {                                                                    \
                if (!(B2F_WINO4_ABLATE & 1)) W4_T_COLS((c + 1) & 1);                                \
                if (!(B2F_WINO4_ABLATE & 2)) W4_WRITE_RAW(c & 1);                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4_T(3);                                                                            \
                __syncthreads();                                                                    \
                W4_T(4);                                                                            \
                if (!(B2F_WINO4_ABLATE & 1)) W4_T_READ(0, c & 1);   /* Tr(c+2), raw(c+2) is in raw buffer c & 1 */ \
                if (!(B2F_WINO4_ABLATE & 2)) W4_LOAD_RAW(min(c + 3, nchunks - 1));                  \
            } else if (x == 7) {                                                                    \
                if (!(B2F_WINO4_ABLATE & 1)) { W4_T_FMA(0, 0); W4_T_READ(1, c & 1); }               \
            } else {                                                                                \
                if (!(B2F_WINO4_ABLATE & 1)) { W4_T_FMA(1, 0); W4_T_READ(2, c & 1); }               \
            }                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if (x == 2) W4_T(1);                                                                    \
            if (x == 5) W4_T(2);                                                                    \
        }                                                                                           \
    } while (0)
    {
        int c2 = 0;
        for (; c2 + 1 < nchunks; c2 += 2) {
            W4_CHUNK(0, c2);
            W4_CHUNK(1, c2 + 1);
        }
        if (c2 < nchunks) W4_CHUNK(0, c2);
    }
#undef W4_CHUNK

    // ---- output, one N tile at a time (LDS holds the 36 planes of 32 channels: 144 KB): the four waves of
    // the tile write their accumulators to X[xi][tile row][co 32] (tile rows t and t ^ 1 swapped when bit 2
    // of t is set: the two lane halves of a ds_write then hit different bank halves); the output stage of
    // BOTH tiles is run by the waves of tile 0, whose accumulators are dead by then (in the code path of the
    // tile-1 waves the register allocator keeps the 144 accumulator registers reserved and spills the output
    // stage, and a spill reload there waits for the stores in flight: vmcnt counts stores too).  The two
    // groups run through separate code paths with the same three barriers ----
#if B2F_WINO_TRACE
    if (tr_on && lane == 0) tr_buf[152] = clock64();
#endif
    float *X = reinterpret_cast<float *>(smem);
    float *ob = p.out + (size_t)img * p.out_img_stride;
    // 72 ds_write2st64_b32 per thread instead of 144 ds_write_b32 (b32 LDS writes run at 64 B/clk): rows r and
    // r + 4 of an accumulator land 8 tiles = 4 x 64 dwords apart, the xi planes 16 x 64 dwords apart, so one
    // base address per (r & 3) covers the whole dump through the two 8-bit offsets.  Inline asm (the compiler
    // pairs only a few of them): the barrier that follows must wait for lgkmcnt itself (W4_LDS_BARRIER).
    unsigned dump_base[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) dump_base[e] = 4u * (unsigned)((9 * g * 32 + ((e + 4 * half) ^ half)) * 32 + m);
#define W4_DUMP_ACC(tag_)                                                                           \
    asm volatile("; accumulators of N tile %0 -> LDS" ::"n"(tag_));                                 \
    _Pragma("unroll") for (int x = 0; x < 9; ++x)                                                   \
        _Pragma("unroll") for (int qq = 0; qq < 4; qq += 2)                                         \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                           \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4"                   \
                             :: "v"(dump_base[e]), "v"(acc[x][4 * qq + e]), "v"(acc[x][4 * (qq + 1) + e]), \
                                "n"(x * 16 + qq * 4), "n"(x * 16 + qq * 4 + 4) : "memory")
    // W4_LDS_BARRIER: barrier that orders LDS traffic only.  __syncthreads() is also a release fence: after
    // the output stage it would hold the barrier until every global store of the tile has been acknowledged
    // (measured: ~25000 cycles per tile), although the next tile only needs the LDS reads to be over.
#define W4_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#if B2F_WINO_TRACE
#define W4_E(k_) do { if (tr_on && lane == 0) tr_buf[140 + (k_)] = clock64(); } while (0)
#else
#define W4_E(k_) do {} while (0)
#endif
    if (n == 0) {
        W4_DUMP_ACC(0);
        W4_E(0);
        W4_LDS_BARRIER();
        W4_E(1);
        wino4_output_tile<0>(X, p, tid & 255, nb * 64, oy0, ox0, ob);
        W4_E(2);
        if (NTV == 2) {
            W4_LDS_BARRIER();          // X free again
            W4_LDS_BARRIER();          // accumulators of N tile 1 are in X
            W4_E(3);
            wino4_output_tile<1>(X, p, tid & 255, nb * 64 + 32, oy0, ox0, ob);
            W4_E(4);
        }
    } else if (NTV == 2) {
        W4_LDS_BARRIER();
        W4_LDS_BARRIER();
        W4_E(0);
        W4_DUMP_ACC(1);
        W4_E(1);
        W4_LDS_BARRIER();
    } else {
        __syncthreads();
    }
#undef W4_E
#undef W4_LDS_BARRIER
#undef W4_DUMP_ACC
#if B2F_WINO_TRACE
    if (tr_on && lane == 0) { tr_buf[153] = clock64(); tr_buf[156] = wall_clock64(); }
#endif
    } else {
    // =====================================================================================================
    // Single N tile (NTV == 1: the last n-block of a layer whose cout is 32 mod 64): the 36 xi are split over
    // ALL eight waves, wave (g, n) takes xi(x) = 9g + 5n + x, x = 0..4 (the fifth step of the n = 1 waves is a
    // duplicate that is never dumped), so every wave keeps multiplying -- with one N tile per wave pair half of
    // the waves had no MFMA work at all.  Only 5 accumulators: registers are plentiful, so the pipeline is the
    // simple one (B operands a whole chunk ahead, A operands read after the barrier, two transform slices per
    // xi step, one barrier at the end of the chunk).
    // =====================================================================================================
    f32x16 acc[5];
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
    int xo[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) xo[x] = min(9 * g + 5 * n + x, 35);
    const int a_lane = half * 2 * 32 + m;                   // f32x2 index: V[xi][k4 = half][pair 0][tile m] = xi * 128 + a_lane; pair 1 -> + 32
    const unsigned b_lane = (half * 64 + m) * 16u;          // bytes: U[xi][k4 = half][co m] = xi * 2048 + b_lane
    f32x4 av[5], bc[5], bn[5];
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wsrc), 0, 0x7fffffff, 0x00020000);
#define W4S_LOAD_U(dst_, c_)                                                                        \
    _Pragma("unroll") for (int x = 0; x < 5; ++x)                                                   \
        dst_[x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)b_lane, (int)((c_) * (U_F4 * 16) + xo[x] * 2048), 0))
    {
        f32x4 keep[3];
        W4_LOAD_RAW(min(1, nchunks - 1));
#pragma unroll
        for (int i = 0; i < 3; ++i) keep[i] = sr[i];
        W4_LOAD_RAW(0);
        W4S_LOAD_U(bc, 0);
        W4_WRITE_RAW(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) sr[i] = keep[i];
        W4_WRITE_RAW(1);
    }
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) { W4_T_READ(s2, 0); W4_T_FMA(s2, 0); }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int cn = min(c + 1, nchunks - 1);
        W4S_LOAD_U(bn, cn);
        W4_LOAD_RAW(min(c + 2, nchunks - 1));
        const f32x2 *Vc = reinterpret_cast<const f32x2 *>(Vb + (c & 1) * V_F4) + a_lane;
#pragma unroll
        for (int x = 0; x < 5; ++x) av[x] = W4_A_READ(Vc + xo[x] * 128);
#pragma unroll
        for (int x = 0; x < 5; ++x) {
            if (x < 4) { W4_T_READ_D(2 * x, (c + 1) & 1, 0); W4_T_READ_D(2 * x + 1, (c + 1) & 1, 2); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x < 4 || n == 0)       // the fifth step of the n = 1 waves would be a duplicate (only MFMAs sit behind this branch)
                    acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][j], bc[x][j], acc[x], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (x < 4) { W4_T_FMA_D(2 * x, 0, (c + 1) & 1); W4_T_FMA_D(2 * x + 1, 2, (c + 1) & 1); }
            else { W4_WRITE_RAW(c & 1); }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 5; ++x) bc[x] = bn[x];
    }
#undef W4S_LOAD_U
    // output: all eight waves dump their xi planes, then every thread takes one half item
    {
        float *X = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int x = 0; x < 5; ++x)
            if (x < 4 || n == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = ((r & 3) + 8 * (r >> 2) + 4 * half) ^ half;
                    X[(xo[x] * 32 + t) * 32 + m] = acc[x][r];
                }
            }
        __syncthreads();
        wino4_output_half(X, p, tid & 255, tid >> 8, nb * 64, oy0, ox0, p.out + (size_t)img * p.out_img_stride);
    }
    }
#undef W4_LOAD_RAW
#undef W4_WRITE_RAW
#undef W4_LOAD_U
}


// =========================================================================================================
// Persistent form of the two-N-tile kernel (round 2): one block per CU walks its tiles (virtual block index
// v = blockIdx.x + k * gridDim.x, same logical order as the one-tile-per-block launch) and the software pipeline
// of the K loop simply keeps running across tile boundaries -- the raw patch of the next tile's first three
// chunks is loaded, staged and transformed by the last iterations of the current tile, so a tile has no prologue
// (11 500 cycles of HBM latency + first transform per block in the one-tile form, which a single resident block
// per CU cannot hide).  For that the output exchange must live beside the pipeline's buffers: LDS is laid out
// [raw 0 | raw 1 | V 0 | gap | V 1] with |V 0 + gap| = |gap + V 1| = one TILE ROW (8 tiles x 64 channels x 36 xi,
// 72 KB) of accumulators, the dead V buffer of the last chunk + the gap are the exchange buffer, and the output
// stage runs in four passes (tile rows) in which ALL eight waves dump four accumulator registers per xi and then
// transform: item = (tile, 4 channels, output column j), lanes ordered (tile column, j, channel half) so that a
// wave's 16-byte stores cover 1 KB of contiguous memory (whole lines; the one-tile form writes 32-byte pieces).
// Zero padding comes from the buffer loads' range check (offset >= num_records reads 0), so the staging state of
// a tile is three byte offsets per thread and no LDS slot is ever "pre-zeroed".
// =========================================================================================================
// HYB: bit x set = xi step x of every wave runs on the bf16 pipe with split operands (NTV == 2 only; needs p.wpk_split)
template <int NTV, int HYB = 0>
__global__ __launch_bounds__(512) void conv3x3_wino4p(const ConvLaunch p)
{
    using namespace wino4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int VSTRIDE = XQ_F4;                               // V 1 starts one exchange buffer after V 0
    f32x4 *Rb = reinterpret_cast<f32x4 *>(smem);                 // [2][RAW_F4]
    f32x4 *Vb = Rb + 2 * RAW_F4;                                  // V 0 | gap | V 1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, half = lane >> 5;
    const int g = wave & 3, n = wave >> 2;

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int total = tiles_x * tiles_y * p.nimg * p.nblk;
    const int G = gridDim.x;
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);

    // ---- staging of the raw patch: thread tid < 408 = (patch row r6 < 6, patch column px < 34, k4 = tid & 1) stages the
    // three pixels (r6 + 6 i, px), i = 0..2, of the 18 x 34 patch: ONE LDS slot and ONE byte offset per thread (item i
    // adds an immediate to the slot and a scalar to the offset).  Nothing a thread holds depends on the tile: the tile
    // enters through the base address of the buffer resource (scalar) and through three 64-bit lane masks (scalar: the
    // lanes whose pixel lies inside the image; the others load at offset -16, which the range check of the buffer load
    // turns into zeros -- the zero padding of the convolution)
    constexpr int NSTG = 2 * 6 * PW;                             // 408 staging threads
    const bool s_act = tid < NSTG;
    int s_slot;
    unsigned l_off;
    {
        const int pix = min(tid, NSTG - 1) >> 1;
        const int r6 = pix / PW, px = pix - r6 * PW;
        s_slot = (tid & 1) * 2 * PL2 + r6 * RWP + colpos(px);     // f32x2 index of pair 0; pair 1 -> + PL2
        l_off = ((unsigned)(r6 * p.W + px) * (unsigned)p.seg[0].pix_stride + (tid & 1) * 4) * 4u;
    }
    const int rowblk = 6 * p.W * p.seg[0].pix_stride * 4;        // bytes between the items of a thread
    typedef unsigned long long u64;
    u64 mk[3], mk_n[3];                                           // load side's tile / the block's next tile
    __amdgpu_buffer_rsrc_t r_rsrc0, r_rsrc1;
    int cur_nb, cur_img, cur_ox0, cur_oy0;                        // tile being computed / stored
    int nxt_nb, nxt_img, nxt_ox0, nxt_oy0;                        // the block's next tile (decoded at the start of a tile)
    bool has_next;
    int lc = 0;                                                   // load side of the pipeline: next chunk of its tile
#define W4P_DECODE(v_, nb_, img_, ox0_, oy0_)                                                       \
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
#define W4P_MASKS(ox0_, oy0_, out_)                                                                 \
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
    // buffer resources based at the patch origin (oy0 - 1, ox0 - 1) of a tile (may lie before the tensor: never dereferenced there)
#define W4P_RSRC(img_, ox0_, oy0_)                                                                  \
    do {                                                                                            \
        const long long o__ = ((long long)((oy0_) - 1) * p.W + ((ox0_) - 1)) * p.seg[0].pix_stride; \
        r_rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[0].ptr) + ((long long)(img_) * p.seg[0].img_stride + o__), 0, 0x7fffffff, 0x00020000); \
        r_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.seg[1].ptr) + ((long long)(img_) * p.seg[1].img_stride + o__), 0, 0x7fffffff, 0x00020000); \
    } while (0)
    f32x4 sr[3];
    // next item of the load stream -> sr; after a tile's last chunk the stream moves on to the block's next tile (and
    // keeps re-reading the very last chunk when there is none: harmless).  Scalar work only at the switch.
#define W4P_LOAD_STREAM() W4P_LOAD_STREAM_TO(sr)
#define W4P_LOAD_STREAM_TO(sr_)                                                                     \
    do {                                                                                            \
        const bool s1 = lc >= p.seg[0].nchunks;                                                     \
        const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                       \
        const int cc = s1 ? lc - p.seg[0].nchunks : lc;                                             \
        const int so = (int)(cc * cstr * 4);                                                        \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                             \
            unsigned vo__;                                                                          \
            asm("v_cndmask_b32_e64 %0, -16, %1, %2" : "=v"(vo__) : "v"(l_off), "s"(mk[i]));         \
            sr_[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s1 ? r_rsrc1 : r_rsrc0, (int)vo__, so + i * rowblk, 0)); \
        }                                                                                           \
        if (++lc == nchunks) {                                                                      \
            if (has_next) {                                                                         \
                lc = 0;                                                                             \
                mk[0] = mk_n[0]; mk[1] = mk_n[1]; mk[2] = mk_n[2];                                  \
                W4P_RSRC(nxt_img, nxt_ox0, nxt_oy0);                                                \
            } else {                                                                                \
                lc = nchunks - 1;                                                                   \
            }                                                                                       \
        }                                                                                           \
    } while (0)
#define W4P_WRITE_RAW(buf_) W4P_WRITE_RAW_FROM(buf_, sr)
#define W4P_WRITE_RAW_FROM(buf_, sr_)                                                               \
    do {                                                                                            \
        f32x4 *r = Rb + (buf_) * RAW_F4;                                                            \
        if (s_act) { W4_RAW_STORE(r, s_slot, sr_[0]); W4_RAW_STORE(r, s_slot + 6 * RWP, sr_[1]); W4_RAW_STORE(r, s_slot + 12 * RWP, sr_[2]); } \
    } while (0)

    // ---- input transform role (as in conv3x3_wino4: row pairs with shared sub-expressions, channel pair th = wave & 1) ----
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
    const int t_tile = lane & 31, t_k4 = lane >> 5;
    const int t_base = (t_k4 * 2 + th) * PL2 + (4 * (t_tile >> 3)) * RWP + (t_tile & 7);
    int t_row[4] = {t_base + rx1 * RWP, t_base + rx2 * RWP, t_base + rx3 * RWP, t_base + rx4 * RWP};
    int t_dst = ((t_orow * 6 * 2 + t_k4) * 2 + th) * 32 + t_tile;
    f32x4 R[6], d[NTV == 1 ? 4 : 2];

#if B2F_WINO_TRACE
    const int trp_slot = blockIdx.x == 40 ? 0 : blockIdx.x == 41 ? 1 : -1;
    const bool trp_on = p.trace && trp_slot >= 0 && (wave == 0 || wave == 4) && lane == 0;
    long long *trp_buf = p.trace + (trp_on ? (trp_slot * 2 + (wave >> 2)) * 160 : 0);
    int trp_tile = 0;
#define W4P_T(k_) do { if (trp_on && trp_tile < 12) trp_buf[trp_tile * 12 + (k_)] = clock64(); } while (0)
    // step-level stamps of tile 3, chunks 8..11 (4 per xi step: step start | before the multiplications | after them | after the
    // transform slice), kept in the spare LDS behind the kernel's buffers and copied out after the tile
    long long *ts_lds = reinterpret_cast<long long *>(smem + P_LDS_BYTES) + (wave >> 2) * 144;
#define W4P_TS(c_, x_, k_) do { if (trp_on && trp_tile == 3 && (c_) >= 8 && (c_) < 12) ts_lds[(((c_) - 8) * 9 + (x_)) * 4 + (k_)] = clock64(); } while (0)
#else
#define W4P_T(k_) do {} while (0)
#define W4P_TS(c_, x_, k_) do {} while (0)
#endif
    if constexpr (NTV == 2) {
    f32x16 acc[9];
    int a_off = (9 * g * 2 + half) * 2 * 32 + m;
    unsigned b_off = ((9 * g * 2 + half) * 64 + n * 32 + m) * 16u;
    f32x4 av[3], bv[6];
    __amdgpu_buffer_rsrc_t w_rsrc, ws_rsrc;
    u32x2 bl[6];                                                  // third term of the split B operand (hybrid steps only)
#define W4P_HYB(x_) ((HYB >> (x_)) & 1)
#define W4P_LOAD_U(slot_, c_, x_)                                                                   \
    do {                                                                                            \
        if (W4P_HYB(x_)) {                                                                          \
            bv[slot_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ws_rsrc, (int)b_off, (int)((c_) * USC_BYTES + (x_) * 2048), 0)); \
            bl[slot_] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(ws_rsrc, (int)(b_off >> 1), (int)((c_) * USC_BYTES + US4_BYTES + (x_) * 1024), 0)); \
        } else {                                                                                    \
            bv[slot_] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)b_off, (int)((B2F_W4_BHOT ? 0 : (c_)) * (U_F4 * 16) + (B2F_W4_BHOT == 2 ? 0 : (x_)) * 2048), 0)); \
        }                                                                                           \
    } while (0)
#define W4P_RSRC_U(nb_)                                                                             \
    do {                                                                                            \
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc(                                                 \
            const_cast<char *>(reinterpret_cast<const char *>(p.wpk) + (size_t)(nb_) * nchunks * U_F4 * 16), 0, 0x7fffffff, 0x00020000); \
        if (HYB) ws_rsrc = __builtin_amdgcn_make_buffer_rsrc(                                       \
            const_cast<char *>(reinterpret_cast<const char *>(p.wpk_split) + (size_t)(nb_) * nchunks * USC_BYTES), 0, 0x7fffffff, 0x00020000); \
    } while (0)
    // the multiplications of xi step x_: four fp32 MFMAs, or (hybrid step) the split of the A operand and three bf16 MFMAs
    // through one accumulator, strictly back to back
#ifndef B2F_W4_BHOT
#define B2F_W4_BHOT 0      // TIMING EXPERIMENT (wrong results): 1 = every chunk reads chunk 0's weights (L2-resident), 2 = every step the same 2 KB (L1-resident); data stays random
#endif
#ifndef B2F_W4_MFMA16_TIMING
#define B2F_W4_MFMA16_TIMING 0
#endif
    // one K pair of xi step x_: a v_mfma_f32_32x32x2_f32 -- or, TIMING EXPERIMENT (wrong results), the same MACs as two
    // v_mfma_f32_16x16x4_f32 on two quarters of the accumulator (tools/mfma_f32_power.hip: that instruction sustains 155 TFLOP/s at 2.39 GHz
    // where the 32x32x2 form is clocked down to 2.17 - 2.27 GHz: half the accumulator traffic per MAC)
#if B2F_W4_MFMA16_TIMING
#define W4P_MF(x_, j_, b_)                                                                          \
    do {                                                                                            \
        f32x4 q0__ = __builtin_shufflevector(acc[x_], acc[x_], 4 * ((j_) & 1), 4 * ((j_) & 1) + 1, 4 * ((j_) & 1) + 2, 4 * ((j_) & 1) + 3); \
        f32x4 q1__ = __builtin_shufflevector(acc[x_], acc[x_], 8 + 4 * ((j_) & 1), 9 + 4 * ((j_) & 1), 10 + 4 * ((j_) & 1), 11 + 4 * ((j_) & 1)); \
        q0__ = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(x_) % 3][j_], (b_)[j_], q0__, 0, 0, 0);     \
        q1__ = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(x_) % 3][j_], (b_)[((j_) + 1) & 3], q1__, 0, 0, 0); \
        _Pragma("unroll") for (int r__ = 0; r__ < 4; ++r__) { acc[x_][4 * ((j_) & 1) + r__] = q0__[r__]; acc[x_][8 + 4 * ((j_) & 1) + r__] = q1__[r__]; } \
    } while (0)
#else
#define W4P_MF(x_, j_, b_) acc[x_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(x_) % 3][j_], (b_)[j_], acc[x_], 0, 0, 0)
#endif
#define W4P_MULT(x_, slot_)                                                                         \
    do {                                                                                            \
        if (W4P_HYB(x_)) {                                                                          \
            unsigned wa__[6];                                                                       \
            w4h_split(av[(x_) % 3], wa__);                                                          \
            const u32x4 bq__ = __builtin_bit_cast(u32x4, bv[slot_]);                                \
            u32x4 a_mh = {wa__[0], wa__[1], wa__[2], wa__[3]};                                      \
            u32x4 a_hl = {wa__[2], wa__[3], wa__[4], wa__[5]};                                      \
            u32x4 b_hl = {bq__[2], bq__[3], bl[slot_][0], bl[slot_][1]};                            \
            asm volatile("" : "+v"(a_mh), "+v"(a_hl), "+v"(b_hl));                                  \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            acc[x_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, bq__), acc[x_], 0, 0, 0); \
            acc[x_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_hl), __builtin_bit_cast(bf16x8, bq__), acc[x_], 0, 0, 0); \
            acc[x_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_mh), __builtin_bit_cast(bf16x8, b_hl), acc[x_], 0, 0, 0); \
        } else {                                                                                    \
            if (B2F_W4_MFMA16_TIMING) {   /* TIMING EXPERIMENT (wrong results): the same MACs as eight v_mfma_f32_16x16x4_f32 */ \
                f32x4 q__[4];                                                                       \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) q__[j] = f32x4{acc[x_][4 * j], acc[x_][4 * j + 1], acc[x_][4 * j + 2], acc[x_][4 * j + 3]}; \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                     \
                    q__[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(x_) % 3][j], bv[slot_][j], q__[j], 0, 0, 0); \
                    q__[(j + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(x_) % 3][j], bv[slot_][(j + 1) & 3], q__[(j + 2) & 3], 0, 0, 0); \
                }                                                                                   \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[x_][4 * j + r] = q__[j][r];   \
            } else {                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
                acc[x_] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(x_) % 3][j], bv[slot_][j], acc[x_], 0, 0, 0); \
            }                                                                                       \
        }                                                                                           \
    } while (0)

    // ---- first tile: prologue as in the one-tile kernel ----
    if ((int)blockIdx.x >= total) return;
    W4P_DECODE((int)blockIdx.x, cur_nb, cur_img, cur_ox0, cur_oy0);
    W4P_MASKS(cur_ox0, cur_oy0, mk);
    W4P_RSRC(cur_img, cur_ox0, cur_oy0);
    has_next = false;                                                           // no switch inside the prologue (nchunks >= 4)
    nxt_nb = cur_nb; nxt_img = cur_img; nxt_ox0 = cur_ox0; nxt_oy0 = cur_oy0;
    mk_n[0] = mk[0]; mk_n[1] = mk[1]; mk_n[2] = mk[2];
    int par = 0;                                                                // parity (V / raw buffer) of the tile's chunk 0
    int v_cur = blockIdx.x;
    {
        f32x4 keep[3];
        W4P_LOAD_STREAM();                       // chunk 0
#pragma unroll
        for (int i = 0; i < 3; ++i) keep[i] = sr[i];
        W4P_LOAD_STREAM();                       // chunk 1
        if (s_act) {
            W4_RAW_STORE(Rb, s_slot, keep[0]); W4_RAW_STORE(Rb, s_slot + 6 * RWP, keep[1]); W4_RAW_STORE(Rb, s_slot + 12 * RWP, keep[2]);
        }
        W4P_WRITE_RAW(1);
        W4P_LOAD_STREAM();                       // chunk 2, stays in flight
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) { W4_T_READ(s, 0); W4_T_FMA(s, 0); }
    __syncthreads();

    W4P_RSRC_U(cur_nb);
    W4P_LOAD_U(0, 0, 0); W4P_LOAD_U(1, 0, 1); W4P_LOAD_U(2, 0, 2); W4P_LOAD_U(3, 0, 3); W4P_LOAD_U(4, 0, 4);
    for (;;) {
        // ---- start of a tile: V[par] holds Tr(0), raw buffer par ^ 1 holds chunk 1, chunk 2 is in flight in sr, the
        // first five B operands are in flight (issued under the last output pass of the previous tile) ----
        W4P_T(0);
#if B2F_WINO_TRACE
        if (trp_on && trp_tile == 0) trp_buf[157] = wall_clock64();
        if (trp_on && trp_tile == 11) { trp_buf[158] = wall_clock64(); trp_buf[159] = clock64(); }
#endif
        // the block's next tile: decoded and its lane masks computed here, where registers are plentiful (the
        // accumulators are dead); the load side switches to it three chunks before this tile ends
        has_next = v_cur + G < total;
        if (has_next) {
            W4P_DECODE(v_cur + G, nxt_nb, nxt_img, nxt_ox0, nxt_oy0);
            W4P_MASKS(nxt_ox0, nxt_oy0, mk_n);
        }
        {   // the per-lane constants of the main loop, recomputed from the hardware lane id: held across the output
            // stage they were spilled, and a spill reload here waits (vmcnt counts in order) for the output stores
            int lz = 0;
            asm volatile("" : "+v"(lz));
            const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)lz));
            const int lm = ln & 31, lh = ln >> 5, ltid = wave * 64 + ln;
            const int tb = (lh * 2 + th) * PL2 + (4 * (lm >> 3)) * RWP + (lm & 7);
            t_row[0] = tb + rx1 * RWP; t_row[1] = tb + rx2 * RWP; t_row[2] = tb + rx3 * RWP; t_row[3] = tb + rx4 * RWP;
            t_dst = ((t_orow * 6 * 2 + lh) * 2 + th) * 32 + lm;
            a_off = (9 * g * 2 + lh) * 2 * 32 + lm;
            b_off = ((9 * g * 2 + lh) * 64 + n * 32 + lm) * 16u;
            const int pix = min(ltid, NSTG - 1) >> 1;
            const int r6 = pix / PW, px = pix - r6 * PW;
            s_slot = (ltid & 1) * 2 * PL2 + r6 * RWP + colpos(px);
            l_off = ((unsigned)(r6 * p.W + px) * (unsigned)p.seg[0].pix_stride + (ltid & 1) * 4) * 4u;
        }
        W4_T_READ(0, par ^ 1); W4_T_FMA(0, 0);
        W4_T_READ(1, par ^ 1); W4_T_FMA(1, 0);
        W4_T_READ(2, par ^ 1);
        av[0] = W4_A_READ(reinterpret_cast<const f32x2 *>(Vb + par * VSTRIDE) + a_off);
        av[1] = W4_A_READ(reinterpret_cast<const f32x2 *>(Vb + par * VSTRIDE) + a_off + 128);
#pragma unroll
        for (int x = 0; x < 9; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    // Order inside a xi step (round 4, B2F_W4P_SCHED = 1).  The step trace of the round-3 order ([B load, A read] -> [4 MFMAs] ->
    // [transform slice + its reads]; profiles/r04_wino4_step_trace.txt) shows a wave spending ~140 cycles issuing its loads, ~280
    // multiplying and ~240 in the slice, one after the other: ~650 cycles per step for 256 cycles of matrix pipe, and two such waves
    // per SIMD.  A wave is in-order, but after it has issued a fp32 MFMA it has 64 cycles in which instructions that need neither
    // the matrix pipe nor the VALU (buffer loads, LDS reads / writes, SALU) issue for free.  So: the slice's VALU work FIRST (its
    // rows were read a step ago), then the four MFMAs with the step's memory instructions between them --
    //     slice VALU | MFMA 0 | rows of the next slice | MFMA 1 | B operand of step x + 5 | MFMA 2 | A operand of step x + 1 | MFMA 3
#ifndef B2F_W4P_SCHED
#define B2F_W4P_SCHED 1
#endif
#define W4P_STEP_LOAD_U(PH_, x_, LAST_)                                                             \
    do {                                                                                            \
        if ((x_) + 5 < 9) W4P_LOAD_U((9 * (PH_) + (x_) + 5) % 6, c, (x_) + 5);                      \
        else if (!(LAST_)) W4P_LOAD_U((9 * (PH_) + (x_) + 5) % 6, c + 1, (x_) + 5 - 9);             \
    } while (0)
#define W4P_STEP_A_READS(x_)                                                                        \
    do {                                                                                            \
        if (!(B2F_WINO4_ABLATE & 16)) {                                                             \
            if ((x_) >= 1 && (x_) <= 6) av[((x_) + 1) % 3] = W4_A_READ(Vc + ((x_) + 1) * 128);      \
            if ((x_) == 6) av[8 % 3] = W4_A_READ(Vc + 8 * 128);                                     \
            if ((x_) == 7) av[0] = W4_A_READ(Vn);                                                   \
            if ((x_) == 8) av[1] = W4_A_READ(Vn + 128);                                             \
        }                                                                                           \
    } while (0)
#define W4P_CHUNK(PH_, c_, LAST_)                                                                        \
    do {                                                                                            \
        const int c = (c_);                                                                         \
        const int pc = (par + c) & 1;                                                               \
        const f32x2 *Vc = reinterpret_cast<const f32x2 *>(Vb + pc * VSTRIDE) + a_off;               \
        const f32x2 *Vn = reinterpret_cast<const f32x2 *>(Vb + (pc ^ 1) * VSTRIDE) + a_off;         \
        _Pragma("unroll") for (int x = 0; x < 9; ++x) {                                             \
            W4P_TS(c, x, 0);                                                                        \
            if (B2F_W4P_SCHED && !W4P_HYB(x) && !(B2F_WINO4_ABLATE & 1)) {                          \
                /* ---- interleaved order ---- */                                                   \
                if (x < 6) W4_T_FMA(x + 2, pc ^ 1);                                                 \
                else if (x == 7) W4_T_FMA(0, 0);                                                    \
                else if (x == 8) W4_T_FMA(1, 0);                                                    \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_TS(c, x, 1);                                                                    \
                W4P_MF(x, 0, bv[(9 * (PH_) + x) % 6]);                                              \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                /* rows of the next slice first: they are needed soonest (at the start of the next step) */ \
                if (x < 6) { if (x + 3 < 8) W4_T_READ(x + 3, pc ^ 1); }                             \
                else if (x == 7) W4_T_READ(1, pc);                                                  \
                else if (x == 8) W4_T_READ(2, pc);                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_MF(x, 1, bv[(9 * (PH_) + x) % 6]);                                              \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_STEP_LOAD_U(PH_, x, LAST_);                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_MF(x, 2, bv[(9 * (PH_) + x) % 6]);                                              \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_STEP_A_READS(x);                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_MF(x, 3, bv[(9 * (PH_) + x) % 6]);                                              \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_TS(c, x, 2);                                                                    \
                if (x == 6) {                                                                       \
                    if (!(B2F_WINO4_ABLATE & 2)) W4P_WRITE_RAW(pc);                                 \
                    __builtin_amdgcn_sched_barrier(0);                                              \
                    __syncthreads();                                                                \
                    W4_T_READ(0, pc);                                                               \
                    if (!(LAST_)) W4P_LOAD_STREAM();   /* the last chunk's is issued in the output stage */ \
                }                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_TS(c, x, 3);                                                                    \
            } else {                                                                                \
            /* ---- round-3 order (hybrid steps, ablation builds, B2F_W4P_SCHED = 0) ---- */        \
            W4P_STEP_LOAD_U(PH_, x, LAST_);                                                         \
            W4P_STEP_A_READS(x);                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            W4P_TS(c, x, 1);                                                                        \
            W4P_MULT(x, (9 * (PH_) + x) % 6);                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            W4P_TS(c, x, 2);                                                                        \
            if (B2F_WINO4_ABLATE & 1) {                                                             \
                if (x == 6) { if (!(B2F_WINO4_ABLATE & 2)) W4P_WRITE_RAW(pc); __builtin_amdgcn_sched_barrier(0); __syncthreads(); if (!(LAST_)) W4P_LOAD_STREAM(); } \
            } else if (x < 6) {                                                                     \
                W4_T_FMA(x + 2, pc ^ 1);                                                            \
                if (x + 3 < 8) W4_T_READ(x + 3, pc ^ 1);                                            \
            } else if (x == 6) {                                                                    \
                if (!(B2F_WINO4_ABLATE & 2)) W4P_WRITE_RAW(pc);                                     \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                __syncthreads();                                                                    \
                W4_T_READ(0, pc);                                                                   \
                if (!(LAST_)) W4P_LOAD_STREAM();   /* the last chunk's is issued in the output stage */ \
            } else if (x == 7) {                                                                    \
                W4_T_FMA(0, 0); W4_T_READ(1, pc);                                                   \
            } else {                                                                                \
                W4_T_FMA(1, 0); W4_T_READ(2, pc);                                                   \
            }                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            W4P_TS(c, x, 3);                                                                        \
            }                                                                                       \
        }                                                                                           \
    } while (0)
        W4P_T(1);
        {
            // the tile's last chunk is a separate instantiation that fetches no B operands of a following chunk (loads
            // in flight into dead registers would hold up the output stage: the register allocator reuses them)
            int c2 = 0;
            for (; c2 + 2 < nchunks; c2 += 2) {
                W4P_CHUNK(0, c2, false);
                W4P_CHUNK(1, c2 + 1, false);
            }
            const bool even = c2 + 2 == nchunks;
            if (even) W4P_CHUNK(0, c2, false);
            if (even) W4P_CHUNK(1, c2 + 1, true);
            if (!even) W4P_CHUNK(0, c2, true);
        }
#undef W4P_CHUNK
#undef W4P_STEP_LOAD_U
#undef W4P_STEP_A_READS
        W4P_T(2);

        // ---- output: four passes (tile rows) through the exchange buffer = dead V buffer + gap ----
        const int pl = (par + nchunks - 1) & 1;                                 // V[pl] is dead, V[pl ^ 1] holds the next tile's Tr(0)
        // output stage constants, derived here from an opaque copy of the lane id so that they are not hoisted out of
        // the tile loop (the main loop has no registers to spare): dump addresses relative to the exchange buffer and
        // the item of this thread
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
        // (constant address space: the weights are never written by this kernel, and only then does the compiler use s_load)
        typedef const __attribute__((address_space(4))) float cfloat;
        cfloat *bp = (cfloat *)(p.bias + cur_nb * 64 + 8 * wave);
        float bb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bb[k] = bp[k];
            asm volatile("" : "+s"(bb[k]));       // keeps the eight loads scalar (a select of loads becomes a vector load of a select)
        }
        const f32x4 bias = (olane & 1) ? f32x4{bb[4], bb[5], bb[6], bb[7]} : f32x4{bb[0], bb[1], bb[2], bb[3]};
        const bool col_ok = co0 < p.cout;
        const int ox = cur_ox0 + 4 * o_tx + o_j;
        float *obase = ob + (size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)(cur_oy0 * p.Wo + ox) * p.out_pix_stride + (co0 & 7);
        const float *xa = X + o_rel;
#define W4P_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define W4P_PASS(q_, EXTRA_)                                                                              \
    do {                                                                                            \
        asm volatile("; dump of tile row %0" ::"n"(q_));                                            \
        if (!(B2F_WINO4_ABLATE & 64))                                                               \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
            const unsigned da = xbase + dump_rel[e];                                                \
            _Pragma("unroll") for (int x = 0; x < 8; x += 2)                                        \
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4"                   \
                             :: "v"(da), "v"(acc[x][4 * (q_) + e]), "v"(acc[x + 1][4 * (q_) + e]), "n"(x * 8), "n"((x + 1) * 8) : "memory"); \
            asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(da), "v"(acc[8][4 * (q_) + e]), "n"(8 * 8 * 256) : "memory"); \
        }                                                                                           \
        W4P_T(3 + 3 * (q_));                                                                        \
        W4P_LDS_BARRIER();                                                                          \
        W4P_T(4 + 3 * (q_));                                                                        \
        EXTRA_                                                                                      \
        if (!(B2F_WINO4_ABLATE & 128)) {                                                            \
            const f32x4 sg4 = {o_sg, o_sg, o_sg, o_sg}, kq4 = {o_kq, o_kq, o_kq, o_kq};             \
            const f32x4 k04 = {o_k0, o_k0, o_k0, o_k0}, k34 = {o_k3, o_k3, o_k3, o_k3};             \
            const f32x4 k2 = {2.f, 2.f, 2.f, 2.f}, k4 = {4.f, 4.f, 4.f, 4.f}, k8 = {8.f, 8.f, 8.f, 8.f}; \
            f32x4 T[6];                                                                             \
            _Pragma("unroll") for (int a = 0; a < 6; ++a) {                                         \
                const float *xr6 = xa + (6 * a) * 512;                                              \
                const f32x4 m1 = *reinterpret_cast<const f32x4 *>(xr6 + 1 * 512), m2 = *reinterpret_cast<const f32x4 *>(xr6 + 2 * 512); \
                const f32x4 m3 = *reinterpret_cast<const f32x4 *>(xr6 + 3 * 512), m4 = *reinterpret_cast<const f32x4 *>(xr6 + 4 * 512); \
                const f32x4 me = *reinterpret_cast<const f32x4 *>(xr6 + o_xe);                      \
                const f32x4 e1 = W4_FMA(sg4, m2, m1), e2 = W4_FMA(sg4, m4, m3);                     \
                /* the one-tile kernel's operations, element for element (batching must not change a bit): */ \
                /* j=0 (M0 + s1) + s2, j=1 fma(2, d2, d1), j=2 fma(4, s2, s1), j=3 fma(8, d2, d1) + M5 */ \
                T[a] = W4_FMA(k34, me, W4_FMA(kq4, e2, W4_FMA(k04, me, e1)));                       \
            }                                                                                       \
            const f32x4 s1 = T[1] + T[2], d1 = T[1] - T[2], s2 = T[3] + T[4], d2 = T[3] - T[4];     \
            f32x4 y[4];                                                                             \
            y[0] = T[0] + s1 + s2;                                                                  \
            y[1] = W4_FMA(k2, d2, d1);                                                              \
            y[2] = W4_FMA(k4, s2, s1);                                                              \
            y[3] = W4_FMA(k8, d2, d1) + T[5];                                                       \
            const int oy = cur_oy0 + 4 * (q_);                                                      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                         \
                f32x4 v = y[i] + bias;                                                              \
                if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);                            \
                if (col_ok && oy + i < p.Ho && ox < p.Wo)                                           \
                    *reinterpret_cast<f32x4 *>(obase + (size_t)((4 * (q_) + i) * p.Wo) * p.out_pix_stride) = v; \
            }                                                                                       \
        }                                                                                           \
        W4P_LDS_BARRIER();                                                                          \
        W4P_T(5 + 3 * (q_));                                                                        \
    } while (0)
        // the first B operands of the block's next tile are fetched under the last pass, when the accumulators are dead
        // (without a next tile the loads re-read this tile's and are dropped)
        // (the stream load the last chunk skipped: issued here, after the first dump, so that nothing is in flight
        // when the dump claims registers; it has the rest of the output stage to land)
        W4P_PASS(0, W4P_LOAD_STREAM(););
        W4P_PASS(1, );
        W4P_PASS(2, );
        W4P_PASS(3,
                 W4P_RSRC_U(nxt_nb);
                 W4P_LOAD_U(0, 0, 0); W4P_LOAD_U(1, 0, 1); W4P_LOAD_U(2, 0, 2); W4P_LOAD_U(3, 0, 3); W4P_LOAD_U(4, 0, 4););
#undef W4P_PASS
#undef W4P_LDS_BARRIER

        // ---- next tile of this block: the one whose Tr(0) the last iteration left in V[pl ^ 1] ----
        par = pl ^ 1;
#if B2F_WINO_TRACE
        if (trp_on && trp_tile == 3)
            for (int i = 0; i < 144; ++i) p.trace[1280 + (trp_slot * 2 + (wave >> 2)) * 144 + i] = ts_lds[i];
        ++trp_tile;
#endif
        if (!has_next) break;
        v_cur += G;
        cur_nb = nxt_nb; cur_img = nxt_img; cur_ox0 = nxt_ox0; cur_oy0 = nxt_oy0;
    }
    } else {
    // =====================================================================================================
    // Single N tile (the last 32 channels of a layer whose cout is 32 mod 64, nblk == 1): the 36 xi split over all
    // eight waves as in conv3x3_wino4<1> (wave (g, n): xi = 9g + 5n + x, x = 0..4), simple pipeline (B operands a
    // whole chunk ahead, the raw patch two chunks ahead, one barrier per chunk), persistent over tiles: the stream of
    // raw chunks and the B operands run on into the next tile (same weights for every tile), the output stage takes
    // two passes (two tile rows = 16 tiles x 32 channels x 36 xi = 72 KB each) through the dead V buffer + gap.
    // Arithmetic of the output transform = wino4_output_half's, element for element.
    // =====================================================================================================
    f32x16 acc[5];
    int xo[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) xo[x] = min(9 * g + 5 * n + x, 35);
    const int a_lane = half * 2 * 32 + m;                   // f32x2 index: V[xi][k4 = half][pair 0][tile m] = xi * 128 + a_lane; pair 1 -> + 32
    const unsigned b_lane = (half * 64 + m) * 16u;          // bytes: U[xi][k4 = half][co m] = xi * 2048 + b_lane
    f32x4 av[5], bc[5], bn[5];
    if ((int)blockIdx.x >= total) return;
    W4P_DECODE((int)blockIdx.x, cur_nb, cur_img, cur_ox0, cur_oy0);
    W4P_MASKS(cur_ox0, cur_oy0, mk);
    W4P_RSRC(cur_img, cur_ox0, cur_oy0);
    has_next = false;
    nxt_nb = cur_nb; nxt_img = cur_img; nxt_ox0 = cur_ox0; nxt_oy0 = cur_oy0;
    mk_n[0] = mk[0]; mk_n[1] = mk[1]; mk_n[2] = mk[2];
    int par = 0;
    int v_cur = blockIdx.x;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(p.wpk) + (size_t)cur_nb * nchunks * U_F4 * 16), 0, 0x7fffffff, 0x00020000);
#define W4Q_LOAD_U(dst_, c_)                                                                        \
    _Pragma("unroll") for (int x = 0; x < 5; ++x)                                                   \
        dst_[x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)b_lane, (int)((c_) * (U_F4 * 16) + xo[x] * 2048), 0))
    {
        f32x4 keep[3];
        W4P_LOAD_STREAM();                       // chunk 0
#pragma unroll
        for (int i = 0; i < 3; ++i) keep[i] = sr[i];
        W4P_LOAD_STREAM();                       // chunk 1
        W4Q_LOAD_U(bc, 0);
        if (s_act) {
            W4_RAW_STORE(Rb, s_slot, keep[0]); W4_RAW_STORE(Rb, s_slot + 6 * RWP, keep[1]); W4_RAW_STORE(Rb, s_slot + 12 * RWP, keep[2]);
        }
        W4P_WRITE_RAW(1);
    }
    // The raw patch THREE chunks ahead where the chunk count allows static register roles (even, >= 4: every layer of the shipped
    // graph that runs this form): two register sets, a set is loaded at the top of a chunk and written to LDS at the end of the NEXT one
    // -- two chunk periods of cover; with one set (load and write in the same chunk) the write waited out most of an HBM round trip.
    const bool two_sets = (nchunks & 1) == 0 && nchunks >= 4;
    f32x4 sq[3];
    if (two_sets) W4P_LOAD_STREAM();             // chunk 2, stays in flight
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) { W4_T_READ(s2, 0); W4_T_FMA(s2, 0); }
    __syncthreads();
    for (;;) {
        has_next = v_cur + G < total;
        if (has_next) {
            W4P_DECODE(v_cur + G, nxt_nb, nxt_img, nxt_ox0, nxt_oy0);
            W4P_MASKS(nxt_ox0, nxt_oy0, mk_n);
        }
        W4P_T(0);
#pragma unroll
        for (int x = 0; x < 5; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
        // one chunk: LD_ = the stream load issued at its top, WR_ = the register set written to LDS at its end
#define W4Q_CHUNK(c_, LD_, WR_, BC_, BN_, COPY_)                                                     \
        do {                                                                                        \
            const int c = (c_);                                                                     \
            const int pc = (par + c) & 1;                                                           \
            const int cn = c + 1 < nchunks ? c + 1 : 0;          /* after the last chunk: chunk 0 of the next tile (same weights) */ \
            W4P_TS(c, 5, 3);                                                                        \
            const f32x2 *Vc = reinterpret_cast<const f32x2 *>(Vb + pc * VSTRIDE) + a_lane;          \
            _Pragma("unroll") for (int x = 0; x < 5; ++x) av[x] = W4_A_READ(Vc + xo[x] * 128);      \
            _Pragma("unroll") for (int x = 0; x < 5; ++x) {                                         \
                W4P_TS(c, x, 0);                                                                    \
                if (x < 4) { W4_T_READ_D(2 * x, pc ^ 1, 0); W4_T_READ_D(2 * x + 1, pc ^ 1, 2); }    \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_TS(c, x, 1);                                                                    \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                       \
                    if (x < 4 || n == 0)       /* the fifth step of the n = 1 waves would be a duplicate (only MFMAs sit behind this branch) */ \
                        acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][j], BC_[x][j], acc[x], 0, 0, 0); \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_TS(c, x, 2);                                                                    \
                /* this step's share of the vector-memory issue: the B operand of xi step x of the next chunk, the raw-patch stream at x = 1 \
                   (bunched at the chunk's top both waves of a SIMD queued ~500 cycles of load issue right behind the barrier) */ \
                BN_[x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)b_lane, (int)(cn * (U_F4 * 16) + xo[x] * 2048), 0)); \
                if (x == 1) { LD_; }                             /* runs on into the next tile */   \
                if (x < 4) { W4_T_FMA_D(2 * x, 0, pc ^ 1); W4_T_FMA_D(2 * x + 1, 2, pc ^ 1); }      \
                else { W4P_WRITE_RAW_FROM(pc, WR_); }                                               \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W4P_TS(c, x, 3);                                                                    \
            }                                                                                       \
            W4P_TS(c, 5, 0);                                                                        \
            __syncthreads();                                                                        \
            W4P_TS(c, 5, 1);                                                                        \
            if (COPY_) { _Pragma("unroll") for (int x = 0; x < 5; ++x) bc[x] = bn[x]; }             \
            W4P_TS(c, 5, 2);                                                                        \
        } while (0)
        if (two_sets) {   // static roles: the B registers alternate too (no copy)
            for (int c2 = 0; c2 < nchunks; c2 += 2) {
                W4Q_CHUNK(c2, W4P_LOAD_STREAM_TO(sq), sr, bc, bn, false);    // chunk c + 3 -> sq; sr (chunk c + 2, loaded a chunk ago) -> LDS
                W4Q_CHUNK(c2 + 1, W4P_LOAD_STREAM_TO(sr), sq, bn, bc, false);
            }
        } else {
            for (int c1 = 0; c1 < nchunks; ++c1) W4Q_CHUNK(c1, W4P_LOAD_STREAM_TO(sr), sr, bc, bn, true);   // two chunks ahead, one register set
        }
#undef W4Q_CHUNK
        W4P_T(1);
        // ---- output: two passes (tile rows 2h, 2h + 1) through the dead V buffer + gap ----
        const int pl = (par + nchunks - 1) & 1;
        float *X = reinterpret_cast<float *>(Vb + pl * V_F4);
        const unsigned xbase = static_cast<unsigned>(reinterpret_cast<size_t>(X));
        int oz = 0;
        asm volatile("" : "+v"(oz));
        const int olane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)oz));
        const int om = olane & 31, ohalf = olane >> 5;
        unsigned dump_rel[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t16 = e + 4 * ohalf;                                      // + 8 for the second tile row of the pass
            dump_rel[e] = 4u * (unsigned)(t16 * 32 + (om ^ (((t16 >> 1) & 3) << 3)));
        }
        const int o_tx = olane >> 3, o_j = (olane >> 1) & 3, o_cq = 2 * (wave >> 1) + (olane & 1);
        const int o_rel = ((wave & 1) * 8 + o_tx) * 32 + ((4 * o_cq) ^ (((o_tx >> 1) & 3) << 3));
        const float o_sg = (o_j & 1) ? -1.f : 1.f;
        const float o_kq = o_j == 0 ? 1.f : o_j == 1 ? 2.f : o_j == 2 ? 4.f : 8.f;
        const float o_ke = (o_j == 0 || o_j == 3) ? 1.f : 0.f;
        const int o_xe = (o_j & 1) ? 5 * 512 : 0;                                // M5 for the odd columns, M0 for the even ones
        float *ob = p.out + (size_t)cur_img * p.out_img_stride;
        const int co0 = cur_nb * 64 + 4 * o_cq;
        typedef const __attribute__((address_space(4))) float cfloat;
        cfloat *bp = (cfloat *)(p.bias + cur_nb * 64 + 8 * (wave >> 1));
        float bb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bb[k] = bp[k];
            asm volatile("" : "+s"(bb[k]));
        }
        const f32x4 bias = (olane & 1) ? f32x4{bb[4], bb[5], bb[6], bb[7]} : f32x4{bb[0], bb[1], bb[2], bb[3]};
        const bool col_ok = co0 < p.cout;
        const int ox = cur_ox0 + 4 * o_tx + o_j;
        float *obase = ob + (size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)((cur_oy0 + 4 * (wave & 1)) * p.Wo + ox) * p.out_pix_stride + (co0 & 7);
        const float *xa = X + o_rel;
#define W4Q_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define W4Q_PASS(h_)                                                                                \
    do {                                                                                            \
        _Pragma("unroll") for (int x = 0; x < 5; ++x)                                               \
            if (x < 4 || n == 0) {                                                                  \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                     \
                    const unsigned da = xbase + dump_rel[e] + (unsigned)xo[x] * 2048u;              \
                    asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:0 offset1:4"                 \
                                 :: "v"(da), "v"(acc[x][8 * (h_) + e]), "v"(acc[x][8 * (h_) + 4 + e]) : "memory"); \
                }                                                                                   \
            }                                                                                       \
        W4Q_LDS_BARRIER();                                                                          \
        {                                                                                           \
            const f32x4 sg4 = {o_sg, o_sg, o_sg, o_sg}, kq4 = {o_kq, o_kq, o_kq, o_kq}, ke4 = {o_ke, o_ke, o_ke, o_ke}; \
            const f32x4 k2 = {2.f, 2.f, 2.f, 2.f}, k4 = {4.f, 4.f, 4.f, 4.f}, k8 = {8.f, 8.f, 8.f, 8.f}; \
            f32x4 T[6];                                                                             \
            _Pragma("unroll") for (int a = 0; a < 6; ++a) {                                         \
                const float *xr6 = xa + (6 * a) * 512;                                              \
                const f32x4 m1 = *reinterpret_cast<const f32x4 *>(xr6 + 1 * 512), m2 = *reinterpret_cast<const f32x4 *>(xr6 + 2 * 512); \
                const f32x4 m3 = *reinterpret_cast<const f32x4 *>(xr6 + 3 * 512), m4 = *reinterpret_cast<const f32x4 *>(xr6 + 4 * 512); \
                const f32x4 me = *reinterpret_cast<const f32x4 *>(xr6 + o_xe);                      \
                const f32x4 e1 = W4_FMA(sg4, m2, m1), e2 = W4_FMA(sg4, m4, m3);                     \
                T[a] = W4_FMA(ke4, me, W4_FMA(kq4, e2, e1));                                        \
            }                                                                                       \
            const f32x4 s1 = T[1] + T[2], d1 = T[1] - T[2], s2 = T[3] + T[4], d2 = T[3] - T[4];     \
            f32x4 y[4];                                                                             \
            y[0] = T[0] + s1 + s2;                                                                  \
            y[1] = W4_FMA(k2, d2, d1);                                                              \
            y[2] = W4_FMA(k4, s2, s1);                                                              \
            y[3] = W4_FMA(k8, d2, d1) + T[5];                                                       \
            const int oy = cur_oy0 + 8 * (h_) + 4 * (wave & 1);                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                         \
                f32x4 v = y[i] + bias;                                                              \
                if (p.leaky) v = __builtin_elementwise_max(v, 0.2f * v);                            \
                if (col_ok && oy + i < p.Ho && ox < p.Wo)                                           \
                    *reinterpret_cast<f32x4 *>(obase + (size_t)((8 * (h_) + i) * p.Wo) * p.out_pix_stride) = v; \
            }                                                                                       \
        }                                                                                           \
        W4Q_LDS_BARRIER();                                                                          \
    } while (0)
        W4Q_PASS(0);
        W4P_T(2);
        W4Q_PASS(1);
        W4P_T(3);
#undef W4Q_PASS
#undef W4Q_LDS_BARRIER
        par = pl ^ 1;
#if B2F_WINO_TRACE
        if (trp_on && trp_tile == 0) trp_buf[157] = wall_clock64();
        if (trp_on && trp_tile == 11) { trp_buf[158] = wall_clock64(); trp_buf[159] = clock64(); }
        if (trp_on && trp_tile == 3)
            for (int i = 0; i < 144; ++i) p.trace[1280 + (trp_slot * 2 + (wave >> 2)) * 144 + i] = ts_lds[i];
        ++trp_tile;
#endif
        if (!has_next) break;
        v_cur += G;
        cur_nb = nxt_nb; cur_img = nxt_img; cur_ox0 = nxt_ox0; cur_oy0 = nxt_oy0;
    }
#undef W4Q_LOAD_U
    }
#undef W4P_LOAD_U
#undef W4P_MULT
#undef W4P_RSRC_U
#undef W4P_HYB
#undef W4P_WRITE_RAW
#undef W4P_WRITE_RAW_FROM
#undef W4P_LOAD_STREAM
#undef W4P_LOAD_STREAM_TO
#undef W4P_RSRC
#undef W4P_MASKS
#undef W4P_DECODE
}

// profiling builds: step-level stamps of the persistent two-N-tile kernel (block 40, waves 0 and 4 = the two waves of SIMD 0,
// tile 3, chunks 8..11): per xi step the cycles from the step's start to the multiplications, of the multiplications, of
// the transform slice, and the step's start relative to the chunk's
static void w4p_print_trace(bool on, long long *trace_dev, hipStream_t s)
{
    if (!on || !trace_dev) return;
    std::vector<long long> h(32 * 160);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h.data(), trace_dev, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
        const long long *t = h.data() + 1280 + w * 144;
        if (!t[0]) continue;
        fprintf(stderr, "wino4p step trace, block 40 wave %d, tile 3: per chunk and xi step x: start (cycles since chunk 8 began) | loads + A reads | multiplications | transform slice\n", 4 * w);
        for (int c = 0; c < 4; ++c) {
            fprintf(stderr, "  chunk %2d:", 8 + c);
            for (int x = 0; x < 9; ++x) {
                const long long *u = t + (c * 9 + x) * 4;
                fprintf(stderr, "  x%d +%5lld|%3lld|%4lld|%4lld", x, u[0] - t[0], u[1] - u[0], u[2] - u[1], u[3] - u[2]);
            }
            fprintf(stderr, "\n");
        }
    }
}

template <int NTV>
static hipError_t launch_wino4_t(const ConvLaunch &p, int nb0, int nblk, hipStream_t s)
{
    using namespace wino4;
    static bool attr_done_dev[64] = {false};
    bool &attr_done = attr_done_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wino4<NTV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    ConvLaunch q = p;
    q.nb0 = nb0;
    q.trace = nullptr;
#if B2F_WINO_TRACE
    static long long *trace_dev = nullptr;
    static int traced = 0;
    // B2F_WINO_TRACE=<total chunks of the layer to trace> (1 = the 16-chunk single-segment layer)
    const int tr_want = getenv("B2F_WINO_TRACE") ? atoi(getenv("B2F_WINO_TRACE")) : 0;
    const bool do_trace = tr_want > 0 && traced < 1 && NTV == (getenv("B2F_WINO_TRACE_NTV1") ? 1 : 2) && p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0) == (tr_want == 1 ? 16 : tr_want) && p.H * p.W >= 256 * 480;
    if (do_trace) {
        if (!trace_dev) hipMalloc(&trace_dev, 32 * 160 * sizeof(long long));     // [0, 640) tile stamps, [1280, 1856) step stamps
        hipMemsetAsync(trace_dev, 0, 32 * 160 * sizeof(long long), s);
        q.trace = trace_dev;
    }
#endif
#if B2F_WINO_TRACE
    const bool do_trace_flag = do_trace;
    long long *trace_ptr = trace_dev;
#else
    const bool do_trace_flag = false;
    long long *trace_ptr = nullptr;
#endif
    const int tiles = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH);
    q.nblk = nblk;                          // n-blocks of THIS launch (the kernel decodes them from the 1-D grid)
    dim3 grid((unsigned)(tiles * p.nimg * nblk));
    if (p.w4_persist) {
        // persistent form
        static int n_cu_dev[64] = {0};                  // per device, like the attribute flags (b2f_init_multi: one worker thread per GPU)
        static bool pattr_done_dev[64] = {false};
        bool &pattr_done = pattr_done_dev[attr_slot()];
        int &n_cu = n_cu_dev[attr_slot()];
        if (!n_cu) {
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
            n_cu &= ~7;                     // the XCD remap of the virtual block index needs a multiple of 8
            if (n_cu < 8) n_cu = 8;
        }
        // one block per CU (fewer when the launch has fewer tiles); w4_persist > 1 (tests): exactly that many blocks.
        // Round 3: every launch whose K loop is long enough for the prologue (4 chunks = 32 input channels) runs the persistent
        // form -- with the row-pair input transform the one-tile two-N-tile kernel no longer fits its 256 registers (22
        // spills, reloaded behind vmcnt(0)); it remains the path of shallower layers and of wino4_persistent = 0, bit-identical.
        const int pcap = p.w4_persist > 1 ? p.w4_persist : n_cu;
        const int pgrid = pcap < (int)grid.x ? pcap : (int)grid.x;
        const int nchunks_p = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
        if (nchunks_p >= 4) {
            if (!pattr_done) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wino4p<NTV>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES + 2304 * B2F_WINO_TRACE);
                if (e != hipSuccess) return e;
                pattr_done = true;
            }
            // hybrid forms (two-N-tile blocks with the split packing): p.w4_hybrid = number of bf16 steps per wave
#if B2F_EXPERIMENTS
            if (NTV == 2 && p.wpk_split && p.w4_hybrid > 0) {
                auto go = [&](auto kern) -> hipError_t {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES + 2304 * B2F_WINO_TRACE);
                    if (e != hipSuccess) return e;
                    hipLaunchKernelGGL(kern, dim3((unsigned)pgrid), dim3(512), P_LDS_BYTES + 2304 * B2F_WINO_TRACE, s, q);
                    hipError_t le = hipGetLastError();
                    w4p_print_trace(do_trace_flag, trace_ptr, s);
                    return le;
                };
                switch (p.w4_hybrid) {
                case 2: return go(&conv3x3_wino4p<2, 0x044>);        // steps 2, 6
                case 3: return go(&conv3x3_wino4p<2, 0x092>);        // steps 1, 4, 7
                case 4: return go(&conv3x3_wino4p<2, 0x0AA>);        // steps 1, 3, 5, 7
                case 5: return go(&conv3x3_wino4p<2, 0x155>);        // steps 0, 2, 4, 6, 8
                case 6: return go(&conv3x3_wino4p<2, 0x16D>);        // all but 1, 4, 7
                default: return go(&conv3x3_wino4p<2, 0x1FF>);       // every step
                }
            }
#endif
            hipLaunchKernelGGL((conv3x3_wino4p<NTV>), dim3((unsigned)pgrid), dim3(512), P_LDS_BYTES + 2304 * B2F_WINO_TRACE, s, q);
            w4p_print_trace(do_trace_flag, trace_ptr, s);
#if B2F_WINO_TRACE
            if (do_trace) {
                ++traced;
                std::vector<long long> h(32 * 160);
                hipStreamSynchronize(s);
                hipMemcpy(h.data(), trace_dev, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
                for (int b = 0; b < 4; ++b) {
                    const long long *t = h.data() + b * 160;
                    fprintf(stderr, "shader clock over tiles 0..10: %.0f MHz\n", (double)(t[159] - t[0]) / ((double)(t[158] - t[157]) / 100.0));
                    fprintf(stderr, "wino4p trace block-slot %d wave %d: per tile, cycles since the tile's first stamp: mini-prologue | loop | 4 x (dump, barrier wait, transform + barrier)\n", b >> 1, 4 * (b & 1));
                    for (int k = 0; k < 12; ++k) {
                        const long long *u = t + k * 12;
                        fprintf(stderr, "  tile %2d  start +%7lld | %6lld %7lld |", k, k ? u[0] - t[(k - 1) * 12 + 0] : 0, u[1] - u[0], u[2] - u[1]);
                        for (int q2 = 0; q2 < 4; ++q2) fprintf(stderr, "  %5lld %5lld %5lld", u[3 + 3 * q2] - u[2 + 3 * q2], u[4 + 3 * q2] - u[3 + 3 * q2], u[5 + 3 * q2] - u[4 + 3 * q2]);
                        fprintf(stderr, "\n");
                    }
                }
            }
#endif
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((conv3x3_wino4<NTV>), grid, dim3(512), LDS_BYTES, s, q);
#if B2F_WINO_TRACE
    if (do_trace) {
        ++traced;
        std::vector<long long> h(32 * 160);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), trace_dev, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        for (int b = 0; b < 2; ++b)
            for (int w = 0; w < 8; ++w) {
                const long long *t = h.data() + (b * 8 + w) * 160;
                fprintf(stderr, "wino4 trace block-slot %d wave %d (cycles since first stamp; top | xi 0-2 done | xi 3-5 | xi 6-7 | xi 8 + cols + raw written):\n", b, w);
                for (int c = 0; c < 16; ++c)
                    fprintf(stderr, "  c=%2d  %7lld %7lld %7lld %7lld %7lld\n", c, t[c * 5] - t[0], t[c * 5 + 1] - t[0], t[c * 5 + 2] - t[0],
                            t[c * 5 + 3] - t[0], t[c * 5 + 4] - t[0]);
                fprintf(stderr, "  shader clock loop start .. end: %.0f MHz\n", (double)(t[153] - t[151]) / ((double)(t[156] - t[155]) / 100.0));
                fprintf(stderr, "  kernel start %lld, loop start %lld, loop end %lld, end %lld | epilogue stamps %lld %lld %lld %lld %lld\n", t[150] - t[0],
                        t[151] - t[0], t[152] - t[0], t[153] - t[0], t[140] - t[0], t[141] - t[0], t[142] - t[0], t[143] - t[0], t[144] - t[0]);
            }
    }
#endif
    return hipGetLastError();
}

bool wino4_supported(const ConvLaunch &p)
{
    if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo) return false;
    if (p.nseg > 1 && p.seg[1].pix_stride != p.seg[0].pix_stride) return false;
    if (((p.out_pix_stride | (int)p.out_chunk_stride) & 3) != 0 || (p.cout & 3) != 0) return false;   // 16-byte stores
    for (int i = 0; i < p.nseg; ++i)        // signed 32-bit scalar chunk offsets of the buffer loads
        if ((double)p.seg[i].nchunks * (double)p.seg[i].chunk_stride * 4.0 >= 2147483648.0) return false;
    return (double)p.H * p.W * p.seg[0].pix_stride * 4.0 < 2147483648.0;   // 32-bit byte offsets inside a plane
}

hipError_t launch_conv3x3_wino4(const ConvLaunch &p, hipStream_t s)
{
    if (!wino4_supported(p)) return hipErrorInvalidValue;
    // n-blocks of 64 channels whose two N tiles both hold real channels, then a half-empty last one
    const int nfull = p.cout / 64, rem = p.cout % 64;
    const int n2 = nfull + (rem > 32 ? 1 : 0);
    hipError_t e = hipSuccess;
    // n-blocks with two full N tiles: on the bf16 matrix pipe with split operands when the layer carries that packing
    // (b2f_wino4s.hip; persistent form only, K loop of at least 4 chunks), else on the fp32 MFMA
    // blocks of 64 outputs: Winograd F(2x2) on the bf16 matrix pipe with split operands when the layer carries that packing
#if B2F_EXPERIMENTS
    if (n2 > 0 && wino2s_supported(p)) e = launch_conv3x3_wino2s(p, 0, n2, s);
    else if (n2 > 0 && p.w4_persist && p.w4_hybrid == 0 && wino4s_supported(p)) e = launch_conv3x3_wino4s(p, 0, n2, s);
    else
#endif
    if (n2 > 0) e = launch_wino4_t<2>(p, 0, n2, s);
    if (e == hipSuccess && rem > 0 && rem <= 32) e = launch_wino4_t<1>(p, nfull, 1, s);
    return e;
}

// only the half-empty last n-block (the last <= 32 outputs) of a layer whose full blocks of 64 run elsewhere (b2f_w1b.hip)
hipError_t launch_conv3x3_wino4_rem(const ConvLaunch &p, hipStream_t s)
{
    if (!wino4_supported(p)) return hipErrorInvalidValue;
    const int rem = p.cout % 64;
    if (rem == 0 || rem > 32) return hipSuccess;
    return launch_wino4_t<1>(p, p.cout / 64, 1, s);
}

int wino4_nblk(int cout) { return (cout + 63) / 64; }

size_t wino4_wpk_floats(int cin_chunks, int nblk)
{
    return (size_t)nblk * cin_chunks * wino4::U_F4 * 4;
}

// U = G g G^T in double, rounded once to fp32; packed [nblk][chunk][xi 36][k4 2][64 co][4 ci]
void wino4_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks,
                        int nblk, float *wpk, float *bpk)
{
    static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    std::vector<double> U((size_t)Co * Ci * 36);
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci) {
            const float *gk = w + ((size_t)co * Ci + ci) * 9;
            double t[6][3];
            for (int a = 0; a < 6; ++a)
                for (int v = 0; v < 3; ++v) t[a][v] = G[a][0] * gk[0 * 3 + v] + G[a][1] * gk[1 * 3 + v] + G[a][2] * gk[2 * 3 + v];
            for (int a = 0; a < 6; ++a)
                for (int bq = 0; bq < 6; ++bq)
                    U[((size_t)co * Ci + ci) * 36 + a * 6 + bq] = t[a][0] * G[bq][0] + t[a][1] * G[bq][1] + t[a][2] * G[bq][2];
        }
    for (int nbk = 0; nbk < nblk; ++nbk)
        for (int c = 0; c < cin_chunks; ++c)
            for (int xi = 0; xi < 36; ++xi)
                for (int h = 0; h < 2; ++h)
                    for (int nn = 0; nn < 64; ++nn)
                        for (int j = 0; j < 4; ++j) {
                            const int co = nbk * 64 + nn;
                            const int k = c * kCK + h * 4 + j;
                            const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                            float v = 0.f;
                            if (co < Co && ci >= 0) v = (float)U[((size_t)co * Ci + ci) * 36 + xi];
                            wpk[((((((size_t)nbk * cin_chunks + c) * 36 + xi) * 2 + h) * 64 + nn) * 4) + j] = v;
                        }
    for (int i = 0; i < nblk * 64; ++i) bpk[i] = i < Co ? b[i] : 0.f;
}

}  // namespace b2f
