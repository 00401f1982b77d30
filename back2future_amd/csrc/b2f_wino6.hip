// Winograd F(6x6, 3x3) convolution on the fp32 MFMA for gfx950 (MI355X): the wide stride-1
// nn.SpatialConvolution(Ci,Co,3,3,1,1,1,1) [+ LeakyReLU(0.2)] layers of /root/reference/models/pwc.lua:62,78-82 (decoders and the
// second conv of every convUnit) -- the layers that ran on F(4x4) (b2f_wino4.hip) through round 5.  Same interface (chunk-planar in /
// out, up to two input K segments, bias + LeakyReLU fused), 64/36 = 1.78 multiplications per output and channel pair instead of 2.25:
//
//   Y(6x6) = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A        d = 8x8 input tile, g = 3x3 filter, points 0, +-1, +-2, +-1/2, inf
//   B^T = [1 0 -21/4 0 21/4 0 -1 0; 0 1 1 -17/4 -17/4 1 1 0; 0 -1 1 17/4 -17/4 -1 1 0; 0 1/2 1/4 -5/2 -5/4 2 1 0;
//          0 -1/2 1/4 5/2 -5/4 -2 1 0; 0 2 4 -5/2 -5 1/2 1 0; 0 -2 4 5/2 -5 -1/2 1 0; 0 -1 0 21/4 0 -21/4 0 1]
//   G   = [1 0 0; -2/9 -2/9 -2/9; -2/9 2/9 -2/9; 1/90 1/45 2/45; 1/90 -1/45 2/45; 32/45 16/45 8/45; 32/45 -16/45 8/45; 0 0 1]
//   A^T = [1 1 1 1 1 1 1 0; 0 1 -1 2 -2 1/2 -1/2 0; 0 1 1 4 4 1/4 1/4 0; 0 1 -1 8 -8 1/8 -1/8 0; 0 1 1 16 16 1/16 1/16 0;
//          0 1 -1 32 -32 1/32 -1/32 1]                                                          (Lavin & Gray 2015)
// fp32 throughout.  Rounding against an fp64 convolution: max 4e-5 .. 1.1e-4, mean 2 - 3.6e-6 of the output's standard deviation on the
// decoder shapes -- 1.3 - 2.8 x F(4x4)'s (tools/wino6_numerics.py, profiles/r06_wino6_numerics.txt).
//
// 64 independent GEMMs M_xi[co][tile] = sum_ci U_xi[co][ci] V_xi[ci][tile] (xi = 8a + b) on v_mfma_f32_16x16x4_f32 (the chip's cheapest
// fp32 multiply: full clock where the 32x32x2 form is power-throttled, tools/mfma_f32_power.hip).
//
// What is different from the F(4x4) kernel, and why (profiles/r05_wino4_notes.txt (3), (4): that kernel's chunk is a fixed sequence of
// LDS round trips and one block-wide barrier that no throughput resource explains):
//   * Work item = 16 tiles (2 x 8 tiles = 12 x 48 output pixels) x 64 outputs, block = 512 threads = 8 waves, persistent (one per CU).
//     Wave a owns row a of the transformed tile: xi = 8a .. 8a+7 for ALL 64 outputs -- 8 xi x 4 output tiles of 16 = 32 accumulators of
//     4 registers = 128 registers.
//   * V never touches LDS.  Lane (tile t = lane & 15, k = lane >> 4) of wave a computes V[8a + b][tile t][channels 2k, 2k+1] for
//     b = 0..7 itself -- the row combination of B^T d for row a (a 5-FMA chain per column with wave-uniform coefficients), then the
//     8-point column pass -- as packed fp32 pairs, and (channel 2k | channel 2k+1) of a pair ARE the B operands of the two K steps of
//     the 16x16x4 MFMA (the K order is ours to choose: the weights are packed to match).  No V buffer, no barrier between transform and
//     multiply, no A-operand reads: the only LDS traffic of the K loop is the raw patch, and the only barrier (one per 8-channel chunk)
//     separates data written a whole chunk period before it is read.
//   * The weights (A operand) stream from L2 through an 8-slot register ring, one 16-byte load per (xi, K step) feeds four MFMAs.
//   * K loop, per chunk and wave: 64 MFMAs (2 048 matrix-pipe cycles), 66 packed VALU ops, 48 ds_read_b64, 16 + 3 buffer loads.
//   * Output: wave a holds M[8a + b] for b = 0..7 of (tile, co) in ONE lane, so half of A^T M A (the b direction, 8 -> 6) happens in
//     registers; the a direction goes through LDS in two passes of 32 outputs (exchange buffer 102 KB beside the 2 x 25 KB raw ring).
#include "b2f_internal.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace b2f {

namespace wino6 {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

constexpr int OH = 12, OW = 48;             // output pixels per item: 2 x 8 tiles of 6 x 6
constexpr int PH = OH + 2, PW = OW + 2;     // 14 x 50 input patch
// Raw patch in LDS: four planes of channel PAIRS (f32x2), [pair 4][14 rows][RP] with the columns permuted colpos(p) = (p % 6) * 9 + p / 6:
// column j of the eight tiles of a tile row sits in eight consecutive slots.  A 32-lane group of a transform read (ds_read_b64: 16 tiles x
// 2 pairs) then covers four runs of eight slots -- tile rows 6 RP = 336 = 16 (mod 32) slots apart, pair planes PL = 8 (mod 32) apart --
// that tile the 32 slots = 64 banks: no conflicts.
constexpr int RP = 56;                      // f32x2 slots per patch row
constexpr int PL = 808;                     // f32x2 slots per pair plane (14 * 56 = 784, padded to 8 mod 32)
constexpr int SLOT_BYTES = 4 * PL * 8;      // 25 856 bytes per ring slot
constexpr int SLOT_STRIDE = 32768;          // ring slot 1 starts here
constexpr int SLOT_F2 = SLOT_STRIDE / 8;    // ... in f32x2
constexpr int BIAS_OFF = SLOT_BYTES;        // the gap between the slots holds the launch's bias
constexpr int BIAS_MAX = (SLOT_STRIDE - SLOT_BYTES) / 4;   // 1 728 outputs
constexpr int X_OFF = SLOT_STRIDE + SLOT_BYTES;            // exchange buffer of the output stage
constexpr int XPS = 2176;                   // bytes per (a, j) plane: 16 tiles x 32 outputs x 4 + 128 (odd multiple of 128: see the output stage)
constexpr int X_BYTES = 48 * XPS;
constexpr int LDS_BYTES = X_OFF + X_BYTES;  // 163 072
constexpr int NSTG = 2 * 5 * PW;            // 500 staging threads: (patch row r5 < 5, column < 50, k4 half), three rows r5, r5 + 5, r5 + 9 each
constexpr int UW = 16 * 64 * 16;            // bytes of weights per (chunk, wave row a): 16 steps x 64 lanes x 16 B
constexpr int UC = 8 * UW;                  // bytes of weights per (n-block of 64 outputs, chunk) = 128 KB
__device__ __host__ constexpr int colpos(int p) { return (p % 6) * 9 + p / 6; }
// slot offset of tile column j relative to the tile's column 0 (columns 6, 7 are columns 0, 1 of the next tile)
__device__ __host__ constexpr int cj(int j) { return j < 6 ? 9 * j : 9 * (j - 6) + 1; }
}  // namespace wino6

// Profiling only (results are wrong): -DW6_ABLATE=bits, 1 no input transform, 2 no raw staging, 4 no weight loads, 8 no MFMAs, 16 no period barrier,
// 32 no output rounds (reads, transform, stores), 64 no in-register output transform + dump, 128 every item loads the same patch (L2-resident),
// 256 every output store dropped by the buffer range check, 512 every item of a block stores to its first item's place (L2-resident stores)
#ifndef W6_ABLATE
#define W6_ABLATE 0
#endif
// cache policy bits of the raw-patch loads / the output stores / the weight loads (buffer instruction aux: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef W6_RAW_AUX
#define W6_RAW_AUX 0
#endif
#ifndef W6_OUT_AUX
#define W6_OUT_AUX 0
#endif
#ifndef W6_B_AUX
#define W6_B_AUX 0
#endif
#ifndef W6_NB_DELAY
#define W6_NB_DELAY 0
#endif
#ifndef W6_NB_XCD
#define W6_NB_XCD 0
#endif
// Profiling only: -DW6_TRACE=1 + B2F_WINO_TRACE=<chunks of the layer> in the environment: s_memtime stamps of block 40 (waves 0 and 4) around the
// K loop and inside the output stage of its first items, printed by the launcher
#ifndef W6_TRACE
#define W6_TRACE 0
#endif
#if W6_TRACE
#define W6_T(k_) do { tr_buf[(tr_item < 6 ? tr_item : 6) * 16 + (k_)] = clock64(); } while (0)   /* no branch: every wave stores, all but two into a dummy area */
#else
#define W6_T(k_) do {} while (0)
#endif
#ifndef W6_RING
#define W6_RING 8      // slots of the weight ring (power of two): loads run W6_RING - 1 steps ahead
#endif
#define W6_HALF(v_, hf_) ((hf_) ? __builtin_shufflevector((v_), (v_), 2, 3) : __builtin_shufflevector((v_), (v_), 0, 1))
#define W6_FMA(a_, b_, c_) __builtin_elementwise_fma((a_), (b_), (c_))
// barrier that orders LDS traffic only (__syncthreads() is also a release fence: it would wait for global stores in flight)
#define W6_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// packed fp32 ops as inline asm: hipcc splits packed ops that follow an MFMA into two scalar ones, and VALU instructions are what this loop pays for.
// A constant operand is one HALF of an SGPR pair (lo_ / hi_ forms: op_sel broadcasts that half to both lanes of the packed op): two constants
// per pair -- the loop's 14 and the output stage's 10 constants as separate pairs overflowed the scalar file (124 SGPR spills, each a
// v_readlane on the VALU port).
#define W6_SEL3_LO " op_sel:[0,0,0] op_sel_hi:[0,1,1]"
#define W6_SEL3_HI " op_sel:[1,0,0] op_sel_hi:[1,1,1]"
#define W6_SEL2_LO " op_sel:[0,0] op_sel_hi:[0,1]"
#define W6_SEL2_HI " op_sel:[1,0] op_sel_hi:[1,1]"
#define W6_PK_FMA(d_, c_, h_, x_, y_) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" W6_SEL3_##h_ : "=v"(d_) : "s"(c_), "v"(x_), "v"(y_))
#define W6_PK_FMA_ACC(d_, c_, h_, x_) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" W6_SEL3_##h_ : "+v"(d_) : "s"(c_), "v"(x_))
#define W6_PK_MUL(d_, c_, h_, x_) asm volatile("v_pk_mul_f32 %0, %1, %2" W6_SEL2_##h_ : "=v"(d_) : "s"(c_), "v"(x_))
#define W6_PK_ADD(d_, x_, y_) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d_) : "v"(x_), "v"(y_))
#define W6_PK_SUB(d_, x_, y_) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d_) : "v"(x_), "v"(y_))
#define W6_PK_ACC(d_, x_) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d_) : "v"(x_))

// ---- output stage of one item: A^T M A, LeakyReLU, stores (the bias is already in M).  acc[b][mt] = M[8 wave + b][co = 16 mt + 4 (lane >> 4) + r][tile = lane & 15] ----
// The b direction (8 -> 6) happens in registers -- T[a][j] = sum_b M[a][b] A^T[j][b], the lane holds all eight b of its (tile, outputs) --, the
// a direction goes through LDS in passes of 32 outputs (two MFMA output tiles): X[a 8][j 6] planes of XPS bytes, a plane = [tile 16][32 outputs]
// with the 16-byte unit of output quad q of tile t at (q ^ (t & 7)).
//   * In-register part: packed ops in program order (volatile asm), one (output tile, register pair) at a time, every result written to LDS as
//     soon as it exists: 20 transient registers.  (As elementwise vector code over whole accumulators the compiler interleaved everything
//     and spilled the accumulators -- and the spills landed inside the K loop.)
//   * Reads (ds_read_b128, item = (quad fastest, then x)): the eight quads of a pixel fill 128 bytes, the pixel's neighbour (j + 1, same tile: x
//     pairs never straddle a tile, 6 is even) sits XPS = 128 (mod 256) bytes on -- the other half of the banks.
//   * Item = (quad q of the pass's 32 outputs, pixel x of the 48-wide row, tile row, parity): output rows i = parity, parity + 2, parity + 4 of
//     the tile (the even rows need M0 and the sums M1+M2, M3+M4, M5+M6 of a column, the odd ones the differences and M7), 768 items per parity, 3 per
//     thread of the four waves that take that parity (wave-uniform: its coefficients are scalars); lanes = (quad, x parity, x / 2 ...): a wave's stores cover 8 pixels x 32 bytes in each of four 8-channel chunks.
//   * Stores are buffer stores whose offset is -16 for anything outside the tensor (dropped by the range check): no control flow.
template <int NT>
__device__ __forceinline__ void wino6_output(wino6::f32x4 (&acc)[8][NT], char *smem, const ConvLaunch &p, int tid, int lane, int wave, int nb,
                                             int img, int ox0, int oy0, bool tr_on, int tr_item, long long *tr_buf)
{
    using namespace wino6;
    // the lane id through an opaque register: everything derived from it is computed HERE and not hoisted out of the K loop (hoisted, it sat
    // in scratch, and a reload in this stage waits -- vmcnt counts in order -- for the raw-patch loads in flight)
    {
        int z = 0;
        asm volatile("" : "+v"(z));
        lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
        tid = wave * 64 + lane;
    }
    const f32x2 c2h = {2.f, .5f}, c4h = {4.f, .25f}, c8h = {8.f, .125f}, c16h = {16.f, .0625f}, c32h = {32.f, .03125f};
    const int t16 = lane & 15, q4 = lane >> 4;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)img * p.out_img_stride, 0, 0x7fffffff, 0x00020000);
    const float slope = p.leaky ? 0.2f : 1.f;
    const f32x2 pSl = {slope, slope};
    const int rowstep = 2 * p.Wo * p.out_pix_stride * 4;                     // bytes between the rows an item stores
#pragma unroll
    for (int ps = 0; ps < NT / 2; ++ps) {
        if (ps > 0) W6_LDS_BARRIER();           // the previous pass's reads are over
#pragma unroll
        for (int mtp = 0; mtp < ((W6_ABLATE & 64) ? 0 : 2); ++mtp) {
            const int mt = 2 * ps + mtp;
            const int quad = mtp * 4 + q4;
            char *dp = smem + X_OFF + (wave * 6) * XPS + t16 * 128 + ((quad ^ (t16 & 7)) * 16);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x2 m0 = W6_HALF(acc[0][mt], hf), m1 = W6_HALF(acc[1][mt], hf), m2 = W6_HALF(acc[2][mt], hf), m3 = W6_HALF(acc[3][mt], hf);
                f32x2 m4 = W6_HALF(acc[4][mt], hf), m5 = W6_HALF(acc[5][mt], hf), m6 = W6_HALF(acc[6][mt], hf), m7 = W6_HALF(acc[7][mt], hf);
                // T0 = M0 + s1 + s2 + s3, T1 = d1 + 2 d2 + d3/2, T2 = s1 + 4 s2 + s3/4, T3 = d1 + 8 d2 + d3/8, T4 = s1 + 16 s2 + s3/16,
                // T5 = d1 + 32 d2 + d3/32 + M7  with s = M1+M2, M3+M4, M5+M6 and d the differences -- one asm statement (no padding between its ops)
                // In place: s1, s2, s3 in three temporaries, d1, d2, d3 over M2, M4, M6, then T1 -> M1, T3 -> M3, T2 -> M5, T4 -> s1's register, T0 -> M0,
                // T5 -> M7 (the statement holds 11 register pairs; with separate outputs it held 20 and the kernel spilled).
                f32x2 s1, s2, s3;
                asm volatile("v_pk_add_f32 %0, %4, %5\n\t"                                     /* s1 = M1 + M2 */
                             "v_pk_add_f32 %5, %4, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"           /* d1 = M1 - M2  (over M2) */
                             "v_pk_add_f32 %1, %6, %7\n\t"                                     /* s2 */
                             "v_pk_add_f32 %7, %6, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"           /* d2 (over M4) */
                             "v_pk_add_f32 %2, %8, %9\n\t"                                     /* s3 */
                             "v_pk_add_f32 %9, %8, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"           /* d3 (over M6) */
                             "v_pk_add_f32 %3, %3, %0\n\t"                                     /* T0 = M0 + s1 + s2 + s3 */
                             "v_pk_fma_f32 %4, %11, %7, %5" W6_SEL3_LO "\n\t"                  /* T1 = d1 + 2 d2 ... (over M1) */
                             "v_pk_fma_f32 %8, %12, %1, %0" W6_SEL3_LO "\n\t"                  /* T2 = s1 + 4 s2 ... (over M5) */
                             "v_pk_fma_f32 %6, %13, %7, %5" W6_SEL3_LO "\n\t"                  /* T3 = d1 + 8 d2 ... (over M3) */
                             "v_pk_add_f32 %3, %3, %1\n\t"
                             "v_pk_fma_f32 %4, %11, %9, %4" W6_SEL3_HI "\n\t"                  /* ... + d3/2 */
                             "v_pk_fma_f32 %8, %12, %2, %8" W6_SEL3_HI "\n\t"                  /* ... + s3/4 */
                             "v_pk_fma_f32 %6, %13, %9, %6" W6_SEL3_HI "\n\t"                  /* ... + d3/8 */
                             "v_pk_add_f32 %3, %3, %2\n\t"
                             "v_pk_fma_f32 %0, %14, %1, %0" W6_SEL3_LO "\n\t"                  /* T4 = s1 + 16 s2 ... (over s1) */
                             "v_pk_fma_f32 %5, %15, %7, %5" W6_SEL3_LO "\n\t"                  /* d1 + 32 d2 ... (over d1) */
                             "v_pk_fma_f32 %0, %14, %2, %0" W6_SEL3_HI "\n\t"                  /* ... + s3/16 */
                             "v_pk_fma_f32 %5, %15, %9, %5" W6_SEL3_HI "\n\t"                  /* ... + d3/32 */
                             "v_pk_add_f32 %10, %10, %5"                                       /* T5 = ... + M7 */
                             : "=&v"(s1), "=&v"(s2), "=&v"(s3), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6), "+v"(m7)
                             : "s"(c2h), "s"(c4h), "s"(c8h), "s"(c16h), "s"(c32h));
                const f32x2 t1 = m1, t2 = m5, t3 = m3, t4 = s1;
                f32x2 *d2p = reinterpret_cast<f32x2 *>(dp + 8 * hf);
                d2p[0 * (XPS / 8)] = m0; d2p[1 * (XPS / 8)] = t1; d2p[2 * (XPS / 8)] = t2;
                d2p[3 * (XPS / 8)] = t3; d2p[4 * (XPS / 8)] = t4; d2p[5 * (XPS / 8)] = m7;
            }
        }
        W6_T(2 + 3 * ps);
        W6_LDS_BARRIER();
        W6_T(3 + 3 * ps);
        // parity is wave-uniform (waves 0-3 even rows, 4-7 odd rows): its coefficients are scalars, two per SGPR pair, and the arithmetic is the
        // packed asm of the K loop (as elementwise vector code hipcc kept every splat coefficient in a VGPR pair: 20 registers, spilled)
        const int par = wave >> 2;
        const f32x2 pSK = {par ? -1.f : 1.f, par ? 0.f : 1.f};                  // (+-1 | weight of M0 in the first row)
        const f32x2 pA = {par ? 2.f : 1.f, par ? .5f : 1.f}, pB = {par ? 8.f : 4.f, par ? .125f : .25f}, pC = {par ? 32.f : 16.f, par ? .03125f : .0625f};
        const f32x2 pK7 = {par ? 1.f : 0.f, 0.f};                               // weight of M7 in the last row
        const int xe_off = (par ? 7 : 0) * 6 * XPS;
#pragma unroll
        for (int rr = 0; rr < ((W6_ABLATE & 32) ? 0 : 3); ++rr) {
            const int idx = (tid & 255) + 256 * rr;                             // 768 items of this parity
            const int quad = idx & 7, xl = (idx >> 3) & 1, mm = idx >> 4;       // mm in 0..47
            const int tyy = mm >= 24 ? 1 : 0, x2 = mm - 24 * tyy;
            const int x = 2 * x2 + xl, txx = x / 6, j = x - 6 * txx;
            const int tile = tyy * 8 + txx;
            const char *xb = smem + X_OFF + j * XPS + tile * 128 + ((quad ^ (tile & 7)) * 16);
            const int co0 = nb * 64 + ps * 32 + quad * 4;
            // even rows: y0 = M0 + s1 + s2 + s3, y2 = s1 + 4 s2 + s3/4, y4 = s1 + 16 s2 + s3/16;  odd: y1 = d1 + 2 d2 + d3/2, y3 = d1 + 8 d2 + d3/8,
            // y5 = d1 + 32 d2 + d3/32 + M7 -- one instruction stream with wave-uniform coefficients, one half (two outputs) of the quad at a time
            f32x2 y[3][2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const f32x2 *xh = reinterpret_cast<const f32x2 *>(xb + 8 * hf);
                f32x2 m1 = xh[1 * 6 * (XPS / 8)], m2 = xh[2 * 6 * (XPS / 8)], m3 = xh[3 * 6 * (XPS / 8)], m4 = xh[4 * 6 * (XPS / 8)];
                f32x2 m5 = xh[5 * 6 * (XPS / 8)], m6 = xh[6 * 6 * (XPS / 8)];
                const f32x2 me = *reinterpret_cast<const f32x2 *>(xb + 8 * hf + xe_off);
                // in place: e1, e2, e3 over M1, M3, M5; the three rows over M2, M4, M6
                asm volatile("v_pk_fma_f32 %0, %7, %1, %0" W6_SEL3_LO "\n\t"                   /* e1 = M1 +- M2 */
                             "v_pk_fma_f32 %2, %7, %3, %2" W6_SEL3_LO "\n\t"                   /* e2 = M3 +- M4 */
                             "v_pk_fma_f32 %4, %7, %5, %4" W6_SEL3_LO "\n\t"                   /* e3 = M5 +- M6 */
                             "v_pk_fma_f32 %1, %7, %6, %0" W6_SEL3_HI "\n\t"                   /* row A = k0 Me + e1 ... */
                             "v_pk_fma_f32 %3, %9, %2, %0" W6_SEL3_LO "\n\t"                   /* row B = e1 + b2 e2 ... */
                             "v_pk_fma_f32 %5, %10, %2, %0" W6_SEL3_LO "\n\t"                  /* row C = e1 + c2 e2 ... */
                             "v_pk_fma_f32 %1, %8, %2, %1" W6_SEL3_LO "\n\t"                   /* ... + a2 e2 */
                             "v_pk_fma_f32 %3, %9, %4, %3" W6_SEL3_HI "\n\t"                   /* ... + b3 e3 */
                             "v_pk_fma_f32 %5, %10, %4, %5" W6_SEL3_HI "\n\t"                  /* ... + c3 e3 */
                             "v_pk_fma_f32 %1, %8, %4, %1" W6_SEL3_HI "\n\t"                   /* ... + a3 e3 */
                             "v_pk_fma_f32 %5, %11, %6, %5" W6_SEL3_LO                         /* ... + k7 Me */
                             : "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(m5), "+v"(m6)
                             : "v"(me), "s"(pSK), "s"(pA), "s"(pB), "s"(pC), "s"(pK7));
                y[0][hf] = m2; y[1][hf] = m4; y[2][hf] = m6;
            }
            const int oy = oy0 + 6 * tyy + par, ox = ox0 + x;
            const unsigned ooff = (unsigned)(((size_t)(co0 >> 3) * p.out_chunk_stride + (size_t)(oy * p.Wo + ox) * p.out_pix_stride + (co0 & 7)) * 4);
            const bool ok = !(W6_ABLATE & 256) && co0 < p.cout && ox < p.Wo;            // (ablation 256: every store dropped by the range check)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                // LeakyReLU(0.2) = max(v, 0.2 v) (slope 1: identity): two packed multiplies, four plain maxima as asm (fmaxf() costs a second
                // v_max per element: hipcc canonicalises NaNs first).  The bias came through the accumulators (W6_ITEM_START).
                f32x2 tl0, tl1;
                W6_PK_MUL(tl0, pSl, LO, y[i][0]);
                W6_PK_MUL(tl1, pSl, LO, y[i][1]);
                f32x4 v;
                asm("v_max_f32 %0, %1, %2" : "=v"(v[0]) : "v"(y[i][0][0]), "v"(tl0[0]));
                asm("v_max_f32 %0, %1, %2" : "=v"(v[1]) : "v"(y[i][0][1]), "v"(tl0[1]));
                asm("v_max_f32 %0, %1, %2" : "=v"(v[2]) : "v"(y[i][1][0]), "v"(tl1[0]));
                asm("v_max_f32 %0, %1, %2" : "=v"(v[3]) : "v"(y[i][1][1]), "v"(tl1[1]));
                // (the row offset goes into the lane offset, NOT into the scalar offset: behind a 16-byte buffer store with a register soffset hipcc
                // emits no wait state before a VALU write of the store's data registers -- LLVM's hazard rule exempts that form -- and on gfx950
                // the next instruction's result reached memory in some lanes: first output of a quad wrong, run to run different)
                const unsigned vo = (ok && oy + 2 * i < p.Ho) ? ooff + (unsigned)(i * rowstep) : 0xfffffff0u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), o_rsrc, (int)vo, 0, W6_OUT_AUX);
            }
            __builtin_amdgcn_sched_barrier(0);                                  // one round at a time: hoisting the next round's reads spilled registers
        }
        W6_T(4 + 3 * ps);
    }
}

template <int NT>
__global__ __launch_bounds__(512) void conv3x3_wino6(const ConvLaunch p)
{
    using namespace wino6;
    static_assert(NT == 4 || NT == 2, "64 or 32 outputs per block");
    constexpr int NSTEP = 4 * NT;           // steps of four MFMAs per chunk: 16 (step = (b, K step), the four output tiles) or 8 (step = b: 2 K steps x 2 output tiles)
    constexpr int SPS = 16 / NSTEP;         // slots of side work per step (the transform of a chunk is 16 slots)
    constexpr int LAG = SPS;                // slots between the reads of a column and its FMA chain (32-output blocks: two, from a second register set)
    constexpr int UCN = UC * NT / 4;        // bytes of weights per chunk of this block form
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x2 *raw2 = reinterpret_cast<f32x2 *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int tiles_x = (p.Wo + OW - 1) / OW, tiles_y = (p.Ho + OH - 1) / OH;
    const int total = tiles_x * tiles_y * p.nimg * p.nblk;
    const int G = gridDim.x;
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    if ((int)blockIdx.x >= total) return;

    // the launch's bias -> LDS (read by the output stage; a vector load there would queue behind the raw-patch loads in flight)
    for (int i = tid; i < p.nblk * 64; i += 512) reinterpret_cast<float *>(smem + BIAS_OFF)[i] = p.bias[p.nb0 * 64 + i];

    // ---- row combination of this wave: (B^T d)[a][j] = d[rA + 6][j] + c1 d[rA + 2][j] + c2 d[rA + 4][j] + c3 d[rB + 1][j] + c4 d[rB + 3][j] + c5 d[rB + 5][j]
    // with two wave-uniform row shifts, so that ONE instruction stream with immediate row offsets serves all eight rows of B^T:
    //   rows 1..6 (rA = rB = 0):  (d6 + al d2 + be d4) +- (ga d1 + de d3 + ep d5)
    //   row 0, NEGATED (rA = 0, rB = -1): d6 - 5.25 d4 + 5.25 d2 - d0 -- the packer negates U_xi for xi = 0..7 to match: (-U)(-V) = U V
    //   row 7 (rA = +1, rB = 0):  d7 - d1 + 5.25 d3 - 5.25 d5
    int rA = 0, rB = 0;
    float rc1, rc2, rc3, rc4, rc5;          // literal per wave: a float product here would be a VALU result, and the "s" constraint silently takes VGPRs
    switch (wave) {
    case 0: rB = -1; rc1 = 5.25f; rc2 = -5.25f; rc3 = -1.f; rc4 = 0.f; rc5 = 0.f; break;
    case 1: rc1 = 1.f; rc2 = -4.25f; rc3 = 1.f; rc4 = -4.25f; rc5 = 1.f; break;
    case 2: rc1 = 1.f; rc2 = -4.25f; rc3 = -1.f; rc4 = 4.25f; rc5 = -1.f; break;
    case 3: rc1 = .25f; rc2 = -1.25f; rc3 = .5f; rc4 = -2.5f; rc5 = 2.f; break;
    case 4: rc1 = .25f; rc2 = -1.25f; rc3 = -.5f; rc4 = 2.5f; rc5 = -2.f; break;
    case 5: rc1 = 4.f; rc2 = -5.f; rc3 = 2.f; rc4 = -2.5f; rc5 = .5f; break;
    case 6: rc1 = 4.f; rc2 = -5.f; rc3 = -2.f; rc4 = 2.5f; rc5 = -.5f; break;
    default: rA = 1; rc1 = 0.f; rc2 = 0.f; rc3 = -1.f; rc4 = 5.25f; rc5 = -5.25f; break;
    }
    const f32x2 pc12 = {rc1, rc2}, pc34 = {rc3, rc4}, pc5x = {rc5, rc5};
    // thread = (tile t = lane & 15 -> (ty, tx), channel pair kq = lane >> 4 of the chunk's four)
    const int t_t = lane & 15, t_kq = lane >> 4;
    const int t_b = t_kq * PL + (6 * (t_t >> 3)) * RP + (t_t & 7);          // f32x2 index of (plane kq, tile row 0, tile column 0) in ring slot 0
    // (wave 0 reads patch row -1 of tile row 0 with weight 0 ... never: its rB rows are 0, 2, 4 >= 0; wave 7's rA rows are 7, 3, 5 <= 13)
    const int t_iA = t_b + rA * RP, t_iB = t_b + rB * RP;

    // ---- staging of the raw patch: thread = (patch row r5 < 5, column px < 50, k4 half sh) stages the pixels (r5 + {0, 5, 9}, px) of the 14 x 50
    // patch -- row 9 twice (r5 = 4 and r5 = 0) and threads 500..511 repeat threads 498, 499: the same bytes to the same place, so that every
    // thread does the same three loads and six LDS writes with no predicate (a branch inside the period costs registers, see W6_LOAD_STREAM).
    // Nothing a thread holds depends on the item: the item enters through the base address of the buffer resource (scalar) and three 64-bit
    // lane masks (scalar: the lanes whose pixel lies inside the image; the others load at offset -16, which the range check of the buffer
    // load turns into zeros -- the zero padding of the convolution) ----
    const int s_idx = min(tid, NSTG - 1) >> 1, s_h = tid & 1;
    const int s_r5 = s_idx / PW, s_px = s_idx - s_r5 * PW;
    const int s_slot = (2 * s_h) * PL + s_r5 * RP + colpos(s_px);           // f32x2 index of pair 2 sh in ring slot 0; pair 2 sh + 1 -> + PL; rows + 5, + 9 -> + 5 RP, + 9 RP
    const unsigned l_off = ((unsigned)(s_r5 * p.W + s_px) * (unsigned)p.seg[0].pix_stride + s_h * 4) * 4u;
    const int rowbytes = p.W * p.seg[0].pix_stride * 4;                      // bytes between patch rows
    u64 mk[3], mk_n[3];
    int cur_nb, cur_img, cur_ox0, cur_oy0;
    int nxt_nb, nxt_img, nxt_ox0, nxt_oy0;
    bool has_next;
    int lc = 0;                                                              // load side: next chunk of its item
    // Virtual block index -> item.  Default: every XCD walks a contiguous range of (tile, n-block) with the n-block fastest (the n-blocks of a
    // tile run side by side and share the raw patch in L2).  W6_NB_XCD (launches of 2, 4 or 8 n-blocks whose tile count divides evenly): an XCD
    // works on ONE n-block -- its L2 then holds that n-block's weights only (200 -> 128: 3.2 of 6.4 MB; an L2 has 4 MB) -- and a contiguous
    // range of tiles; the patch is read by nblk XCDs instead of one.
    const int nb_xcd = (W6_NB_XCD && p.nblk > 1 && 8 % p.nblk == 0 && (G & 7) == 0 && (total / p.nblk) % (8 / p.nblk) == 0) ? 1 : 0;
#define W6_DECODE(v_, nb_, img_, ox0_, oy0_)                                                        \
    do {                                                                                            \
        int bid__;                                                                                  \
        if (nb_xcd) {                                                                               \
            const int xcd__ = (v_) & 7, k__ = (v_) >> 3;                                            \
            nb_ = xcd__ % p.nblk + p.nb0;                                                           \
            bid__ = (xcd__ / p.nblk) * ((total / p.nblk) / (8 / p.nblk)) + k__;                     \
        } else {                                                                                    \
            bid__ = xcd_remap((v_), total);                                                         \
            nb_ = bid__ % p.nblk + p.nb0;                                                           \
            bid__ /= p.nblk;                                                                        \
        }                                                                                           \
        ox0_ = (bid__ % tiles_x) * OW;                                                              \
        bid__ /= tiles_x;                                                                           \
        oy0_ = (bid__ % tiles_y) * OH;                                                              \
        img_ = bid__ / tiles_y;                                                                     \
    } while (0)
#define W6_MASKS(ox0_, oy0_, out_)                                                                  \
    do {                                                                                            \
        const int gx__ = (ox0_) - 1 + s_px;                                                         \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                             \
            const int gy__ = (oy0_) - 1 + s_r5 + (i == 2 ? 9 : 5 * i);                              \
            out_[i] = __builtin_amdgcn_ballot_w64(gy__ >= 0 && gy__ < p.H && gx__ >= 0 && gx__ < p.W); \
        }                                                                                           \
    } while (0)
    // base addresses of the two K segments at the patch origin (oy0 - 1, ox0 - 1) of an item (may lie before the tensor: never dereferenced there)
#define W6_BASES(img_, ox0_, oy0_, b0_, b1_)                                                        \
    do {                                                                                            \
        const long long o__ = ((long long)((oy0_) - 1) * p.W + ((ox0_) - 1)) * p.seg[0].pix_stride; \
        b0_ = p.seg[0].ptr + ((W6_ABLATE & 128) ? 0 : ((long long)(img_) * p.seg[0].img_stride + o__)); \
        b1_ = p.seg[1].ptr + ((W6_ABLATE & 128) ? 0 : ((long long)(img_) * p.seg[1].img_stride + o__)); \
    } while (0)
    const float *ld_b0, *ld_b1, *nxt_b0, *nxt_b1;                            // load side's item / the block's next item
    f32x4 sr[3];
    // next chunk of the load stream -> sr; after an item's last chunk the stream moves on to the block's next item (and keeps re-reading the very
    // last chunk when there is none: harmless).  The switch is a handful of scalar selects, NOT a branch: control flow in the middle of a
    // period (everything live) made the register allocator spill 270 registers.
#define W6_LOAD_STREAM()                                                                            \
    do {                                                                                            \
        const bool s1__ = lc >= p.seg[0].nchunks;                                                   \
        const long cstr__ = s1__ ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;                   \
        const int cc__ = s1__ ? lc - p.seg[0].nchunks : lc;                                         \
        const int so__ = (int)(cc__ * cstr__ * 4);                                                  \
        const __amdgpu_buffer_rsrc_t rs__ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(s1__ ? ld_b1 : ld_b0), 0, 0x7fffffff, 0x00020000); \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                             \
            unsigned vo__;                                                                          \
            asm("v_cndmask_b32_e64 %0, -16, %1, %2" : "=v"(vo__) : "v"(l_off), "s"(mk[i]));         \
            sr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs__, (int)vo__, so__ + (i == 2 ? 9 : 5 * i) * rowbytes, W6_RAW_AUX)); \
        }                                                                                           \
        const bool wrap__ = lc + 1 == nchunks, sw__ = wrap__ && has_next;                           \
        lc = wrap__ ? (has_next ? 0 : nchunks - 1) : lc + 1;                                        \
        mk[0] = sw__ ? mk_n[0] : mk[0]; mk[1] = sw__ ? mk_n[1] : mk[1]; mk[2] = sw__ ? mk_n[2] : mk[2]; \
        ld_b0 = sw__ ? nxt_b0 : ld_b0; ld_b1 = sw__ ? nxt_b1 : ld_b1;                               \
    } while (0)
    // sr -> ring slot based at wr_ (f32x2 pointer): a float4 (4 channels of a k4 half) goes to the two pair planes
#define W6_RAW_WRITE(wr_)                                                                           \
    do {                                                                                            \
        f32x2 *w__ = (wr_) + s_slot;                                                                \
        w__[0] = __builtin_shufflevector(sr[0], sr[0], 0, 1); w__[PL] = __builtin_shufflevector(sr[0], sr[0], 2, 3); \
        w__[5 * RP] = __builtin_shufflevector(sr[1], sr[1], 0, 1); w__[5 * RP + PL] = __builtin_shufflevector(sr[1], sr[1], 2, 3); \
        w__[9 * RP] = __builtin_shufflevector(sr[2], sr[2], 0, 1); w__[9 * RP + PL] = __builtin_shufflevector(sr[2], sr[2], 2, 3); \
    } while (0)

    // ---- input transform: column col_ of the row combination (reads, then the 5-FMA chain), and the column pass in four parts ----
    f32x2 V[2][8];      // B operands of the MFMAs: V[parity][b] = (K step 0 | K step 1)
    f32x2 R[8];         // (B^T d)[a][0..7]
    f32x2 dq[SPS][6];
#define W6_ROW_READ(col_, rd_)                                                                      \
    do {                                                                                            \
        const f32x2 *r__ = (rd_) + cj(col_);                                                        \
        f32x2 *d__ = dq[(col_) & (SPS - 1)];                                                        \
        d__[0] = r__[t_iA + 6 * RP]; d__[1] = r__[t_iA + 2 * RP]; d__[2] = r__[t_iA + 4 * RP];      \
        d__[3] = r__[t_iB + 1 * RP]; d__[4] = r__[t_iB + 3 * RP]; d__[5] = r__[t_iB + 5 * RP];      \
    } while (0)
    // (one asm statement per chain: between two dependent asm statements hipcc pads an s_nop that the hardware does not need -- VALU results
    // are interlocked -- 32 of them per chunk)
#define W6_ROW_FMA(col_)                                                                            \
    do {                                                                                            \
        f32x2 t__;                                                                                  \
        const f32x2 *d__ = dq[(col_) & (SPS - 1)];                                                  \
        asm volatile("v_pk_fma_f32 %0, %1, %5, %4" W6_SEL3_LO "\n\t"                                \
                     "v_pk_fma_f32 %0, %1, %6, %0" W6_SEL3_HI "\n\t"                                \
                     "v_pk_fma_f32 %0, %2, %7, %0" W6_SEL3_LO "\n\t"                                \
                     "v_pk_fma_f32 %0, %2, %8, %0" W6_SEL3_HI "\n\t"                                \
                     "v_pk_fma_f32 %0, %3, %9, %0" W6_SEL3_LO                                        \
                     : "=&v"(t__) : "s"(pc12), "s"(pc34), "s"(pc5x), "v"(d__[0]), "v"(d__[1]), "v"(d__[2]), "v"(d__[3]), "v"(d__[4]), "v"(d__[5])); \
        R[(col_) & 7] = t__;                                                                        \
    } while (0)
    const f32x2 qA = {5.25f, -4.25f}, qB = {.25f, -1.25f}, qC = {.5f, -2.5f}, qD = {2.f, 4.f}, qE = {-5.f, -5.f};
#define W6_COLPASS(part_, vp_)                                                                      \
    do {                                                                                            \
        f32x2 P__, Q__;                                                                             \
        if ((part_) == 0) {                                                                         \
            /* V0 = (R0 - R6) + 5.25 (R4 - R2),  V7 = (R7 - R1) + 5.25 (R3 - R5) */                 \
            asm volatile("v_pk_add_f32 %2, %6, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"                    \
                         "v_pk_add_f32 %0, %4, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"                    \
                         "v_pk_fma_f32 %0, %12, %2, %0" W6_SEL3_LO "\n\t"                           \
                         "v_pk_add_f32 %3, %8, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"                    \
                         "v_pk_add_f32 %1, %11, %10 neg_lo:[0,1] neg_hi:[0,1]\n\t"                  \
                         "v_pk_fma_f32 %1, %12, %3, %1" W6_SEL3_LO                                  \
                         : "=&v"(V[vp_][0]), "=&v"(V[vp_][7]), "=&v"(P__), "=&v"(Q__)               \
                         : "v"(R[0]), "v"(R[2]), "v"(R[4]), "v"(R[6]), "v"(R[3]), "v"(R[5]), "v"(R[1]), "v"(R[7]), "s"(qA)); \
        } else if ((part_) == 1) {                                                                  \
            /* P = R2 + R6 - 4.25 R4, Q = R1 + R5 - 4.25 R3, V1 = P + Q, V2 = P - Q */              \
            asm volatile("v_pk_add_f32 %2, %4, %6\n\t"                                              \
                         "v_pk_fma_f32 %2, %10, %5, %2" W6_SEL3_HI "\n\t"                           \
                         "v_pk_add_f32 %3, %7, %9\n\t"                                              \
                         "v_pk_fma_f32 %3, %10, %8, %3" W6_SEL3_HI "\n\t"                           \
                         "v_pk_add_f32 %0, %2, %3\n\t"                                              \
                         "v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]"                        \
                         : "=&v"(V[vp_][1]), "=&v"(V[vp_][2]), "=&v"(P__), "=&v"(Q__)               \
                         : "v"(R[2]), "v"(R[4]), "v"(R[6]), "v"(R[1]), "v"(R[3]), "v"(R[5]), "s"(qA)); \
        } else if ((part_) == 2) {                                                                  \
            /* P = R6 + .25 R2 - 1.25 R4, Q = .5 R1 - 2.5 R3 + 2 R5, V3 = P + Q, V4 = P - Q */      \
            asm volatile("v_pk_fma_f32 %2, %10, %4, %6" W6_SEL3_LO "\n\t"                           \
                         "v_pk_fma_f32 %2, %10, %5, %2" W6_SEL3_HI "\n\t"                           \
                         "v_pk_mul_f32 %3, %11, %7" W6_SEL2_LO "\n\t"                               \
                         "v_pk_fma_f32 %3, %11, %8, %3" W6_SEL3_HI "\n\t"                           \
                         "v_pk_fma_f32 %3, %12, %9, %3" W6_SEL3_LO "\n\t"                           \
                         "v_pk_add_f32 %0, %2, %3\n\t"                                              \
                         "v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]"                        \
                         : "=&v"(V[vp_][3]), "=&v"(V[vp_][4]), "=&v"(P__), "=&v"(Q__)               \
                         : "v"(R[2]), "v"(R[4]), "v"(R[6]), "v"(R[1]), "v"(R[3]), "v"(R[5]), "s"(qB), "s"(qC), "s"(qD)); \
        } else {                                                                                    \
            /* P = R6 + 4 R2 - 5 R4, Q = 2 R1 - 2.5 R3 + .5 R5, V5 = P + Q, V6 = P - Q */           \
            asm volatile("v_pk_fma_f32 %2, %12, %4, %6" W6_SEL3_HI "\n\t"                           \
                         "v_pk_fma_f32 %2, %10, %5, %2" W6_SEL3_LO "\n\t"                           \
                         "v_pk_mul_f32 %3, %12, %7" W6_SEL2_LO "\n\t"                               \
                         "v_pk_fma_f32 %3, %11, %8, %3" W6_SEL3_HI "\n\t"                           \
                         "v_pk_fma_f32 %3, %11, %9, %3" W6_SEL3_LO "\n\t"                           \
                         "v_pk_add_f32 %0, %2, %3\n\t"                                              \
                         "v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]"                        \
                         : "=&v"(V[vp_][5]), "=&v"(V[vp_][6]), "=&v"(P__), "=&v"(Q__)               \
                         : "v"(R[2]), "v"(R[4]), "v"(R[6]), "v"(R[1]), "v"(R[3]), "v"(R[5]), "s"(qE), "s"(qC), "s"(qD)); \
        }                                                                                           \
    } while (0)

    // ---- weights: one 16-byte load per step feeds its four MFMAs; 8-slot ring, seven steps ahead.
    //   64-output blocks  [n-block][chunk][wave row a][step = 2 b + s][lane][4 output tiles]: lane (co = lane & 15, kk = lane >> 4) holds
    //                     U[xi = 8a + b][co 16 mt + (lane & 15)][channel 2 kk + s], mt = 0..3
    //   32-output block   [chunk][wave row a][step = b][lane][s 2][mt 2]   (the last block of a layer whose outputs are <= 32 mod 64) ----
    f32x4 bq[W6_RING];
    f32x4 acc[8][NT];
    const unsigned b_voff = (unsigned)lane * 16u + (unsigned)wave * (unsigned)(UW * NT / 4);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wpk), 0, 0x7fffffff, 0x00020000);
    int bo_c, bo_n;                                                          // byte offsets of the compute side's chunk and of the chunk that follows it
#define W6_B_LOAD(u_)                                                                               \
    do {                                                                                            \
        const int o__ = (u_) < NSTEP ? bo_c + (u_) * 1024 : bo_n + ((u_) - NSTEP) * 1024;           \
        bq[(u_) & (W6_RING - 1)] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (int)b_voff, o__, W6_B_AUX)); \
    } while (0)
    // MFMA i of step st_: (b, K step s, output tile mt) = (st >> 1, st & 1, i) resp. (st, i >> 1, i & 1)
#define W6_MFMA(pp_, st_, i_)                                                                       \
    do {                                                                                            \
        const int b__ = NT == 4 ? (st_) >> 1 : (st_), s__ = NT == 4 ? (st_) & 1 : (i_) >> 1, mt__ = NT == 4 ? (i_) : (i_) & 1;     \
        acc[b__][mt__] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[(st_) & (W6_RING - 1)][i_], V[pp_][b__][s__], acc[b__][mt__], 0, 0, 0); \
    } while (0)

    // One chunk period (parity PP_): the MFMAs of the compute side's chunk on V[PP_], and between them, pinned by scheduling barriers, the 16
    // slots of side work --
    //   slots 0..7      the reads of column k of the NEXT chunk's row combination (ring slot PP_ ^ 1); slots LAG..7 + LAG its FMA chain,
    //   slots 12..15    the column pass -> V[PP_ ^ 1] (late: the registers of the V[PP_] pairs already multiplied are free by then),
    //   slot 9          the raw patch two chunks ahead (loaded a period ago) -> ring slot PP_,   slot 10   the loads of the patch three chunks ahead,
    //   every step      one weight load, seven steps ahead (after MFMA 0: the slot it overwrites was last read by the previous step).
    // VALU work first (its operands were read LAG slots ago), memory instructions between the MFMAs.  One barrier per period: what it orders
    // was written at least a step before it and is read a whole period later.
#define W6_SLOT_VALU(k_, PP_)                                                                       \
    do {                                                                                            \
        if ((k_) >= LAG && (k_) < 8 + LAG && !(W6_ABLATE & 1)) W6_ROW_FMA((k_) - LAG);              \
        if ((k_) >= 12 && !(W6_ABLATE & 1)) W6_COLPASS((k_) - 12, (PP_) ^ 1);                       \
    } while (0)
#define W6_SLOT_LDS(k_, PP_)                                                                        \
    do {                                                                                            \
        if ((k_) <= 7 && !(W6_ABLATE & 1)) W6_ROW_READ((k_), rd__);                                 \
        if ((k_) == 9 && !(W6_ABLATE & 2)) W6_RAW_WRITE(wr__);                                      \
    } while (0)
#define W6_SLOT_VMEM(k_) do { if ((k_) == 10 && !(W6_ABLATE & 2)) W6_LOAD_STREAM(); } while (0)
#define W6_PERIOD(PP_)                                                                              \
    do {                                                                                            \
        const f32x2 *rd__ = raw2 + ((PP_) ^ 1) * SLOT_F2;                                           \
        f32x2 *wr__ = raw2 + (PP_) * SLOT_F2;                                                       \
        _Pragma("unroll") for (int st = 0; st < NSTEP; ++st) {                                      \
            W6_SLOT_VALU(SPS * st, PP_);                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if (!(W6_ABLATE & 8)) W6_MFMA(PP_, st, 0);                                              \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            W6_SLOT_LDS(SPS * st, PP_);                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if (!(W6_ABLATE & 8)) W6_MFMA(PP_, st, 1);                                              \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if (SPS == 2) W6_SLOT_VALU(2 * st + 1, PP_);                                            \
            if (!(W6_ABLATE & 4)) W6_B_LOAD(st + W6_RING - 1);                                      \
            W6_SLOT_VMEM(SPS * st);                                                                 \
            if (SPS == 2) W6_SLOT_VMEM(2 * st + 1);                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            if (!(W6_ABLATE & 8)) W6_MFMA(PP_, st, 2);                                              \
            if (SPS == 2) {                                                                         \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                W6_SLOT_LDS(2 * st + 1, PP_);                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                  \
            }                                                                                       \
            if (!(W6_ABLATE & 8)) W6_MFMA(PP_, st, 3);                                              \
            __builtin_amdgcn_sched_barrier(0);                                                      \
        }                                                                                           \
        if (!(W6_ABLATE & 16)) W6_LDS_BARRIER();                                                    \
    } while (0)

    // ---- first item: prologue ----
    int v_cur = blockIdx.x;
    W6_DECODE(v_cur, cur_nb, cur_img, cur_ox0, cur_oy0);
    // The n-blocks of a tile run on neighbouring CUs of one XCD and read the same raw patch.  Started together they miss L2 together; the odd
    // n-blocks start W6_NB_DELAY x 64 cycles late (about a chunk period) and keep that distance -- every item takes the same time --, so their
    // patch loads find the lines the even n-block fetched a period earlier (weight loads queue behind the patch loads: vmcnt is in order).
    if (W6_NB_DELAY > 0 && p.nblk > 1 && ((cur_nb - p.nb0) & 1)) __builtin_amdgcn_s_sleep(W6_NB_DELAY);
    W6_MASKS(cur_ox0, cur_oy0, mk);
    W6_BASES(cur_img, cur_ox0, cur_oy0, ld_b0, ld_b1);
    nxt_b0 = ld_b0; nxt_b1 = ld_b1;
    has_next = false;                                                        // no switch inside the prologue (nchunks >= 4)
    nxt_nb = cur_nb; nxt_img = cur_img; nxt_ox0 = cur_ox0; nxt_oy0 = cur_oy0;
    mk_n[0] = mk[0]; mk_n[1] = mk[1]; mk_n[2] = mk[2];
    bo_c = cur_nb * nchunks * UC;            // (a 32-output block is the layer's last: every block before it has the full size)
    bo_n = bo_c + UCN;
    {
        W6_LOAD_STREAM();                                                    // chunk 0
        f32x4 keep[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) keep[i] = sr[i];
        W6_LOAD_STREAM();                                                    // chunk 1
#pragma unroll
        for (int u = 0; u < W6_RING - 1; ++u) W6_B_LOAD(u);
        W6_RAW_WRITE(raw2 + SLOT_F2);                                        // chunk 1 -> slot 1
#pragma unroll
        for (int i = 0; i < 3; ++i) sr[i] = keep[i];
        W6_RAW_WRITE(raw2);                                                  // chunk 0 -> slot 0
        W6_LOAD_STREAM();                                                    // chunk 2, stays in flight
    }
    __syncthreads();
#pragma unroll
    for (int col = 0; col < 8; ++col) { W6_ROW_READ(col, raw2); W6_ROW_FMA(col); }
#pragma unroll
    for (int part = 0; part < 4; ++part) W6_COLPASS(part, 0);
    W6_LDS_BARRIER();                                                        // slot 0 is read: period 0 overwrites it

    const bool tr_on = W6_TRACE && p.trace && p.nb0 >= 0 && blockIdx.x == 40 && (wave == 0 || wave == 4) && p.w8 == 77;
    long long *tr_buf = W6_TRACE ? (p.trace ? p.trace : reinterpret_cast<long long *>(p.out)) + (tr_on ? (wave >> 2) * 128 : 256) : nullptr;
    int tr_item = 0;
    int c = 0;                                                               // compute side: chunk of the current item
    bool more = true;
#define W6_ITEM_START()                                                                             \
    do {                                                                                            \
        has_next = v_cur + G < total;                                                               \
        if (has_next) {                                                                             \
            W6_DECODE(v_cur + G, nxt_nb, nxt_img, nxt_ox0, nxt_oy0);                                \
            W6_MASKS(nxt_ox0, nxt_oy0, mk_n);                                                       \
            W6_BASES(nxt_img, nxt_ox0, nxt_oy0, nxt_b0, nxt_b1);                                    \
        }                                                                                           \
        _Pragma("unroll") for (int b = 0; b < 8; ++b)                                               \
            _Pragma("unroll") for (int mt = 0; mt < NT; ++mt) acc[b][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; \
        /* the bias rides in the accumulators: A^T's column of the point x = 1 is all ones, so a constant added to M[xi = (1, 1)] comes out */ \
        /* of A^T M A as that constant in every output pixel -- wave 1 starts its b = 1 accumulators at the bias instead of zero */ \
        {                                                                                           \
            int z__ = 0;                                                                            \
            asm volatile("" : "+v"(z__));                                                           \
            const int q4__ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z__)) >> 4; \
            const float bsel__ = wave == 1 ? 1.f : 0.f;                                             \
            _Pragma("unroll") for (int mt = 0; mt < NT; ++mt)                                       \
                acc[1][mt] = *reinterpret_cast<const f32x4 *>(smem + BIAS_OFF + ((cur_nb - p.nb0) * 64 + 16 * mt + 4 * q4__) * 4) * bsel__; \
        }                                                                                           \
    } while (0)
    // bookkeeping around a period: the chunk that follows the compute side's (the next item's first after the last), then the period, then --
    // after an item's last chunk -- the output stage and the step to the block's next item
#define W6_RUN(PP_)                                                                                 \
    do {                                                                                            \
        bo_n = c + 1 < nchunks ? bo_c + UCN : (has_next ? nxt_nb : cur_nb) * nchunks * UC;          \
        W6_PERIOD(PP_);                                                                             \
        bo_c = bo_n;                                                                                \
        if (++c == nchunks) {                                                                       \
            W6_T(1);                                                                                \
            if (W6_ABLATE & 512) wino6_output<NT>(acc, smem, p, tid, lane, wave, cur_nb, first_img, first_ox0, first_oy0, tr_on, tr_item, tr_buf); /* (timing: every item of a block stores to its first item's place -- L2-resident) */ \
            else wino6_output<NT>(acc, smem, p, tid, lane, wave, cur_nb, cur_img, cur_ox0, cur_oy0, tr_on, tr_item, tr_buf); \
            if (!has_next) { more = false; }                                                        \
            else {                                                                                  \
                v_cur += G;                                                                         \
                cur_nb = nxt_nb; cur_img = nxt_img; cur_ox0 = nxt_ox0; cur_oy0 = nxt_oy0;           \
                c = 0;                                                                              \
                W6_ITEM_START();                                                                    \
                W6_T(8);                                                                            \
                ++tr_item;                                                                          \
                W6_T(0);                                                                            \
            }                                                                                       \
        }                                                                                           \
    } while (0)
    const int first_img = cur_img, first_ox0 = cur_ox0, first_oy0 = cur_oy0;
    W6_ITEM_START();
    W6_T(0);
    while (more) {
        W6_RUN(0);
        if (!more) break;
        W6_RUN(1);
    }
#undef W6_RUN
#undef W6_ITEM_START
#undef W6_PERIOD
#undef W6_SLOT_VMEM
#undef W6_SLOT_LDS
#undef W6_SLOT_VALU
#undef W6_MFMA
#undef W6_B_LOAD
#undef W6_COLPASS
#undef W6_ROW_FMA
#undef W6_ROW_READ
#undef W6_RAW_WRITE
#undef W6_LOAD_STREAM
#undef W6_BASES
#undef W6_MASKS
#undef W6_DECODE
}

bool wino6_supported(const ConvLaunch &p)
{
    if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo) return false;
    if (p.nseg > 1 && p.seg[1].pix_stride != p.seg[0].pix_stride) return false;
    if (((p.out_pix_stride | (int)p.out_chunk_stride) & 3) != 0 || (p.cout & 3) != 0) return false;   // 16-byte stores
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    if (nchunks < 4) return false;                                           // the load stream runs three chunks ahead of at most one item boundary
    if ((long)wino6_nblk(p.cout) * 64 > wino6::BIAS_MAX) return false;
    if ((double)wino6_nblk(p.cout) * nchunks * wino6::UC >= 2147483648.0) return false;             // 32-bit scalar offsets into the weights
    for (int i = 0; i < p.nseg; ++i)        // signed 32-bit scalar chunk offsets of the buffer loads
        if ((double)p.seg[i].nchunks * (double)p.seg[i].chunk_stride * 4.0 + 9.0 * p.W * p.seg[0].pix_stride * 4.0 >= 2147483648.0) return false;
    if ((double)((p.cout + 7) / 8) * (double)p.out_chunk_stride * 4.0 + (double)p.Ho * p.Wo * p.out_pix_stride * 4.0 >= 2147483648.0) return false;   // 31-bit store offsets (the range check drops anything beyond)
    return (double)p.H * p.W * p.seg[0].pix_stride * 4.0 < 2147483648.0;   // 32-bit byte offsets inside a plane
}

template <int NT>
static hipError_t launch_wino6_t(const ConvLaunch &p, int nb0, int nblk, hipStream_t s)
{
    using namespace wino6;
    static bool attr_done_dev[64] = {false};
    static int n_cu_dev[64] = {0};
    bool &attr_done = attr_done_dev[attr_slot()];
    int &n_cu = n_cu_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wino6<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
        n_cu &= ~7;                         // the XCD remap of the virtual block index wants a multiple of 8
        if (n_cu < 8) n_cu = 8;
        attr_done = true;
    }
    ConvLaunch q = p;
    q.wpk = reinterpret_cast<const float *>(p.wpk_w6);
    q.bias = p.bias_w6;
    q.nb0 = nb0;
    q.nblk = nblk;
    q.trace = nullptr;
#if W6_TRACE
    static long long *trace_dev = nullptr;
    static int traced = 0;
    const int tr_want = getenv("B2F_WINO_TRACE") ? atoi(getenv("B2F_WINO_TRACE")) : 0;
    const int nch = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    const bool do_trace = NT == 4 && tr_want == nch && traced < 1 && p.H * p.W >= 256 * 480;
    if (do_trace) {
        if (!trace_dev) hipMalloc(&trace_dev, 512 * sizeof(long long));
        hipMemsetAsync(trace_dev, 0, 512 * sizeof(long long), s);
    }
    if (!trace_dev) hipMalloc(&trace_dev, 512 * sizeof(long long));
    q.trace = trace_dev;                 // (every launch of a trace build stamps: untraced ones into the dummy area)
    q.w8 = do_trace ? 77 : 0;            // (the kernel does not read w8: marks the traced launch)
#endif
    const long items = (long)((p.Wo + OW - 1) / OW) * ((p.Ho + OH - 1) / OH) * p.nimg * nblk;
    const int pcap = p.w4_persist > 1 ? p.w4_persist : n_cu;               // (tests: exactly that many blocks)
    const int grid = (int)(items < pcap ? items : pcap);
    hipLaunchKernelGGL((conv3x3_wino6<NT>), dim3((unsigned)grid), dim3(512), LDS_BYTES, s, q);
#if W6_TRACE
    if (do_trace) {
        ++traced;
        long long h[256];
        hipStreamSynchronize(s);
        hipMemcpy(h, trace_dev, sizeof h, hipMemcpyDeviceToHost);
        for (int w = 0; w < 2; ++w) {
            fprintf(stderr, "wino6 trace, block 40 wave %d, %d chunks: per item, cycles: K loop (per chunk) | dump 0 | barrier | rounds 0 | barrier + dump 1 | barrier | rounds 1 | item switch\n", 4 * w, nch);
            for (int it = 0; it < 6; ++it) {
                const long long *t = h + w * 128 + it * 16;
                if (!t[0]) continue;
                fprintf(stderr, "  item %d: %7lld (%5lld) | %5lld | %5lld | %5lld | %5lld | %5lld | %5lld | %5lld\n", it, t[1] - t[0], (t[1] - t[0]) / nch, t[2] - t[1], t[3] - t[2], t[4] - t[3],
                        t[5] - t[4], t[6] - t[5], t[7] - t[6], t[8] - t[7]);
            }
        }
    }
#endif
    return hipGetLastError();
}

// the whole layer: blocks of 64 outputs, then -- when the outputs are <= 32 mod 64 -- one block of 32
hipError_t launch_conv3x3_wino6(const ConvLaunch &p, hipStream_t s)
{
    if (!wino6_supported(p) || !p.wpk_w6) return hipErrorInvalidValue;
    const int rem = p.cout % 64, n64 = p.cout / 64 + (rem > 32 ? 1 : 0);
    hipError_t e = hipSuccess;
    if (n64 > 0) e = launch_wino6_t<4>(p, 0, n64, s);
    if (e == hipSuccess && rem > 0 && rem <= 32) e = launch_wino6_t<2>(p, p.cout / 64, 1, s);
    return e;
}

int wino6_nblk(int cout) { return (cout + 63) / 64; }

size_t wino6_wpk_floats(int cin_chunks, int cout)
{
    const int rem = cout % 64;
    return (size_t)(cout / 64) * cin_chunks * (wino6::UC / 4) + (rem > 32 ? (size_t)cin_chunks * (wino6::UC / 4) : rem > 0 ? (size_t)cin_chunks * (wino6::UC / 8) : 0);
}

// U = G g G^T in double, rounded once to fp32 (row a = 0 negated: wave 0 forms -(B^T d)[0]).  Blocks of 64 outputs:
// [n-block][chunk][a 8][step 2 b + s][lane 64][mt 4], lane (co = 64 nb + 16 mt + (lane & 15), input channel = 8 chunk + 2 (lane >> 4) + s), xi = 8 a + b;
// a last block of <= 32 outputs: [chunk][a 8][b 8][lane 64][s 2][mt 2]
void wino6_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk)
{
    static const double G[8][3] = {{1, 0, 0},
                                   {-2.0 / 9, -2.0 / 9, -2.0 / 9},
                                   {-2.0 / 9, 2.0 / 9, -2.0 / 9},
                                   {1.0 / 90, 1.0 / 45, 2.0 / 45},
                                   {1.0 / 90, -1.0 / 45, 2.0 / 45},
                                   {32.0 / 45, 16.0 / 45, 8.0 / 45},
                                   {32.0 / 45, -16.0 / 45, 8.0 / 45},
                                   {0, 0, 1}};
    const int nblk = wino6_nblk(Co);
    std::vector<float> U((size_t)Co * Ci * 64);
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci) {
            const float *gk = w + ((size_t)co * Ci + ci) * 9;
            double t[8][3];
            for (int a = 0; a < 8; ++a)
                for (int v = 0; v < 3; ++v) t[a][v] = G[a][0] * gk[0 * 3 + v] + G[a][1] * gk[1 * 3 + v] + G[a][2] * gk[2 * 3 + v];
            for (int a = 0; a < 8; ++a)
                for (int bq = 0; bq < 8; ++bq)
                    U[((size_t)co * Ci + ci) * 64 + a * 8 + bq] = (float)(t[a][0] * G[bq][0] + t[a][1] * G[bq][1] + t[a][2] * G[bq][2]) * (a == 0 ? -1.f : 1.f);
        }
    auto u_at = [&](int co, int k, int xi) -> float {
        const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
        return (co < Co && ci >= 0) ? U[((size_t)co * Ci + ci) * 64 + xi] : 0.f;
    };
    const size_t blk_floats = (size_t)cin_chunks * (wino6::UC / 4);
    for (int nbk = 0; nbk < nblk; ++nbk) {
        float *dst = wpk + (size_t)nbk * blk_floats;
        const bool small = nbk == Co / 64 && Co % 64 <= 32;                 // (the last block of a layer whose outputs are <= 32 mod 64; Co % 64 > 0 there)
        for (int c = 0; c < cin_chunks; ++c)
            for (int a = 0; a < 8; ++a)
                for (int lane = 0; lane < 64; ++lane) {
                    const int kk = 2 * (lane >> 4), col = lane & 15;
                    if (!small) {
                        for (int st = 0; st < 16; ++st)
                            for (int mt = 0; mt < 4; ++mt)
                                dst[((((size_t)c * 8 + a) * 16 + st) * 64 + lane) * 4 + mt] = u_at(nbk * 64 + mt * 16 + col, c * kCK + kk + (st & 1), a * 8 + (st >> 1));
                    } else {
                        for (int bq = 0; bq < 8; ++bq)
                            for (int sk = 0; sk < 2; ++sk)
                                for (int mt = 0; mt < 2; ++mt)
                                    dst[((((size_t)c * 8 + a) * 8 + bq) * 64 + lane) * 4 + sk * 2 + mt] = u_at(nbk * 64 + mt * 16 + col, c * kCK + kk + sk, a * 8 + bq);
                    }
                }
    }
    for (int i = 0; i < nblk * 64; ++i) bpk[i] = i < Co ? b[i] : 0.f;
}

}  // namespace b2f
