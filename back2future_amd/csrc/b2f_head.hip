// Head of the feature pyramid, two layers in ONE streaming kernel on the BF16 matrix pipe with exactly split fp32 operands (round 4):
//   nn.SpatialConvolution(16,16,3,3,1,1,1,1) + LeakyReLU(0.2)     second conv of the level-2 convUnit   (/root/reference/models/pwc.lua:62)
//   nn.SpatialConvolution(16,32,3,3,2,2,1,1) + LeakyReLU(0.2)     first conv of the level-3 convUnit    (/root/reference/models/pwc.lua:60)
// The 16-channel map between them (H/2 x W/2 x 3 frames: 1.5 GB written and read back per 16-triplet step at 1024 x 1920) is the
// largest tensor of the whole graph and nothing else reads it (the decoder starts at level 3).  With the multiplications on the bf16
// pipe (b2f_conv16b.hip: the 16 -> 16 layer alone went from 0.94 ms, bound by the fp32 MFMA, to 0.61 ms, bound by HBM) the two layers
// together are memory-shaped, so the intermediate never leaves the CU:
//   * a block owns a vertical strip of 30 output columns (61 columns of the intermediate, 63 of the input) and walks DOWN it, one
//     output row per step: two new input rows are staged, two new intermediate rows computed, one output row produced -- no vertical
//     halo is recomputed, the horizontal one costs 63 / 60;
//   * the operands live in LDS as rolling rows, already split: per (pixel, channel quad) three 8-byte pairs M = (m01 m23), H, L
//     ([row][plane 3][kg 4][64 columns], column slot rotated by 16 kg so that a 32-lane half of a ds_read touches every bank once);
//     an MFMA operand window is two of them (Xa = M|H, Xb = H|L).  Input ring 6 rows, intermediate ring 4 rows: 60 KB, two blocks per CU;
//   * step t:   [split + write input rows 2t+4, 2t+5 (requested a step earlier) | request the next two | conv1: rows 2t, 2t+1 of the
//     intermediate from input rows 2t..2t+3: wave w = 16-column tile w, both rows, 54 MFMAs; bias, LeakyReLU, zero outside the image
//     (it is conv2's padding), split, write]   barrier   [conv2: output row from intermediate rows 2t-1..2t+1, wave = (column tile,
//     16-output tile), 27 MFMAs, store]   barrier.   Intermediate columns are stored even | odd so that the stride-2 taps of conv2 read
//     consecutive slots.
// Arithmetic: v_mfma_f32_16x16x32_bf16, D[co 16][pixel 16], three MFMAs per tap (Wa Xa, Wb Xa, Wa Xb = six of the nine term products,
// fp32 accumulation; see b2f_conv16b.hip).  Weights: the fp32 packings of the two fp32 kernels ([tap][kg][co][4], [tap][c2][kg][co][4]),
// split in the prologue into 144 VGPRs.  A triplet's result does not depend on the batch or on how the rows are cut into blocks.
#include "b2f_internal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace b2f {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace head {
constexpr int WO = 30;                  // output columns per strip
constexpr int WI = 2 * WO + 1;          // 61 intermediate columns
constexpr int WC = WI + 2;              // 63 input columns
constexpr int CROWS = 6, IROWS = 6;     // ring depths
constexpr int ROW_U2 = 3 * 4 * 64;      // u32x2 per ring row: [plane M H L][kg][column slot]
constexpr int LDS_BYTES = (CROWS + IROWS) * ROW_U2 * 8;   // 73 728: two blocks per CU
}  // namespace head

__device__ __forceinline__ unsigned head_pk(float a, float b)
{
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));   // one v_cvt_pk_bf16_f32 (RNE)
}
// fp32 quad -> pairs M = (m01 m23), H = (h01 h23), L = (l01 l23), v = h + m + l exactly
__device__ __forceinline__ void head_split(const f32x4 v, u32x2 &M, u32x2 &H, u32x2 &Lo)
{
    const unsigned h01 = head_pk(v[0], v[1]), h23 = head_pk(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, h01 << 16), r1 = v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h23 << 16), r3 = v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u);
    const unsigned m01 = head_pk(r0, r1), m23 = head_pk(r2, r3);
    const float l0 = r0 - __builtin_bit_cast(float, m01 << 16), l1 = r1 - __builtin_bit_cast(float, m01 & 0xffff0000u);
    const float l2 = r2 - __builtin_bit_cast(float, m23 << 16), l3 = r3 - __builtin_bit_cast(float, m23 & 0xffff0000u);
    M = u32x2{m01, m23};
    H = u32x2{h01, h23};
    Lo = u32x2{head_pk(l0, l1), head_pk(l2, l3)};
}

#ifndef B2F_HEAD_TRACE
#define B2F_HEAD_TRACE 0     // profiling builds: clock stamps of steps 30..33 of block 0 (B2F_WINO_TRACE=1 prints them)
#endif
#ifndef B2F_HEAD_ABLATE
#define B2F_HEAD_ABLATE 0    // profiling only (wrong results): 1 no MFMAs, 2 no global loads, 4 no staging split / writes, 8 no conv1 epilogue split / writes
#endif
#if B2F_HEAD_ABLATE & 1
#define HEAD_MF(acc_, a_, b_) do { acc_[0] += __builtin_bit_cast(float, (a_)[0] ^ (b_)[0] ^ (a_)[3] ^ (b_)[3]); } while (0)
#else
#define HEAD_MF(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)
#endif

__global__ __launch_bounds__(256, 2) void conv_head16_kernel(const HeadLaunch p)
{
    using namespace head;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x2 *C0 = reinterpret_cast<u32x2 *>(smem);                  // [CROWS][3][4][64]
    u32x2 *IN = C0 + CROWS * ROW_U2;                              // [IROWS][3][4][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;

    int bid = blockIdx.x;
    const int sx = bid % p.nsx;
    bid /= p.nsx;
    const int sy = bid % p.nsy;
    const int img = bid / p.nsy;
    const int X0 = sx * WO, y0 = sy * p.rows_per_block;
    const int nrows = min(p.rows_per_block, p.Ho - y0);
    const int crow0 = 2 * y0 - 3;                                 // input row of ring index 0
    const int irow0 = 2 * y0 - 2;                                 // intermediate row of ring index 0

    // ---- weight windows: conv1 (co = n, kg), conv2 (co = 16 c2 + n, kg) with c2 = wave >> 1 ----
    u32x4 wa1[9], wb1[9], wa2[9], wb2[9];
    {
        const f32x4 *wp1 = reinterpret_cast<const f32x4 *>(p.w1) + kg * 16 + n;
        const f32x4 *wp2 = reinterpret_cast<const f32x4 *>(p.w2) + ((wave >> 1) * 4 + kg) * 16 + n;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            u32x2 M, H, Lo;
            head_split(wp1[t * 64], M, H, Lo);
            wa1[t] = u32x4{M[0], M[1], H[0], H[1]};
            wb1[t] = u32x4{H[0], H[1], Lo[0], Lo[1]};
            head_split(wp2[t * 128], M, H, Lo);
            wa2[t] = u32x4{M[0], M[1], H[0], H[1]};
            wb2[t] = u32x4{H[0], H[1], Lo[0], Lo[1]};
        }
    }
    const f32x4 bias1 = *reinterpret_cast<const f32x4 *>(p.b1 + 4 * kg);
    const f32x4 bias2 = *reinterpret_cast<const f32x4 *>(p.b2 + 16 * (wave >> 1) + 4 * kg);

    // ---- staging: thread < 252 = (chunk, input column, half); two rows per step ----
    const bool s_on = tid < 252;
    const int s_chunk = tid >= 126 ? 1 : 0, s_rem = tid - 126 * s_chunk;
    const int s_col = s_rem >> 1, s_half = s_rem & 1, s_kg = 2 * s_chunk + s_half;
    const int s_cx = 2 * X0 - 2 + s_col;
    const bool s_colok = s_on && s_cx >= 0 && s_cx < p.W1;
    const float *s_base = p.in + (size_t)img * p.in_img_stride + (size_t)s_chunk * p.in_chunk_stride + (size_t)(s_colok ? s_cx : 0) * p.in_pix_stride + s_half * 4;
    const int s_slot = s_kg * 64 + ((s_col + 16 * s_kg) & 63);
    f32x4 raw[2];
    auto request = [&](int k) {                                   // input ring rows 2k + 4, 2k + 5
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = crow0 + 2 * k + 4 + j;
            raw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!(B2F_HEAD_ABLATE & 2) && s_colok && row >= 0 && row < p.H1) raw[j] = *reinterpret_cast<const f32x4 *>(s_base + (size_t)row * p.W1 * p.in_pix_stride);
        }
    };
    auto stage_write = [&](const int rs0) {                          // ring slots rs0, rs0 + 1 (compile-time at every call)
        if (s_on && !(B2F_HEAD_ABLATE & 4)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u32x2 M, H, Lo;
                head_split(raw[j], M, H, Lo);
                u32x2 *d = C0 + (rs0 + j) * ROW_U2 + s_slot;
                d[0] = M;
                d[256] = H;
                d[512] = Lo;
            }
        }
    };

    // ---- conv1: wave = 16-column tile of the intermediate; lane (n, kg) ----
    const int i1 = 16 * wave + n;                                  // intermediate column (valid < 61)
    const int ci = 2 * X0 - 1 + i1;                                // its image column
    const bool i_colok = i1 < WI && ci >= 0 && ci < p.W1;
    const int i_wslot = kg * 64 + (((i1 >> 1) + 32 * (i1 & 1) + 16 * kg) & 63);   // even | odd column slots
    int c_rslot[3];                                                // input column slots of the three kx taps
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) c_rslot[kx] = kg * 64 + ((i1 + kx + 16 * kg) & 63);
    // ---- conv2: wave = (column tile wave & 1, output tile wave >> 1) ----
    const int x2 = 16 * (wave & 1) + n;                            // output column in the strip (valid < 30)
    int i_rslot[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) i_rslot[kx] = kg * 64 + ((x2 + (kx >> 1) + 32 * (kx & 1) + 16 * kg) & 63);
    const bool o_colok = x2 < WO && X0 + x2 < p.Wo;
    float *o_base = p.out + (size_t)img * p.out_img_stride + (size_t)(2 * (wave >> 1) + (kg >> 1)) * p.out_chunk_stride +
                    (size_t)(X0 + (o_colok ? x2 : 0)) * p.out_pix_stride + (kg & 1) * 4;

    request(-2);
    stage_write(0);
    request(-1);
    stage_write(2);
    request(0);
    __syncthreads();

    // operand windows of one (row, column slot): the three pairs as ONE 192-bit register tuple, Xa = dwords 0..3, Xb = dwords 2..5
    // (sub-ranges of the tuple: no copies)
    typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
    auto fetch = [&](const u32x2 *row, const int slot) __attribute__((always_inline)) {
        const u32x2 M = row[slot], H = row[256 + slot], Lo = row[512 + slot];
        return u32x6{M[0], M[1], H[0], H[1], Lo[0], Lo[1]};
    };
#define HEAD_XA(v_) __builtin_shufflevector(v_, v_, 0, 1, 2, 3)
#define HEAD_XB(v_) __builtin_shufflevector(v_, v_, 2, 3, 4, 5)

    // One step; PH = t % 3 makes every ring slot a compile-time constant (both rings hold 6 rows and advance 2 per step), so the LDS
    // addresses are per-lane bases + immediate offsets (no address arithmetic, no modulo in the loop).
    //   stage input rows 2t + 4, 2t + 5 | conv1(t): intermediate rows 2t, 2t + 1 from input rows 2t .. 2t + 3 | conv2(t - 1): output row
    //   y0 + t - 2 from intermediate rows 2t - 3 .. 2t - 1 (written before the last barrier) | ONE barrier
#if B2F_HEAD_TRACE
    const bool ts_on = p.trace && blockIdx.x == 0 && lane == 0;
#define HEAD_TS(k_) do { __builtin_amdgcn_sched_barrier(0); if (ts_on && t >= 30 && t < 34) p.trace[((t - 30) * 4 + wave) * 8 + (k_)] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HEAD_TS(k_) do {} while (0)
#endif
    auto step = [&](auto phv, const int t) __attribute__((always_inline)) {
        constexpr int PH = decltype(phv)::value;
        HEAD_TS(0);
        stage_write((2 * PH + 4) % 6);
        request(t + 1);
        HEAD_TS(1);
        {
            f32x4 accA[3], accB[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) { accA[kx] = f32x4{0.f, 0.f, 0.f, 0.f}; accB[kx] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            u32x6 xv[2][3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) xv[0][kx] = fetch(C0 + ((2 * PH) % 6) * ROW_U2, c_rslot[kx]);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                __builtin_amdgcn_sched_barrier(0);
                if (rr < 3) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) xv[(rr + 1) & 1][kx] = fetch(C0 + ((2 * PH + rr + 1) % 6) * ROW_U2, c_rslot[kx]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const u32x4 xa = HEAD_XA(xv[rr & 1][kx]), xb = HEAD_XB(xv[rr & 1][kx]);
                    if (rr <= 2) {                                 // row A = intermediate ring row 2t: tap ky = rr
                        HEAD_MF(accA[kx], wa1[rr * 3 + kx], xa);
                        HEAD_MF(accA[kx], wb1[rr * 3 + kx], xa);
                        HEAD_MF(accA[kx], wa1[rr * 3 + kx], xb);
                    }
                    if (rr >= 1) {                                 // row B = 2t + 1: tap ky = rr - 1
                        HEAD_MF(accB[kx], wa1[(rr - 1) * 3 + kx], xa);
                        HEAD_MF(accB[kx], wb1[(rr - 1) * 3 + kx], xa);
                        HEAD_MF(accB[kx], wa1[(rr - 1) * 3 + kx], xb);
                    }
                }
            }
            HEAD_TS(2);
            const f32x4 acc[2] = {(accA[0] + accA[1]) + accA[2], (accB[0] + accB[1]) + accB[2]};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ir = irow0 + 2 * t + j;                  // image row of this intermediate row
                f32x4 o4 = acc[j] + bias1;
                o4 = __builtin_elementwise_max(o4, 0.2f * o4);
                if (!(i_colok && ir >= 0 && ir < p.H1)) o4 = f32x4{0.f, 0.f, 0.f, 0.f};   // conv2's zero padding
                if (B2F_HEAD_ABLATE & 8) { if (o4[0] == 1234.5f) IN[i_wslot] = u32x2{1u, 2u}; continue; }
                u32x2 M, H, Lo;
                head_split(o4, M, H, Lo);
                u32x2 *d = IN + ((2 * PH + j) % 6) * ROW_U2 + i_wslot;
                d[0] = M;
                d[256] = H;
                d[512] = Lo;
            }
        }
        HEAD_TS(3);
        if (t >= 2) {
            f32x4 acc3[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc3[kx] = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x6 xv[2][3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) xv[0][kx] = fetch(IN + ((2 * PH + 3) % 6) * ROW_U2, i_rslot[kx]);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                __builtin_amdgcn_sched_barrier(0);
                if (ky < 2) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) xv[(ky + 1) & 1][kx] = fetch(IN + ((2 * PH + 4 + ky) % 6) * ROW_U2, i_rslot[kx]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const u32x4 xa = HEAD_XA(xv[ky & 1][kx]), xb = HEAD_XB(xv[ky & 1][kx]);
                    HEAD_MF(acc3[kx], wa2[ky * 3 + kx], xa);
                    HEAD_MF(acc3[kx], wb2[ky * 3 + kx], xa);
                    HEAD_MF(acc3[kx], wa2[ky * 3 + kx], xb);
                }
            }
            HEAD_TS(4);
            const f32x4 acc = (acc3[0] + acc3[1]) + acc3[2];
            f32x4 o4 = acc + bias2;
            o4 = __builtin_elementwise_max(o4, 0.2f * o4);
            const int oy = y0 + t - 2;
            if (o_colok) *reinterpret_cast<f32x4 *>(o_base + (size_t)oy * p.Wo * p.out_pix_stride) = o4;
        }
        HEAD_TS(5);
        __syncthreads();
        HEAD_TS(6);
    };
    for (int t = 0; t <= nrows + 1; t += 3) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 <= nrows + 1) step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 <= nrows + 1) step(std::integral_constant<int, 2>{}, t + 2);
    }
#undef HEAD_XA
#undef HEAD_XB
}

bool head16_supported(const HeadLaunch &p)
{
    return p.Ho == (p.H1 - 1) / 2 + 1 && p.Wo == (p.W1 - 1) / 2 + 1 && (p.in_pix_stride & 3) == 0 && (p.in_chunk_stride & 3) == 0 &&
           (p.out_pix_stride & 3) == 0 && (p.out_chunk_stride & 3) == 0;
}

hipError_t launch_conv_head16(HeadLaunch p, hipStream_t s)
{
    using namespace head;
    static bool attr_done_dev[64] = {false};
    bool &attr_done = attr_done_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_head16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    p.nsx = (p.Wo + WO - 1) / WO;
    // cut every strip into row blocks so that the launch has at least ~6 blocks per CU slot pair (two blocks per CU); a block
    // recomputes one intermediate row at its top, so not finer than 16 rows
    int nsy = 1;
    while ((long)p.nimg * p.nsx * nsy < 1536 && (p.Ho + nsy - 1) / nsy > 16) ++nsy;
    p.rows_per_block = (p.Ho + nsy - 1) / nsy;
    p.nsy = (p.Ho + p.rows_per_block - 1) / p.rows_per_block;
#if B2F_HEAD_TRACE
    static long long *trace_dev = nullptr;
    static int traced = 0;
    const bool do_trace = getenv("B2F_WINO_TRACE") && !traced && p.rows_per_block >= 40;
    p.trace = nullptr;
    if (do_trace) {
        if (!trace_dev) (void)hipMalloc(&trace_dev, 4 * 4 * 8 * sizeof(long long));
        (void)hipMemsetAsync(trace_dev, 0, 4 * 4 * 8 * sizeof(long long), s);
        p.trace = trace_dev;
    }
#endif
    hipLaunchKernelGGL(conv_head16_kernel, dim3((unsigned)(p.nimg * p.nsx * p.nsy)), dim3(256), LDS_BYTES, s, p);
#if B2F_HEAD_TRACE
    if (do_trace) {
        traced = 1;
        (void)hipStreamSynchronize(s);
        long long h[4 * 4 * 8];
        (void)hipMemcpy(h, trace_dev, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "conv_head16 step trace, block 0, steps 30..33, cycles since wave 0's stamp 0 of step 30:\n"
                        "  0 step starts | 1 staged rows split + written, next rows requested | 2 conv1 MFMAs issued | 3 intermediate rows split + written | 4 conv2 MFMAs issued | 5 output stored | 6 barrier passed\n");
        for (int t = 0; t < 4; ++t)
            for (int w = 0; w < 4; ++w) {
                fprintf(stderr, "  step %d wave %d:", 30 + t, w);
                for (int k = 0; k < 7; ++k) fprintf(stderr, " %7lld", h[(t * 4 + w) * 8 + k] - h[0]);
                fprintf(stderr, "\n");
            }
    }
#endif
    return hipGetLastError();
}

}  // namespace b2f
