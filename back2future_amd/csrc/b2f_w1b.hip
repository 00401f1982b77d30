// conv3x3 (stride 1) of the wide layers as a ONE-dimensional Winograd F(4,3) along x, direct along y, on the BF16 matrix pipe with
// exactly split fp32 operands, in a loader / consumer (wave-specialised) persistent block (round 5).  The layers are the stride-1
// convolutions of the convUnits and of the decoders, nn.SpatialConvolution(Ci,Co,3,3,1,1,1,1) + LeakyReLU(0.2) of
// /root/reference/models/pwc.lua:58-65,76-85.
//
// Why this form.  The fp32 "MFMA" of gfx950 runs at the vector rate and holds the SIMD's VALU, so the two-dimensional F(4x4)
// kernel (b2f_wino4.hip) adds its transform and load issue time to its multiply time (matrix pipe busy 0.62 for three rounds).  The
// bf16 pipe is 16x faster and leaves the VALU free, and an fp32 product costs six bf16 products (x = xh + xm + xl exactly, six
// of nine term products kept: b2f_convb.hip) -- but the 36 accumulator planes of F(4x4) cap a block at 32 tiles x 64 outputs, so
// its operand bytes per MAC (110 KB of weights per 1 728 matrix cycles) and its transform + split work cannot be amortised and
// round 4's three bf16 F(4x4) forms ended at parity.  Winograd along x only has SIX planes: 4.5 MACs per output instead of 2.25,
// which the bf16 pipe has to spare (4.5 x 6 / 16 = 1.69 fp32-MFMA-equivalent cycles per output against F(4x4)'s 2.25), a 2 x 1
// register tile of 32 x 32 MFMA tiles per wave, a one-dimensional transform, an output transform that stays in registers (no
// exchange through LDS), and smaller transform constants (rounding ~ 5x below F(4x4)'s).
//
//   m_xi[y][t][co] = sum_ky sum_ci V_xi[y + ky - 1][t][ci] U_xi[ky][ci][co]     xi = 0..5, t = tile of 4 output columns
//   V_xi[r][t][ci] = sum_j BT[xi][j] x[ci][r][4 t - 1 + j]                       (input transform along x, j = 0..5)
//   U_xi[ky][ci][co] = sum_kx G[xi][kx] w[co][ci][ky][kx]                        (host, double, rounded once, split in three bf16)
//   out[y][4 t + i][co] = sum_xi AT[i][xi] m_xi[y][t][co]                        (i = 0..3)
//
// GEMM view per (xi, ky) "step": D[co 32][(row, t) 32] += U[co][k 16] V[k 16][(row, t)], v_mfma_f32_32x32x16_bf16; a lane
// (column / row = lane & 31, kh = lane >> 5) holds channels 4 kh .. 4 kh + 3 of the 8-channel chunk as two windows of bf16 pairs
// Wa = [m01 m23 h01 h23], Wb = [h01 h23 l01 l23] (Xa, Xb alike): Wa Xa + Wb Xa + Wa Xb = the six kept term products.
//
// Block = 512 threads, one per CU, persistent over its tiles (16 rows x 32 columns x 64 outputs):
//   * waves 0-3, one per SIMD: CONSUMERS.  Wave (mw, nw) owns output rows 8 mw .. 8 mw + 7 (two pixel tiles of 4 rows x 8
//     column tiles) x the 32 outputs of N tile nw x 6 xi: 12 accumulators = 192 registers.  Per step: the weight windows from
//     L2 through a three-slot register ring (requested two steps ahead; the only loads in this wave's vmcnt queue), the pixel
//     windows from LDS (reloaded right behind the MFMAs that read them), 6 MFMAs.  No VALU work in the K loop.
//   * waves 4-7, one per SIMD: PRODUCERS.  They load the raw patch (18 rows x 34 columns of the 8-channel chunk, global ->
//     registers, two chunks ahead: HBM latency sits in THEIR vmcnt queue), transform it along x, split it into the bf16 windows
//     and write V[xi][window][kh][row 18][t 8] x 16 bytes into one of two LDS buffers (55 KB each).
//   * two barriers per chunk, neither of which the consumers normally wait at: B ("V of the next chunk is complete", consumers
//     pass it two steps before the end of the chunk and prefetch the first windows of the next chunk) and B' ("done reading this
//     chunk's V").  The chunk stream runs across tile boundaries; the producers work on the next tile while the consumers
//     run the output transform (registers only), bias, LeakyReLU and 16-byte stores.
#include "b2f_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

namespace b2f {
namespace w1b {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 16, TW = 32;                  // output tile of a block
constexpr int PR = TH + 2;                       // patch rows
constexpr int VPLANE = PR * 8;                   // u32x4 slots of one (xi, window, kh) plane: [row 18][t 8]
constexpr int VBUF = 6 * 2 * 2 * VPLANE;         // 3 456 slots = 55 296 bytes per chunk
constexpr int RAWBUF = 20 * 64;                  // u32x4 slots of one raw-patch buffer: 18 rows x 34 columns x (kh 2) = 1 224, in 20 DMA pieces
constexpr int XCH = 4 * 128;                     // u32x4 slots of the consumers' output exchange areas (2 KB per wave: one pixel pair of a pixel tile)
constexpr int LDS_BYTES = (2 * VBUF + 2 * RAWBUF + XCH) * 16;   // 110 592 + 40 960 + 8 192 = 159 744 of the 163 840
constexpr int NSTEP = 18;                        // (xi, ky)
constexpr int WSTEP = 2 * 2 * 2 * 32 * 16;       // bytes of one step: [window 2][N tile 2][kh 2][co 32] x 16
constexpr int WCHUNK = NSTEP * WSTEP;            // 73 728 bytes per (n-block, chunk)

#ifndef W1B_RING
#define W1B_RING 3     // slots of the weight ring (a divisor of 18); requested W1B_RING - 1 steps ahead
#endif
#ifndef B2F_W1B_ABLATE
#define B2F_W1B_ABLATE 0   // profiling only (wrong results): 1 no raw loads, 2 no weight loads, 4 no MFMAs, 8 no transform / split, 16 no V writes, 32 no pixel-window reads, 64 no epilogue stores, 128 one weight load per step instead of two, 256 every tile reads the patch of tile 0 (raw loads hit L2)
#endif

__device__ __forceinline__ unsigned pk(float a, float b)
{
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));   // one v_cvt_pk_bf16_f32 (RNE)
}
// v = h + m + l exactly, as the windows [m01 m23 h01 h23] and [h01 h23 l01 l23]
__device__ __forceinline__ void split(const f32x4 v, u32x4 &wa, u32x4 &wb)
{
    const unsigned h01 = pk(v[0], v[1]), h23 = pk(v[2], v[3]);
    const float r0 = v[0] - __builtin_bit_cast(float, h01 << 16), r1 = v[1] - __builtin_bit_cast(float, h01 & 0xffff0000u);
    const float r2 = v[2] - __builtin_bit_cast(float, h23 << 16), r3 = v[3] - __builtin_bit_cast(float, h23 & 0xffff0000u);
    const unsigned m01 = pk(r0, r1), m23 = pk(r2, r3);
    const float l0 = r0 - __builtin_bit_cast(float, m01 << 16), l1 = r1 - __builtin_bit_cast(float, m01 & 0xffff0000u);
    const float l2 = r2 - __builtin_bit_cast(float, m23 << 16), l3 = r3 - __builtin_bit_cast(float, m23 & 0xffff0000u);
    wa = u32x4{m01, m23, h01, h23};
    wb = u32x4{h01, h23, pk(l0, l1), pk(l2, l3)};
}

// Profiling only: -DB2F_W1B_TRACE=1 records s_memtime at five points of the first 64 chunks of block 8's consumer wave 0 and producer
// waves 4..7 (p.trace; the launcher prints them when B2F_W1B_TRACE=<chunks of the layer to trace> is in the environment)
#ifndef B2F_W1B_TRACE
#define B2F_W1B_TRACE 0
#endif
#if B2F_W1B_TRACE
#define W1B_STAMP(role_, v_, i_)                                                                                              \
    do {                                                                                                                      \
        if (p.trace && blockIdx.x == 8 && (v_) < 64 && (B2F_W1B_TRACE == 1 || ((role_) == 0 && (i_) == 0))) {                                                                        \
            const long long t__ = (long long)__builtin_amdgcn_s_memtime();                                                    \
            if (lane == 0) p.trace[(((role_) * 64 + (v_)) * 5 + (i_))] = t__;                                                 \
        }                                                                                                                     \
    } while (0)
#define W1B_ESTAMP(k_, i_)                                                                                                    \
    do {                                                                                                                      \
        if (p.trace && blockIdx.x == 8 && wave == 0 && (k_) < 4 && B2F_W1B_TRACE == 1) {                                      \
            const long long t__ = (long long)__builtin_amdgcn_s_memtime();                                                    \
            if (lane == 0) p.trace[5 * 64 * 5 + 2 + (k_) * 8 + (i_)] = t__;                                                   \
        }                                                                                                                     \
    } while (0)
#else
#define W1B_STAMP(role_, v_, i_) do { } while (0)
#define W1B_ESTAMP(k_, i_) do { } while (0)
#endif
#define W1B_MF(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)
#define W1B_BARRIER() asm volatile("s_barrier" ::: "memory")
#define W1B_LDS_DONE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct Item {
    int nb, img, ox0, oy0;
};

__global__ __launch_bounds__(512) void conv3x3_w1b(const ConvLaunch p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *V = reinterpret_cast<u32x4 *>(smem);                   // [buffer 2][xi 6][window 2][kh 2][row 18][t 8]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int total = tiles_x * tiles_y * p.nimg * p.nblk;
    const int G = (int)gridDim.x;
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    const int nitems = (total - (int)blockIdx.x + G - 1) / G;       // items of this block: blockIdx.x + k G
    const int nstream = nitems * nchunks;                          // chunks of this block's stream

    auto decode = [&](const int k) {
        Item it;
        int bid = xcd_remap((int)blockIdx.x + k * G, total);
        it.nb = bid % p.nblk;
        bid /= p.nblk;
        it.ox0 = (bid % tiles_x) * TW;
        bid /= tiles_x;
        it.oy0 = (bid % tiles_y) * TH;
        it.img = bid / tiles_y;
        return it;
    };

    if (wave >= 4) {
        // ============================================ PRODUCERS ============================================
        const int pw = wave - 4;
        // Raw patch of a chunk: 18 rows x 34 columns x (kh 2) x 16 bytes = 1 224 slots, brought by LDS-DMA in 20 pieces of 1 KB
        // (lane i of piece P = slot 64 P + i; consecutive slots are consecutive bytes of a patch row in memory: whole lines per
        // piece, where per-lane register loads of the six pixels of an item touched 32 lines per instruction and filled the CU's
        // vector-memory pipe together with the consumers' weight loads).  Producer pw issues pieces pw, pw + 4, ...
        const unsigned raw_lds = static_cast<unsigned>(reinterpret_cast<size_t>(smem)) + 2 * VBUF * 16;
        const u32x4 *RAW = V + 2 * VBUF;                            // [2][RAWBUF]
        // transform items: item 0 = patch rows 4 pw .. 4 pw + 3: lane = (kh, row 4, t 8); item 1 = rows 16, 17: lanes 0..31 = (kh, row 2, t 8),
        // shared by the four producers (see produce)
        const int t = lane & 7;
        const int row0 = 4 * pw + ((lane >> 3) & 3), kh0 = lane >> 5;
        const int row1 = 16 + ((lane >> 3) & 1), kh1 = (lane >> 4) & 1;
        const int dst0 = kh0 * VPLANE + row0 * 8 + t, dst1 = kh1 * VPLANE + row1 * 8 + t;
        const int src0 = row0 * 68 + 8 * t + kh0, src1 = row1 * 68 + 8 * t + kh1;   // + 2 j
        const bool act1 = lane < 32;

        int doff[5];                                                // byte offsets of this lane's five slots inside a chunk plane (or past the resource: zero)
        i32x4 rs0, rs1;
        int pk_item = -1;                                           // item the offsets belong to
        auto setup_item = [&](const int k) {
            Item it = decode(k);
            if (B2F_W1B_ABLATE & 256) { it.img = 0; it.ox0 = 0; it.oy0 = 0; }   // every tile reads the same patch: the raw loads hit L2
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int i = 64 * (pw + 4 * q) + lane;
                const int row = i / 68, rem = i - 68 * row;
                const int gy = it.oy0 - 1 + row, gx = it.ox0 - 1 + (rem >> 1);
                const bool ok = i < 18 * 68 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                doff[q] = ok ? (int)(((unsigned)(gy * p.W + gx) * (unsigned)p.seg[0].pix_stride + 4u * (rem & 1)) * 4u) : -16;   // 0xfffffff0 >= num_records: reads as zero
            }
            const unsigned long long b0 = reinterpret_cast<unsigned long long>(p.seg[0].ptr + (size_t)it.img * p.seg[0].img_stride);
            const unsigned long long b1 = reinterpret_cast<unsigned long long>(p.seg[1].ptr + (size_t)it.img * p.seg[1].img_stride);
            rs0 = i32x4{(int)(unsigned)b0, (int)((unsigned)(b0 >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
            rs1 = i32x4{(int)(unsigned)b1, (int)((unsigned)(b1 >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
            pk_item = k;
        };
        // stream position of the load side: chunk lc of item lk
        int lk = 0, lc = 0;
        auto request = [&](const int v) {
            if (lk != pk_item) setup_item(lk);
            const bool s1 = lc >= p.seg[0].nchunks;
            const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;
            // descriptor and scalar offset through v_readfirstlane: the "s" constraint alone does not make the compiler keep them in
            // SGPRs (it silently emitted VGPR operands); s_nop 4 = the wait states between v_readfirstlane and a VMEM read of the SGPR
            const int so = __builtin_amdgcn_readfirstlane((int)((s1 ? lc - p.seg[0].nchunks : lc) * cstr * 4));
            const i32x4 rsel = s1 ? rs1 : rs0;
            const i32x4 rs = {__builtin_amdgcn_readfirstlane(rsel[0]), __builtin_amdgcn_readfirstlane(rsel[1]), __builtin_amdgcn_readfirstlane(rsel[2]), __builtin_amdgcn_readfirstlane(rsel[3])};
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (!(B2F_W1B_ABLATE & 1)) {
                    const int ldst = __builtin_amdgcn_readfirstlane((int)(raw_lds + (unsigned)(((v & 1) * RAWBUF + 64 * (pw + 4 * q)) * 16)));
                    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(doff[q]), "s"(rs), "s"(ldst), "s"(so) : "memory");
                }
            if (++lc == nchunks) {                                  // past the end of the stream the last chunk is requested again (harmless)
                if (lk + 1 < nitems) { lc = 0; ++lk; } else lc = nchunks - 1;
            }
        };
        auto produce_one = [&](const u32x4 *src, u32x4 *dst, const bool active, auto xmask) {   // xmask: bit xi set = produce V_xi
            constexpr int XM = decltype(xmask)::value;
            // BT of F(4,3): rows (4 0 -5 0 1 0), (0 -4 -4 1 1 0), (0 4 -4 -1 1 0), (0 -2 -1 2 1 0), (0 2 -1 -2 1 0), (0 4 0 -5 0 1)
            f32x4 d[6], vv[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = __builtin_bit_cast(f32x4, src[2 * j]);
            if (B2F_W1B_ABLATE & 8) {
#pragma unroll
                for (int x = 0; x < 6; ++x) vv[x] = d[x];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float d0 = d[0][c], d1 = d[1][c], d2 = d[2][c], d3 = d[3][c], d4 = d[4][c], d5 = d[5][c];
                    const float P = __builtin_fmaf(-4.f, d2, d4), Q = __builtin_fmaf(-4.f, d1, d3);
                    const float R = d4 - d2, S = d3 - d1;
                    vv[0][c] = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
                    vv[1][c] = P + Q;
                    vv[2][c] = P - Q;
                    vv[3][c] = __builtin_fmaf(2.f, S, R);
                    vv[4][c] = __builtin_fmaf(-2.f, S, R);
                    vv[5][c] = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
                }
            }
#pragma unroll
            for (int x = 0; x < 6; ++x) {
                if (!((XM >> x) & 1)) continue;
                u32x4 a, b;
                if (B2F_W1B_ABLATE & 8) { a = __builtin_bit_cast(u32x4, vv[x]); b = a; }
                else split(vv[x], a, b);
                if (!(B2F_W1B_ABLATE & 16) && active) {
                    dst[(x * 2 + 0) * 2 * VPLANE] = a;
                    dst[(x * 2 + 1) * 2 * VPLANE] = b;
                }
            }
        };
        auto produce = [&](const int v) {
            u32x4 *dst = V + (v & 1) * VBUF;
            const u32x4 *src = RAW + (v & 1) * RAWBUF;
            produce_one(src + src0, dst + dst0, true, std::integral_constant<int, 63>());
            // patch rows 16, 17 (half a wave of items): every producer loads and transforms them and splits / writes its share of
            // the six xi (pairs that share sub-expressions together) -- with the whole item on one producer in turn, that one took
            // 3 400 cycles per chunk against the others' 1 900 and the consumers waited for it at B every chunk
#ifdef W1B_NO_BALANCE
            if ((v & 3) == pw) produce_one(src + src1, dst + dst1, act1, std::integral_constant<int, 63>());
            return;
#endif
            if (pw == 0) produce_one(src + src1, dst + dst1, act1, std::integral_constant<int, 0x06>());
            else if (pw == 1) produce_one(src + src1, dst + dst1, act1, std::integral_constant<int, 0x18>());
            else if (pw == 2) produce_one(src + src1, dst + dst1, act1, std::integral_constant<int, 0x01>());
            else produce_one(src + src1, dst + dst1, act1, std::integral_constant<int, 0x20>());
        };
#define W1B_DMA_DONE_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
        // Barriers of the stream (the consumers execute the same sequence): P0 "raw patch of chunk 0 is in LDS", A0 "V of chunk 0 is
        // complete (and the raw patch of chunk 1 has landed)", then per chunk v: B_v "V of chunk v + 1 complete, raw patch of chunk
        // v + 2 landed", B'_v "consumers are done reading V of chunk v".  A DMA piece is waited for by the wave that issued it
        // (vmcnt) before the barrier that publishes it.  Requests past the end of the stream fetch the last chunk again; the chunk
        // produced past its end lands in the free buffer.
        request(0);
        W1B_DMA_DONE_BARRIER();                                     // P0
        request(1);
        produce(0);
        W1B_DMA_DONE_BARRIER();                                     // A0
        for (int v = 0; v < nstream; ++v) {
            W1B_STAMP(1 + pw, v, 0);
            request(v + 2);                                         // raw buffer v & 1 is free: chunk v was produced from it before B_{v-1}
            W1B_STAMP(1 + pw, v, 1);
            produce(v + 1);
            W1B_STAMP(1 + pw, v, 2);
            W1B_DMA_DONE_BARRIER();                                 // B_v
            W1B_STAMP(1 + pw, v, 3);
            W1B_BARRIER();                                          // B'_v
            W1B_STAMP(1 + pw, v, 4);
        }
        return;
    }

    // ============================================== CONSUMERS ==============================================
    const int mw = wave & 1, nw = wave >> 1;
    const int j32 = lane & 31, kh = lane >> 5;
    const int rr = j32 >> 3, t = j32 & 7;
    const int x_lane = kh * VPLANE + (8 * mw + rr) * 8 + t;         // + ((xi 2 + window) 2) VPLANE + (4 mt + ky) 8
    const int w_lane = ((nw * 2 + kh) * 32 + j32) * 16;             // + window 2048
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.wpk_w1b)), 0, 0x7fffffff, 0x00020000);

    f32x16 acc[2][6];
    u32x4 wa[W1B_RING], wb[W1B_RING];
    u32x4 xa[2], xb[2];
    auto load_w = [&](const int slot, const int chunk_off, const int step) {
        if (B2F_W1B_ABLATE & 2) { wa[slot] = u32x4{1u, 2u, 3u, (unsigned)step}; wb[slot] = u32x4{4u, 5u, 6u, (unsigned)chunk_off}; return; }
#ifndef W1B_WAUX
#define W1B_WAUX 0      // cache policy of the weight loads: 0 default, 16 sc1 (served by L2, not allocated in L1), 2 nt, 18 sc1 nt
#endif
        wa[slot] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, chunk_off + step * WSTEP, W1B_WAUX));
        if (B2F_W1B_ABLATE & 128) { wb[slot] = wa[slot]; return; }        // one of the two weight loads only
        wb[slot] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, chunk_off + step * WSTEP + 2048, W1B_WAUX));
    };
    auto load_x = [&](const u32x4 *xbuf, const int mt, const int step, const int win) {
        const int xi = step / 3, ky = step - 3 * xi;
        if (B2F_W1B_ABLATE & 32) return u32x4{(unsigned)step, 2u, 3u, (unsigned)mt};
        return xbuf[((xi * 2 + win) * 2) * VPLANE + (4 * mt + ky) * 8];
    };

    int k = 0, c = 0;                                               // item / chunk of the compute side
    Item cur = decode(0);
    Item nxt = nitems > 1 ? decode(1) : cur;
    int w_cur = (cur.nb + p.nb0) * nchunks * WCHUNK;                // byte offset of the chunk being multiplied
    if (nstream > 0) {
#pragma unroll
        for (int s = 0; s < W1B_RING - 1; ++s) load_w(s, w_cur, s);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int x = 0; x < 6; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][x][r] = 0.f;
    W1B_BARRIER();                                                  // P0
    W1B_BARRIER();                                                  // A0
    if (nstream > 0) {
        const u32x4 *xb0 = V + x_lane;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { xa[mt] = load_x(xb0, mt, 0, 0); xb[mt] = load_x(xb0, mt, 0, 1); }
    }
    for (int v = 0; v < nstream; ++v) {
        const u32x4 *xcur = V + (v & 1) * VBUF + x_lane;
        const u32x4 *xnext = xcur;
        const bool last_chunk = c + 1 == nchunks;
        const int w_nxt = !last_chunk ? w_cur + WCHUNK : (k + 1 < nitems ? (nxt.nb + p.nb0) * nchunks * WCHUNK : w_cur);
        if (wave == 0) W1B_STAMP(0, v, 0);
#if B2F_W1B_TRACE
        if (p.trace && blockIdx.x == 8 && wave == 0 && (v == 16 || v == 47) && lane == 0) p.trace[5 * 64 * 5 + (v == 47)] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int xi = s / 3;
            if (s + W1B_RING - 1 < NSTEP) load_w((s + W1B_RING - 1) % W1B_RING, w_cur, s + W1B_RING - 1);
            else load_w((s + W1B_RING - 1) % W1B_RING, w_nxt, s + W1B_RING - 1 - NSTEP);
            if (s == NSTEP - 2) {
                if (wave == 0) W1B_STAMP(0, v, 1);
                W1B_BARRIER();                                      // B_v: V of chunk v + 1 is complete
                if (wave == 0) W1B_STAMP(0, v, 2);
            }
            if (s == NSTEP - 1) xnext = V + ((v + 1) & 1) * VBUF + x_lane;
            __builtin_amdgcn_sched_barrier(0);                      // the scheduler otherwise sinks every load to its first use
            // The six MFMAs of a step alternate between the two accumulators, and every other instruction sits between MFMAs on
            // DIFFERENT accumulators: one instruction between two dependent MFMAs costs ~43 cycles (the trace showed 40 cycles per
            // MFMA with the pixel-window reloads between the second and third MFMA of an accumulator, whatever the loads did).
            if (B2F_W1B_ABLATE & 4) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][xi][0] += __builtin_bit_cast(float, wa[s % W1B_RING][0] ^ xa[mt][1] ^ wb[s % W1B_RING][2] ^ xb[mt][3]);
            } else {
                W1B_MF(acc[0][xi], wa[s % W1B_RING], xa[0]);
                W1B_MF(acc[1][xi], wa[s % W1B_RING], xa[1]);
                W1B_MF(acc[0][xi], wb[s % W1B_RING], xa[0]);
                __builtin_amdgcn_sched_barrier(0);
                xa[0] = s + 1 < NSTEP ? load_x(xcur, 0, s + 1, 0) : load_x(xnext, 0, 0, 0);   // behind the MFMAs that read it: same registers
                __builtin_amdgcn_sched_barrier(0);
                W1B_MF(acc[1][xi], wb[s % W1B_RING], xa[1]);
                __builtin_amdgcn_sched_barrier(0);
                xa[1] = s + 1 < NSTEP ? load_x(xcur, 1, s + 1, 0) : load_x(xnext, 1, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                W1B_MF(acc[0][xi], wa[s % W1B_RING], xb[0]);
                __builtin_amdgcn_sched_barrier(0);
                xb[0] = s + 1 < NSTEP ? load_x(xcur, 0, s + 1, 1) : load_x(xnext, 0, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                W1B_MF(acc[1][xi], wa[s % W1B_RING], xb[1]);
                __builtin_amdgcn_sched_barrier(0);
                xb[1] = s + 1 < NSTEP ? load_x(xcur, 1, s + 1, 1) : load_x(xnext, 1, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (wave == 0) W1B_STAMP(0, v, 3);
        W1B_BARRIER();                                              // B'_v: done reading V of chunk v
        if (wave == 0) W1B_STAMP(0, v, 4);
        w_cur = w_nxt;
        if (++c == nchunks) {
            // ---- output transform AT = (1 1 1 1 1 0), (0 1 -1 2 -2 0), (0 1 1 4 4 0), (0 1 -1 8 -8 1), bias, LeakyReLU, stores.
            // Lane (column j32 = (row rr, tile t), kh) holds outputs 8 i + 4 kh + r of its N tile for i = 0..3, r = 0..3: stored
            // straight from there every store instruction would touch 32 lines with 32 bytes each (measured: half the kernel's
            // time).  Each (pixel tile, 8-output chunk i) goes through a 4 KB LDS area of the wave instead -- [row 4][slot 64] x 16
            // bytes, slot = (column 0..31, kh), rotated by the tile pair so that the 16-byte writes of a lane group fall into
            // different banks -- and leaves as four 1 KB rows of whole lines. ----
            W1B_ESTAMP(k, 0);
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)cur.img * p.out_img_stride, 0, 0x7fffffff, 0x00020000);
            int le = lane;
            asm volatile("" : "+v"(le));                            // recomputed here, not held in registers across the K loop
            const int e_kh = le >> 5, e_rr = (le >> 3) & 3, e_t = le & 7;
            // Exchange area of this wave: 2 KB = [row 4][slot 32] x 16 bytes for ONE pair of the four pixels of every column tile,
            // logical slot = (t, pixel of the pair j, kh), stored at slot ^ (t >> 2) (the eight lanes of a 16-byte write group then fall
            // into eight different bank groups; a read lane finds its slot inside the same aligned pair).  All lanes write (two pixels
            // each), all lanes read (rows 0, 1 then 2, 3): no exec-masked halves, no branch.
            u32x4 *xw = V + 2 * VBUF + 2 * RAWBUF + wave * 128;
            const int w_slot = e_rr * 32 + ((4 * e_t + e_kh) ^ (e_t >> 2));             // + 2 j (bit 1: untouched by the rotation)
            const int r_s = le & 31, r_half = le >> 5;              // read side: logical slot, row of the pair of rows
            const int r_slot = r_half * 32 + (r_s ^ (r_s >> 4));                        // + 64 for rows 2, 3
            const int s_t = r_s >> 2, s_j = (r_s >> 1) & 1, s_kh = r_s & 1;
            const int cob = (cur.nb + p.nb0) * 64 + 32 * nw;
            // store address = resource base (image) + scalar (chunk, row pair, pixel pair) + lane (row of the pair, column, half);
            // lanes below / right of the image or past the last output channel point past the resource (the store is dropped)
            const int oy_w = cur.oy0 + 8 * mw;
            const unsigned lane_off = ((unsigned)((oy_w + r_half) * p.Wo + cur.ox0 + 4 * s_t + s_j) * (unsigned)p.out_pix_stride + 4u * s_kh) * 4u;
            typedef unsigned long long u64;
            const u64 m_px0 = __builtin_amdgcn_ballot_w64(cur.ox0 + 4 * s_t + s_j < p.Wo), m_px1 = __builtin_amdgcn_ballot_w64(cur.ox0 + 4 * s_t + s_j + 2 < p.Wo);
            const u64 m_lo = __builtin_amdgcn_ballot_w64(r_half == 0), m_kh0 = __builtin_amdgcn_ballot_w64(s_kh == 0);
            const int row_bytes = p.Wo * p.out_pix_stride * 4, pair_bytes = 2 * p.out_pix_stride * 4;
            const float slope = p.leaky ? 0.2f : 1.f;               // max(v, 1 v) = v
            const f32x2 slope2 = {slope, slope};
            u32x4 rd[2][2];                                         // pipeline: the unit read while the previous one is stored
            f32x4 o[4];
#pragma unroll
            for (int h = 0; h <= 16; ++h) {                         // half-unit h = (pixel tile mt, 8-output chunk i, pixel pair pp)
                if (h < 16) {
                    const int mt = h >> 3, i = (h >> 1) & 3, pp = h & 1;
                    if (pp == 0) {
                        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias_w1b + cob + 8 * i + 4 * e_kh);   // padded to whole n-blocks
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {            // two outputs at a time: packed fp32 ops (no MFMA beside them here)
                            const int q = 4 * i + r;
                            const f32x2 m0 = {acc[mt][0][q], acc[mt][0][q + 1]}, m1 = {acc[mt][1][q], acc[mt][1][q + 1]}, m2 = {acc[mt][2][q], acc[mt][2][q + 1]};
                            const f32x2 m3 = {acc[mt][3][q], acc[mt][3][q + 1]}, m4 = {acc[mt][4][q], acc[mt][4][q + 1]}, m5 = {acc[mt][5][q], acc[mt][5][q + 1]};
                            const f32x2 bb = {bias[r], bias[r + 1]};
                            const f32x2 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                            f32x2 o0 = ((m0 + s12) + s34) + bb;
                            f32x2 o1 = __builtin_elementwise_fma(f32x2{2.f, 2.f}, d34, d12) + bb;
                            f32x2 o2 = __builtin_elementwise_fma(f32x2{4.f, 4.f}, s34, s12) + bb;
                            f32x2 o3 = (__builtin_elementwise_fma(f32x2{8.f, 8.f}, d34, d12) + m5) + bb;
                            o0 = __builtin_elementwise_max(o0, o0 * slope2);
                            o1 = __builtin_elementwise_max(o1, o1 * slope2);
                            o2 = __builtin_elementwise_max(o2, o2 * slope2);
                            o3 = __builtin_elementwise_max(o3, o3 * slope2);
                            o[0][r] = o0[0]; o[0][r + 1] = o0[1];
                            o[1][r] = o1[0]; o[1][r + 1] = o1[1];
                            o[2][r] = o2[0]; o[2][r + 1] = o2[1];
                            o[3][r] = o3[0]; o[3][r + 1] = o3[1];
                        }
                    }
                    xw[w_slot] = __builtin_bit_cast(u32x4, o[2 * pp]);
                    xw[w_slot + 2] = __builtin_bit_cast(u32x4, o[2 * pp + 1]);
                    // Other LANES wrote what this lane reads next: to the compiler a lane still holds what it loaded from the same address
                    // one unit earlier unless told that memory changed; the hardware side needs nothing, a wave's LDS operations execute
                    // in order (which also keeps the writes of the next unit behind these reads)
                    asm volatile("" ::: "memory");
                    rd[h & 1][0] = xw[r_slot];
                    rd[h & 1][1] = xw[r_slot + 64];
                    asm volatile("" ::: "memory");
                }
                if (h > 0) {                                        // stores of the previous half-unit
                    const int g = h - 1, mt = g >> 3, i = (g >> 1) & 3, pp = g & 1;
                    const int co_s = cob + 8 * i;                   // chunk of this store (+ 4 s_kh per lane)
                    const u64 m_co = co_s + 4 < p.cout ? ~0ull : (co_s < p.cout ? m_kh0 : 0ull);
                    const int s_chunk = (co_s >> 3) * (int)p.out_chunk_stride * 4 + pp * pair_bytes;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int row = 4 * mt + 2 * q;             // rows row, row + 1
                        const u64 m_row = oy_w + row + 1 < p.Ho ? ~0ull : (oy_w + row < p.Ho ? m_lo : 0ull);
                        const u64 m = m_co & m_row & (pp ? m_px1 : m_px0);
                        int v_off;
                        asm("v_cndmask_b32_e64 %0, -16, %1, %2" : "=v"(v_off) : "v"(lane_off), "s"(m));
                        if (!(B2F_W1B_ABLATE & 64)) __builtin_amdgcn_raw_buffer_store_b128(rd[g & 1][q], o_rsrc, v_off, s_chunk + row * row_bytes, 0);
                    }
                }
            }
            W1B_ESTAMP(k, 7);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int x = 0; x < 6; ++x)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][x][r] = 0.f;
            c = 0;
            ++k;
            cur = nxt;
            if (k + 1 < nitems) nxt = decode(k + 1);
        }
    }
}

}  // namespace w1b

bool w1b_supported(const ConvLaunch &p)
{
    if (p.stride != 1 || p.H != p.Ho || p.W != p.Wo || !p.wpk_w1b || !p.bias_w1b) return false;
    if (p.nseg > 1 && p.seg[1].pix_stride != p.seg[0].pix_stride) return false;
    if (((p.out_pix_stride | (int)p.out_chunk_stride) & 3) != 0 || (p.cout & 3) != 0) return false;   // 16-byte stores
    for (int i = 0; i < p.nseg; ++i) {
        if ((p.seg[i].pix_stride & 3) != 0 || (p.seg[i].chunk_stride & 3) != 0 || (p.seg[i].img_stride & 3) != 0) return false;  // 16-byte loads
        if ((double)p.seg[i].nchunks * (double)p.seg[i].chunk_stride * 4.0 >= 2147483648.0) return false;   // signed 32-bit scalar chunk offsets
    }
    if ((double)p.Ho * p.Wo * p.out_pix_stride * 4.0 + (double)((p.cout + 7) / 8) * (double)p.out_chunk_stride * 4.0 >= 2147483648.0) return false;   // 32-bit store offsets
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    if ((double)w1b_nblk(p.cout) * nchunks * (double)w1b::WCHUNK >= 2147483648.0) return false;
    return (double)p.H * p.W * p.seg[0].pix_stride * 4.0 < 2147483648.0;   // 32-bit byte offsets inside a plane
}

hipError_t launch_conv3x3_w1b(const ConvLaunch &p, hipStream_t s)
{
    using namespace w1b;
    if (!w1b_supported(p)) return hipErrorInvalidValue;
    static bool attr_done_dev[64] = {false};
    static int n_cu_dev[64] = {0};
    bool &attr_done = attr_done_dev[attr_slot()];
    int &n_cu = n_cu_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_w1b), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
        n_cu &= ~7;                         // the XCD remap of the virtual block index wants a multiple of 8
        if (n_cu < 8) n_cu = 8;
        attr_done = true;
    }
    ConvLaunch q = p;
    q.nblk = p.w1b_nblk > 0 ? p.w1b_nblk : w1b_nblk(p.cout);
    q.nb0 = 0;
    const int total = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH) * p.nimg * q.nblk;
    const int cap = p.w4_persist > 1 ? p.w4_persist : n_cu;          // tests: exactly that many blocks
    const int grid = total < cap ? total : cap;
#if B2F_W1B_TRACE
    static long long *trace_dev = nullptr;
    static int traced = 0;
    const int tr_want = getenv("B2F_W1B_TRACE") ? atoi(getenv("B2F_W1B_TRACE")) : 0;
    const int nch = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    const bool do_trace = tr_want > 0 && traced < 1 && nch == tr_want && p.H * p.W >= 256 * 480 && grid > 8;
    q.trace = nullptr;
    if (do_trace) {
        if (!trace_dev) hipMalloc(&trace_dev, (5 * 64 * 5 + 2 + 32) * sizeof(long long));
        hipMemsetAsync(trace_dev, 0, (5 * 64 * 5 + 2 + 32) * sizeof(long long), s);
        q.trace = trace_dev;
    }
#endif
    hipLaunchKernelGGL(w1b::conv3x3_w1b, dim3((unsigned)grid), dim3(512), LDS_BYTES, s, q);
#if B2F_W1B_TRACE
    if (do_trace) {
        ++traced;
        std::vector<long long> h(5 * 64 * 5 + 2 + 32);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), trace_dev, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        fprintf(stderr, "w1b trace, block 8, %d chunks per tile (s_memtime ticks = 100 MHz? scaled raw): consumer wave 0: chunk | steps 0-15 | wait B | steps 16-17 | wait B' | total ;  producers 4..7: request | produce | wait B | wait B'\n", nch);
        fprintf(stderr, "  shader clock over chunks 16..47: %.0f MHz (s_memtime ticks per 100 MHz s_memrealtime tick)\n",
                (double)(h[(0 * 64 + 47) * 5] - h[(0 * 64 + 16) * 5]) / ((double)(h[5 * 64 * 5 + 1] - h[5 * 64 * 5]) / 100.0));
        for (int k = 0; k < 3; ++k) {
            const long long *e = h.data() + 5 * 64 * 5 + 2 + k * 8;
            fprintf(stderr, "  epilogue of tile %d (consumer wave 0): start -> (mt 0, i 0) transform done %lld -> its first LDS write done %lld ... (mt 1, i 0) transform %lld -> write %lld ... end %lld\n",
                    k, e[1] - e[0], e[2] - e[0], e[4] - e[0], e[5] - e[0], e[7] - e[0]);
        }
        if (B2F_W1B_TRACE == 2) {
            fprintf(stderr, "  consumer wave 0, loop top to loop top:");
            for (int v = 17; v < 48; ++v) fprintf(stderr, " %lld", h[(0 * 64 + v) * 5] - h[(0 * 64 + v - 1) * 5]);
            fprintf(stderr, "\n");
        }
        for (int v = 0; v < 48 && B2F_W1B_TRACE == 1; ++v) {
            const long long *c = h.data() + (0 * 64 + v) * 5;
            fprintf(stderr, "  v=%2d  C: %6lld %6lld %6lld %6lld  tot %6lld |", v, c[1] - c[0], c[2] - c[1], c[3] - c[2], c[4] - c[3], c[4] - c[0]);
            for (int r = 1; r <= 4; ++r) {
                const long long *t = h.data() + (r * 64 + v) * 5;
                fprintf(stderr, "  P%d: %5lld %5lld %5lld %5lld |", r - 1, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3]);
            }
            fprintf(stderr, "\n");
        }
    }
#endif
    return hipGetLastError();
}

int w1b_nblk(int cout) { return (cout + 63) / 64; }
size_t w1b_wpk_floats(int cin_chunks, int cout) { return (size_t)w1b_nblk(cout) * cin_chunks * (w1b::WCHUNK / 4); }

static inline unsigned short w1b_bf16_rne(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float w1b_bf16_f32(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// U_xi[ky] = sum_kx G[xi][kx] w[.][.][ky][kx] in double, rounded once to fp32, split u = h + m + l exactly;
// [n-block of 64][chunk][step = 3 xi + ky][window 2][N tile 2][kh 2][co 32] x 8 bf16: Wa = (m0..3 h0..3), Wb = (h0..3 l0..3) of the
// channels 8 chunk + 4 kh + 0..3; bpk: bias padded to whole blocks
void w1b_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk)
{
    static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    const int nblk = w1b_nblk(Co);
    unsigned short *out = reinterpret_cast<unsigned short *>(wpk);
    for (int nbk = 0; nbk < nblk; ++nbk)
        for (int c = 0; c < cin_chunks; ++c)
            for (int step = 0; step < w1b::NSTEP; ++step) {
                const int xi = step / 3, ky = step % 3;
                for (int nt = 0; nt < 2; ++nt)
                    for (int h = 0; h < 2; ++h)
                        for (int nn = 0; nn < 32; ++nn) {
                            const size_t base = (((size_t)(nbk * cin_chunks + c) * w1b::NSTEP + step) * 2) * 2 * 2 * 32;   // 16-byte units, window 0
                            unsigned short *qa = out + (base + (size_t)((0 * 2 + nt) * 2 + h) * 32 + nn) * 8;
                            unsigned short *qb = out + (base + (size_t)((1 * 2 + nt) * 2 + h) * 32 + nn) * 8;
                            for (int j = 0; j < 4; ++j) {
                                const int co = nbk * 64 + nt * 32 + nn;
                                const int k = c * kCK + h * 4 + j;
                                const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                                float v = 0.f;
                                if (co < Co && ci >= 0) {
                                    const float *g = w + ((size_t)co * Ci + ci) * 9 + ky * 3;
                                    v = (float)(G[xi][0] * (double)g[0] + G[xi][1] * (double)g[1] + G[xi][2] * (double)g[2]);
                                }
                                const unsigned short hh = w1b_bf16_rne(v);
                                const float r1 = v - w1b_bf16_f32(hh);
                                const unsigned short mm = w1b_bf16_rne(r1);
                                const float r2 = r1 - w1b_bf16_f32(mm);
                                const unsigned short ll = w1b_bf16_rne(r2);
                                qa[j] = mm; qa[4 + j] = hh;
                                qb[j] = hh; qb[4 + j] = ll;
                            }
                        }
            }
    for (int i = 0; i < nblk * 64; ++i) bpk[i] = i < Co ? b[i] : 0.f;
}

}  // namespace b2f
