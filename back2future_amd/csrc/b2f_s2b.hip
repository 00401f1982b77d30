// The stride-2 first convs of the convUnits, nn.SpatialConvolution(Ci,Co,3,3,2,2,1,1) + LeakyReLU(0.2) of /root/reference/models/pwc.lua:60
// (32 -> 64, 64 -> 96, 96 -> 128, 128 -> 192), as a direct implicit GEMM on the BF16 matrix pipe with exactly split fp32 operands
// (x = xh + xm + xl, six of nine term products: b2f_convb.hip) in the loader / consumer persistent block of b2f_w1b.hip (round 5).
//
// These layers are small GEMMs over large maps: 9 x Ci MACs per output against 32 x Ci + 4 x Co bytes of compulsory traffic per output
// pixel -- HBM floor 0.38 ms for the four of them at batch 16 x 3x1024x1920, matrix pipe 0.15 ms; conv3x3_bf6 (b2f_convb.hip) takes
// 1.07 ms because a block brings its patch global -> registers -> LDS with nothing else to do meanwhile, once per 64 outputs, and at
// small maps (batch 1: 18 blocks for 128 -> 192 at 32 x 60) runs 16 chunks back to back on an empty chip.  Here:
//   * a block computes ALL outputs of a tile of 8 x 16 output pixels (the patch is read from HBM once, by LDS-DMA in 1 KB pieces, two
//     chunks ahead, by the producer waves; small tiles keep small maps parallel);
//   * waves 4-7 produce: raw patch (17 x 33 pixels x 8 channels) -> split into the bf16 windows [m01 m23 h01 h23], [h01 h23 l01 l23] ->
//     V[window][kh][row][column parity][column / 2] in LDS (even | odd columns apart: a tap's 16 pixels are consecutive slots);
//   * waves 0-3 consume: wave = (mw, nw), MTC pixel tiles (2 rows x 16 columns each) x one tile of 32 outputs; per tap the weight
//     windows from L2 through a three-slot register ring, the pixel windows from LDS one tap ahead, 3 MTC MFMAs; no VALU in the K loop;
//   * two barriers per 8-channel chunk as in b2f_w1b.hip; the epilogue stores straight from the accumulators (a lane holds four
//     consecutive outputs of one pixel, 16 lanes cover 16 consecutive pixels: 512-byte runs).
#include "b2f_internal.h"

#include <cstring>
#include <vector>

namespace b2f {
namespace s2b {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16;                   // output tile of a block
constexpr int PR = 2 * TH + 1, PC = 2 * TW + 1;  // patch: 17 rows x 33 columns
constexpr int RS = 40;                           // V row stride in slots (>= 34 = 2 x 17 half columns; a multiple of 8: the two rows of a pixel tile fall on the same banks)
constexpr int VPL = PR * RS;                     // slots of one (window, kh) plane
constexpr int VBUF = 4 * VPL;                    // 2 720 slots = 43 520 bytes per chunk
constexpr int NRAW = PR * PC * 2;                // 1 122 raw slots: [row][column][kh] x 16 bytes
constexpr int NPIECE = (NRAW + 63) / 64;         // 18 DMA pieces of 1 KB
constexpr int RAWBUF = NPIECE * 64;
constexpr int NRING = 4;                         // raw-patch buffers: the DMA runs THREE chunks ahead of the split, two whole chunk periods between request and use
                                                 // (a chunk of these layers is short: ~1 700 matrix cycles, a few hundred in a small launch's one-tile blocks --
                                                 // less than a memory round trip; three buffers, one period of cover, made a single triplet's coarse levels wait)
constexpr int LDS_BYTES = (2 * VBUF + NRING * RAWBUF) * 16;   // 87 040 + 73 728 = 160 768
constexpr int NTAP = 9;

__device__ __forceinline__ unsigned pk(float a, float b)
{
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pk_f32x2{a, b}, pk_bf16x2));   // one v_cvt_pk_bf16_f32 (RNE)
}
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

#define S2B_MF(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)
#define S2B_BARRIER() asm volatile("s_barrier" ::: "memory")
#define S2B_DONE_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

struct Item {
    int img, ox0, oy0;
};

// MTC pixel tiles x one output tile per consumer; consumers = MW = 4 / MTC pixel-tile groups x NW = 4 / MW output tiles (from tile p.nb0 on).
// p.nsplit = TG > 0: the grid is TG groups of blocks, group g computes output tiles p.nb0 + NW g ... of ALL pixel tiles (a launch that
// cannot fill the chip otherwise -- a single triplet's coarse levels: <1, 1> with TG = the layer's tiles makes 4 x the blocks, each
// with a quarter of the MFMAs in a row; the raw patch is then loaded and split once per group).  Same operations per output: same bits.
template <int MTC, int NTC>
__global__ __launch_bounds__(512) void conv3x3_s2b(const ConvLaunch p)
{
    constexpr int MW = 4 / MTC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *V = reinterpret_cast<u32x4 *>(smem);                   // [buffer 2][window 2][kh 2][row 17][RS]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int total = tiles_x * tiles_y * p.nimg;
    const int TG = p.nsplit > 0 ? p.nsplit : 1;
    const int G = (int)gridDim.x / TG;                              // blocks that share the pixel tiles of one output-tile group
    const int bx = (int)blockIdx.x % G, tg = (int)blockIdx.x / G;
    const int nchunks = p.seg[0].nchunks + (p.nseg > 1 ? p.seg[1].nchunks : 0);
    const int nitems = (total - bx + G - 1) / G;
    const int nstream = nitems * nchunks;
    const int NT = (p.cout + 31) / 32;                              // 32-output tiles of the layer

    auto decode = [&](const int k) {
        Item it;
        int bid = xcd_remap(bx + k * G, total);
        it.ox0 = (bid % tiles_x) * TW;
        bid /= tiles_x;
        it.oy0 = (bid % tiles_y) * TH;
        it.img = bid / tiles_y;
        return it;
    };

    if (wave >= 4) {
        // ============================================ PRODUCERS ============================================
        const int pw = wave - 4;
        const unsigned raw_lds = static_cast<unsigned>(reinterpret_cast<size_t>(smem)) + 2 * VBUF * 16;
        const u32x4 *RAW = V + 2 * VBUF;                            // [NRING][RAWBUF]: slot = (row 33 + column) 2 + kh
        // split items: raw slot i = 256 q + 64 pw + lane, q = 0..4 -> V slot
        int sdst[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int i = 256 * q + 64 * pw + lane;
            const int row = i / (2 * PC), rem = i - row * 2 * PC, px = rem >> 1, kh = rem & 1;
            sdst[q] = i < NRAW ? kh * VPL + row * RS + (px & 1) * 17 + (px >> 1) : -1;
        }
        // DMA pieces pw, pw + 4, ... (18 in all: five for producers 0, 1, four for 2, 3)
        int doff[5];
        i32x4 rs0, rs1;
        int pk_item = -1;
        auto setup_item = [&](const int k) {
            const Item it = decode(k);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int i = 64 * (pw + 4 * q) + lane;
                const int row = i / (2 * PC), rem = i - row * 2 * PC;
                const int gy = 2 * it.oy0 - 1 + row, gx = 2 * it.ox0 - 1 + (rem >> 1);
                const bool ok = i < NRAW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                doff[q] = ok ? (int)(((unsigned)(gy * p.W + gx) * (unsigned)p.seg[0].pix_stride + 4u * (rem & 1)) * 4u) : -16;   // 0xfffffff0 >= num_records: reads as zero
            }
            const unsigned long long b0 = reinterpret_cast<unsigned long long>(p.seg[0].ptr + (size_t)it.img * p.seg[0].img_stride);
            const unsigned long long b1 = reinterpret_cast<unsigned long long>(p.seg[1].ptr + (size_t)it.img * p.seg[1].img_stride);
            rs0 = i32x4{(int)(unsigned)b0, (int)((unsigned)(b0 >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
            rs1 = i32x4{(int)(unsigned)b1, (int)((unsigned)(b1 >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
            pk_item = k;
        };
        int lk = 0, lc = 0;
        auto request = [&](const int ring) {                        // next chunk of the stream -> raw buffer `ring`
            if (lk != pk_item) setup_item(lk);
            const bool s1 = lc >= p.seg[0].nchunks;
            const long cstr = s1 ? p.seg[1].chunk_stride : p.seg[0].chunk_stride;
            const int so = __builtin_amdgcn_readfirstlane((int)((s1 ? lc - p.seg[0].nchunks : lc) * cstr * 4));
            const i32x4 rsel = s1 ? rs1 : rs0;
            const i32x4 rs = {__builtin_amdgcn_readfirstlane(rsel[0]), __builtin_amdgcn_readfirstlane(rsel[1]), __builtin_amdgcn_readfirstlane(rsel[2]), __builtin_amdgcn_readfirstlane(rsel[3])};
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (pw + 4 * q >= NPIECE) continue;                 // wave-uniform (pw is): no load under a divergent branch
                const int ldst = __builtin_amdgcn_readfirstlane((int)(raw_lds + (unsigned)((ring * RAWBUF + 64 * (pw + 4 * q)) * 16)));
                asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(doff[q]), "s"(rs), "s"(ldst), "s"(so) : "memory");
            }
            if (++lc == nchunks) {
                if (lk + 1 < nitems) { lc = 0; ++lk; } else lc = nchunks - 1;
            }
        };
        auto produce = [&](const int v, const int ring) {
            u32x4 *dst = V + (v & 1) * VBUF;
            const u32x4 *src = RAW + ring * RAWBUF;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const f32x4 d = __builtin_bit_cast(f32x4, src[min(256 * q + 64 * pw + lane, RAWBUF - 1)]);
                u32x4 a, b;
                split(d, a, b);
                if (sdst[q] >= 0) {
                    dst[sdst[q]] = a;
                    dst[2 * VPL + sdst[q]] = b;
                }
            }
        };
        // The DMA of chunk v + 4 is issued while chunk v + 1 is split: before a barrier that publishes chunk v + 2's raw patch a producer
        // waits until only its TWO NEWEST batches of pieces (five each for producers 0, 1, four for 2, 3) are still in flight.
#define S2B_LANDED_BARRIER() do { if (pw < 2) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
        static_assert(NRING == 4, "the waits below count two batches in flight");
        request(0);
        S2B_DONE_BARRIER();                                         // P0: raw patch of chunk 0 is in LDS
        request(1);
        request(2);
        request(3);
        produce(0, 0);
        S2B_LANDED_BARRIER();                                       // A0: V of chunk 0 complete, raw patch of chunk 1 landed
        int r1 = 1, r4 = 0;                                         // ring slots of chunks v + 1 and v + 4
        for (int v = 0; v < nstream; ++v) {
            request(r4);                                            // slot of chunk v: split before B_{v-1}
            produce(v + 1, r1);
            S2B_LANDED_BARRIER();                                   // B_v: V of chunk v + 1 complete, raw patch of chunk v + 2 landed
            S2B_BARRIER();                                          // B'_v
            r1 = r1 == NRING - 1 ? 0 : r1 + 1;
            r4 = r4 == NRING - 1 ? 0 : r4 + 1;
        }
        return;
    }

    // ============================================== CONSUMERS ==============================================
    const int mw = wave % MW, nw = wave / MW;
    const int j32 = lane & 31, kh = lane >> 5;
    const int dr = j32 >> 4, dc = j32 & 15;
    const int x_lane = kh * VPL + (2 * (2 * MTC * mw + dr)) * RS + dc;   // + window 2 VPL + (4 mt + ky) RS + (kx & 1) 17 + (kx >> 1)
    const int w_lane = (kh * 32 + j32) * 16;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(p.wpk_s2b)), 0, 0x7fffffff, 0x00020000);
    const int tap_bytes = 2048 * NT, chunk_bytes = NTAP * tap_bytes;

    static_assert(NTC == 1, "one output tile per consumer (more tiles: more launches, see launch_conv3x3_s2b)");
    f32x16 acc[MTC];
    // weight windows: a ring over the taps, loaded WLA taps ahead.  Two taps (24 MFMAs) cover the L2 round trip when a consumer has four
    // pixel tiles; with ONE (the small launches' <1, 1> form: six MFMAs per tap) the consumers waited for every tap's weights -- 3 700 cycles
    // per chunk for 900 of MFMAs -- so that form looks seven taps ahead (72 registers it has to spare).
    constexpr int WLA = MTC == 1 ? 7 : 2, WR = MTC == 1 ? 9 : 3;
    static_assert(NTAP % WR == 0 && WLA < WR, "ring slots are compile-time constants of the tap index");
    u32x4 wa[WR], wb[WR];
    // Pixel windows in THREE rotating register sets: a set is loaded one tap ahead and was last read a whole tap before that.  A ds_read into a register that a just-issued MFMA still has to read corrupts that MFMA (measured: the hardware
    // does not interlock it and hipcc adds no wait states; wrong outputs in the lanes of the first LDS return group) -- so no register
    // is reloaded "right behind the MFMAs that read it" here, and consecutive MFMAs never share an accumulator.
    u32x4 xa[3][MTC], xb[3][MTC];
    const int tile = nw + p.nb0 + (4 / MW) * tg;                    // this consumer's tile of 32 outputs
    const bool active = tile < NT;
    auto load_w = [&](const int slot, const int chunk_off, const int tap) {
        const int so = chunk_off + tap * tap_bytes + min(tile, NT - 1) * 1024;   // a tile past the last one: its accumulators stay unused
        wa[slot] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, so, 0));
        wb[slot] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_lane, so + 1024 * NT, 0));
    };
    auto load_x = [&](const u32x4 *xbuf, const int mt, const int tap, const int win) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        return xbuf[win * 2 * VPL + (4 * mt + ky) * RS + (kx & 1) * 17 + (kx >> 1)];
    };
    auto zero = [&]() {
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    };

    int k = 0, c = 0;
    Item cur = decode(0);
    int w_cur = 0;
#pragma unroll
    for (int t = 0; t < WLA; ++t) load_w(t, 0, t);
    zero();
    S2B_BARRIER();                                                  // P0
    S2B_BARRIER();                                                  // A0
    {
        const u32x4 *xb0 = V + x_lane;
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt) { xa[0][mt] = load_x(xb0, mt, 0, 0); xb[0][mt] = load_x(xb0, mt, 0, 1); }
    }
    for (int v = 0; v < nstream; ++v) {
        const u32x4 *xcur = V + (v & 1) * VBUF + x_lane;
        const u32x4 *xnext = V + ((v + 1) & 1) * VBUF + x_lane;
        const int w_nxt = c + 1 < nchunks ? w_cur + chunk_bytes : 0;
#pragma unroll
        for (int s = 0; s < NTAP; ++s) {
            constexpr int XS[NTAP + 1] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 0};     // register set of tap s (entry 9 = tap 0 of the next chunk): the set loaded during tap s was last read in tap s - 2
            if (s + WLA < NTAP) load_w((s + WLA) % WR, w_cur, s + WLA);
            else load_w((s + WLA) % WR, w_nxt, s + WLA - NTAP);
            if (s == NTAP - 2) S2B_BARRIER();                       // B_v: V of chunk v + 1 is complete
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt) {                      // the next tap's windows into the set that was read a tap ago
                xa[XS[s + 1]][mt] = s + 1 < NTAP ? load_x(xcur, mt, s + 1, 0) : load_x(xnext, mt, 0, 0);
                xb[XS[s + 1]][mt] = s + 1 < NTAP ? load_x(xcur, mt, s + 1, 1) : load_x(xnext, mt, 0, 1);
            }
            __builtin_amdgcn_sched_barrier(0);                      // the scheduler otherwise sinks every load to its first use
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt) S2B_MF(acc[mt], wa[s % WR], xa[XS[s]][mt]);
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt) S2B_MF(acc[mt], wb[s % WR], xa[XS[s]][mt]);
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt) S2B_MF(acc[mt], wa[s % WR], xb[XS[s]][mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        S2B_BARRIER();                                              // B'_v: done reading V of chunk v
        w_cur = w_nxt;
        if (++c == nchunks) {
            // ---- epilogue: lane (pixel (dr, dc) of its tiles, kh) holds outputs 8 i + 4 kh + r of every 32-output tile: bias,
            // LeakyReLU, 16-byte stores (16 lanes = 16 consecutive pixels of a row, the kh halves complete each pixel's 32 bytes) ----
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)cur.img * p.out_img_stride, 0, 0x7fffffff, 0x00020000);
            int le = lane;
            asm volatile("" : "+v"(le));
            const int e_kh = le >> 5, e_dr = (le >> 4) & 1, e_dc = le & 15;
            const int oy_w = cur.oy0 + 2 * MTC * mw;
            const unsigned lane_off = ((unsigned)((oy_w + e_dr) * p.Wo + cur.ox0 + e_dc) * (unsigned)p.out_pix_stride + 4u * e_kh) * 4u;
            typedef unsigned long long u64;
            const u64 m_px = __builtin_amdgcn_ballot_w64(cur.ox0 + e_dc < p.Wo), m_r0 = __builtin_amdgcn_ballot_w64(e_dr == 0), m_kh0 = __builtin_amdgcn_ballot_w64(e_kh == 0);
            const int row_bytes = p.Wo * p.out_pix_stride * 4;
            const float slope = p.leaky ? 0.2f : 1.f;
            if (active) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = 32 * tile + 8 * i;               // + 4 kh per lane
                    const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias_s2b + co + 4 * e_kh);   // padded to whole tiles
                    const u64 m_co = co + 4 < p.cout ? ~0ull : (co < p.cout ? m_kh0 : 0ull);
                    const int s_chunk = (co >> 3) * (int)p.out_chunk_stride * 4;
#pragma unroll
                    for (int mt = 0; mt < MTC; ++mt) {
                        f32x4 val;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = acc[mt][4 * i + r] + bias[r];
                            val[r] = __builtin_fmaxf(a, slope * a);
                        }
                        const int row = 2 * mt;                     // rows row, row + 1
                        const u64 m_row = oy_w + row + 1 < p.Ho ? ~0ull : (oy_w + row < p.Ho ? m_r0 : 0ull);
                        const u64 m = m_co & m_row & m_px;
                        int v_off;
                        asm("v_cndmask_b32_e64 %0, -16, %1, %2" : "=v"(v_off) : "v"(lane_off), "s"(m));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), o_rsrc, v_off, s_chunk + row * row_bytes, 0);
                    }
                }
            }
            zero();
            c = 0;
            ++k;
            if (k < nitems) cur = decode(k);
        }
    }
}

}  // namespace s2b

int s2b_ntiles(int cout) { return (cout + 31) / 32; }
size_t s2b_wpk_floats(int cin_chunks, int cout) { return (size_t)cin_chunks * s2b::NTAP * 2048 * s2b_ntiles(cout) / 4; }

bool s2b_supported(const ConvLaunch &p)
{
    if (p.stride != 2 || !p.wpk_s2b || !p.bias_s2b) return false;
    if (p.Ho != (p.H - 1) / 2 + 1 || p.Wo != (p.W - 1) / 2 + 1) return false;
    if (p.nseg > 1 && p.seg[1].pix_stride != p.seg[0].pix_stride) return false;
    if (((p.out_pix_stride | (int)p.out_chunk_stride) & 3) != 0 || (p.cout & 3) != 0 || p.cout > 256) return false;   // 16-byte stores; <= 8 output tiles per block
    for (int i = 0; i < p.nseg; ++i) {
        if ((p.seg[i].pix_stride & 3) != 0 || (p.seg[i].chunk_stride & 3) != 0 || (p.seg[i].img_stride & 3) != 0) return false;
        if ((double)p.seg[i].nchunks * (double)p.seg[i].chunk_stride * 4.0 >= 2147483648.0) return false;
    }
    if ((double)p.Ho * p.Wo * p.out_pix_stride * 4.0 + (double)((p.cout + 7) / 8) * (double)p.out_chunk_stride * 4.0 >= 2147483648.0) return false;
    return (double)p.H * p.W * p.seg[0].pix_stride * 4.0 < 2147483648.0;
}

template <int MTC, int NTC>
static hipError_t s2b_launch_t(const ConvLaunch &p, int grid, hipStream_t s)
{
    static bool attr_done_dev[64] = {false};
    bool &attr_done = attr_done_dev[attr_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&s2b::conv3x3_s2b<MTC, NTC>), hipFuncAttributeMaxDynamicSharedMemorySize, s2b::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((s2b::conv3x3_s2b<MTC, NTC>), dim3((unsigned)grid), dim3(512), s2b::LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t launch_conv3x3_s2b(const ConvLaunch &p, hipStream_t s)
{
    using namespace s2b;
    if (!s2b_supported(p)) return hipErrorInvalidValue;
    static int n_cu_dev[64] = {0};
    int &n_cu = n_cu_dev[attr_slot()];
    if (!n_cu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
        n_cu &= ~7;
        if (n_cu < 8) n_cu = 8;
    }
    const int total = ((p.Wo + TW - 1) / TW) * ((p.Ho + TH - 1) / TH) * p.nimg;
    const int cap = p.w4_persist > 1 ? p.w4_persist : n_cu;          // tests: exactly that many blocks
    const int grid = total < cap ? total : cap;
    const int NT = s2b_ntiles(p.cout);
    // consumers: two output-tile groups x two pixel-tile groups for <= 2 output tiles, four output tiles (each consumer all four pixel
    // tiles) above; a layer with more tiles takes one launch per four (the patch is then read once per launch: 128 -> 192 only)
    ConvLaunch q = p;
    q.nsplit = 0;
    // a launch whose blocks cannot fill the chip (single triplets, the coarse levels): one output tile per block, the four consumers
    // take one pixel tile each -- NT x the blocks, a quarter of the MFMAs in a row per wave (p.nsplit < 0 switches it off: option s2_tile_groups = 0)
    if (p.nsplit >= 0 && total * ((NT + 3) / 4) * 2 <= cap && total * NT <= 2 * cap) {
        const int gp = total * NT <= cap ? total : (cap / NT > 0 ? cap / NT : 1);
        q.nb0 = 0;
        q.nsplit = NT;
        return s2b_launch_t<1, 1>(q, gp * NT, s);
    }
    if (NT <= 2) { q.nb0 = 0; return s2b_launch_t<2, 1>(q, grid, s); }
    for (int t0 = 0; t0 < NT; t0 += 4) {
        q.nb0 = t0;
        hipError_t e = s2b_launch_t<4, 1>(q, grid, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static inline unsigned short s2b_bf16_rne(float f)
{
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float s2b_bf16_f32(unsigned short h)
{
    const unsigned u = (unsigned)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// [chunk][tap 9][window 2][output tile NT][kh 2][co 32] x 8 bf16: Wa = (m0..3 h0..3), Wb = (h0..3 l0..3) of the channels 8 chunk + 4 kh + 0..3,
// w = h + m + l exactly; bpk: bias padded to whole tiles
void s2b_pack_weights(const float *w, const float *b, int Co, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk)
{
    const int NT = s2b_ntiles(Co);
    unsigned short *out = reinterpret_cast<unsigned short *>(wpk);
    for (int c = 0; c < cin_chunks; ++c)
        for (int tap = 0; tap < s2b::NTAP; ++tap)
            for (int t = 0; t < NT; ++t)
                for (int h = 0; h < 2; ++h)
                    for (int nn = 0; nn < 32; ++nn) {
                        const size_t base = ((size_t)c * s2b::NTAP + tap) * 2 * NT * 2 * 32;      // 16-byte units, window 0
                        unsigned short *qa = out + (base + (size_t)((0 * NT + t) * 2 + h) * 32 + nn) * 8;
                        unsigned short *qb = out + (base + (size_t)((1 * NT + t) * 2 + h) * 32 + nn) * 8;
                        for (int j = 0; j < 4; ++j) {
                            const int co = t * 32 + nn;
                            const int k = c * kCK + h * 4 + j;
                            const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                            float v = 0.f;
                            if (co < Co && ci >= 0) v = w[((size_t)co * Ci + ci) * 9 + tap];
                            const unsigned short hh = s2b_bf16_rne(v);
                            const float r1 = v - s2b_bf16_f32(hh);
                            const unsigned short mm = s2b_bf16_rne(r1);
                            const float r2 = r1 - s2b_bf16_f32(mm);
                            const unsigned short ll = s2b_bf16_rne(r2);
                            qa[j] = mm; qa[4 + j] = hh;
                            qb[j] = hh; qb[4 + j] = ll;
                        }
                    }
    for (int i = 0; i < NT * 32; ++i) bpk[i] = i < Co ? b[i] : 0.f;
}

}  // namespace b2f
