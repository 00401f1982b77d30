// Fused bilinear warp + 9x9 cost volume, "unit" form (round 3, corr_variant 5) for gfx950 (MI355X).
//
// Same arithmetic, in the same order, as the kernels of b2f_corr.hip (results are bit-identical):
//   ws[f][l] = BilinearSamplerBHWD(cs[f][l], ufs[l+1] * 20(f-2)/2^(l-1))        BilinearSamplerBHWD.cu:41-115, pwc.lua:393-409
//   fwd[c]   = 1/C * sum_k ref[y,x,k] * W3[y-qy, x-qx, k],   c = (qx+4)*9 + (qy+4)  CostVolMulti.lua:62-100
//   bwd[c]   = 1/C * sum_k ref[y,x,k] * W1[y+qy, x+qx, k]
// What is different is WHO holds the 162 sums of a pixel and HOW the neighbour map reaches the CU.
//
// The kernels of b2f_corr.hip keep 81 (or 2 x 81) accumulators per thread -- two waves per SIMD, no registers for loads in
// flight -- and gather the four bilinear taps of every halo pixel straight from memory: measured (profiles/
// r03_corr_unit_kernel.txt) a wave's 64 x 16-byte tap load costs ~45 cycles of the CU's texture-address path, 4 taps x a
// halo of 2.25 - 3 pixels per output pixel make that path, not HBM, the bound.  Here:
//
//   stage  = (tile of 8 x 16 pixels, direction, group of 16 channels);
//   window = the UNWARPED source pixels the stage's 16 x 24 halo taps fall into (bounding box of the clamped tap
//            coordinates, up to 20 x 28 pixels), brought to LDS by LDS-DMA (global_load_lds_dwordx4: no registers, 1 KB
//            contiguous per instruction -> 35 coalesced pieces instead of 96 scattered tap loads) two stages ahead;
//   blend  = warped halo of the NEXT stage from its window (4 ds_read_b128 taps per item, the sampler's arithmetic), all
//            twelve waves, LDS to LDS;
//   unit   = (stage, chunk j of 8 output channels): wave j, lane = vertical pixel pair, 16 (18 for j = 9) accumulators --
//            the two pixels of a pair share the neighbour rows (row f serves qy = -f of the upper and qy = 1 - f of the
//            lower pixel), so a unit reads ~10 neighbour float4 + 2 reference float4 from LDS per 64 - 72 FMAs;
//   block  = 12 waves (768 threads), persistent, one per CU: waves 0..9 compute one unit per stage, waves 10, 11 compute
//            the sampling records + bounding boxes of the tile-directions ahead and DMA the reference tiles.
// One LDS-only barrier per stage:  [DMA window(s+2), reference tile(s+1)] [blend halo(s+1)] [units: FMAs of s | aux: records]
// [stores of a finished tile-direction] barrier.  A tile-direction whose taps spread over more than the window holds (flow
// varying by more than ~3 pixels inside a tile) gathers its taps from memory instead, synchronously (block-uniform fallback).
// A tile's result leaves as one 32-byte chunk per pixel and unit straight from the accumulators (the record's slot order
// [fwd 0..79 | bwd 0..79 | fwd80 bwd80 u v ub vb 0 0] makes every unit one whole chunk, b2f_internal.h).
#include "b2f_internal.h"

#include <climits>
#include <cstdio>

namespace b2f {

namespace v5 {
constexpr int R = 4;
constexpr int TH = 8, TW = 16;                 // output tile
constexpr int HH = TH + 2 * R, HW = TW + 2 * R;   // 16 x 24 halo
constexpr int NHP = HH * HW;                   // 384 halo pixels
constexpr int PLN = NHP + 4;                   // float4 per k4 plane of the warped halo: = 4 (mod 8), so that the lane pair (pixel,
                                               // k4 even | odd) of the blend writes distinct LDS banks
constexpr int NK4 = 4;                         // float4 planes of a stage (16 channels)
constexpr int NCC = NK4 / 2;                   // 8-channel chunks of a stage
constexpr int CGC = 4 * NK4;                   // channels of a stage
constexpr int WR = 20, WC = 28;                // source window: rows x columns
constexpr int WPX = WR * WC;
constexpr int WIN_F4 = NCC * WPX * 2;          // float4 of a window buffer, layout [chunk][row][col][16-byte half]: 2 240
constexpr int WPIECES = WIN_F4 / 64;           // 1-KB LDS-DMA pieces per window: 35
static_assert(WIN_F4 % 64 == 0, "window = whole DMA pieces");
constexpr int NTHR = 768;
constexpr int NUNIT = 10;
constexpr int HALO_F4 = NK4 * PLN;
constexpr int REF_F4 = NK4 * TH * TW;
constexpr int NREC = 3;                        // sampling-record buffers (tile-directions alive at once)
constexpr int OFF_WIN = 0, OFF_HALO = OFF_WIN + 2 * WIN_F4, OFF_REF = OFF_HALO + 2 * HALO_F4, OFF_REC = OFF_REF + 2 * REF_F4,
              OFF_BB = OFF_REC + NREC * NHP;   // float4 offsets inside the dynamic LDS
constexpr int LDS_BYTES = 16 * (OFF_BB + 8);   // 156 288 B; bounding boxes: NREC x 2 aux waves x {min x, min y, max x, max y}
// unit J of direction D: output slots e = 0..7 (and 8 for J = 9) are channels c = 8 J + e (c = 80 for e = 8) of that
// direction; the bwd volume runs the fwd code on the mirrored window (bwd channel c uses the offset of fwd channel 80 - c)
__host__ __device__ constexpr int cprime(int D, int J, int e) { return D ? 80 - (e < 8 ? 8 * J + e : 80) : (e < 8 ? 8 * J + e : 80); }
__host__ __device__ constexpr int qx_of(int cp) { return cp / 9 - 4; }
__host__ __device__ constexpr int qy_of(int cp) { return cp % 9 - 4; }
}  // namespace v5

// Profiling only (results are wrong): -DB2F_C5_ABLATE=bits, 1 no window DMA, 2 no unit FMAs / LDS operand reads, 4 no record stores,
// 8 no blend / halo writes, 16 no reference-tile DMA (variant 6), 32 every tile-direction takes the gather fallback (correct
// results), 64 no sampling records after the prologue (variant 6)
#ifndef B2F_C5_ABLATE
#define B2F_C5_ABLATE 0
#endif
// 1: the source window of a stage is staged in LDS by LDS-DMA where the tile-direction's taps fit it (see the header); 0: the taps
// are always gathered from memory, software-pipelined under the FMAs (measured faster so far: profiles/r03_corr_unit_kernel.txt)
#ifndef B2F_C5_WINDOW
#define B2F_C5_WINDOW 0
#endif
#ifndef B2F_C5_BATCH
#define B2F_C5_BATCH 6
#endif
// Profiling only: -DB2F_C5_TRACE=1 stamps clock64() at the phase boundaries of the first stages of block 40 (lane 0 of waves 0, 4,
// 9 and 10); the launcher prints them after the third large launch
#ifndef B2F_C5_TRACE
#define B2F_C5_TRACE 0
#endif
#if B2F_C5_TRACE
__device__ long long c5_trace_buf[4 * 64 * 8];
#define C5_T(k_) do { if (tr_on && s < 64) c5_trace_buf[(tslot * 64 + s) * 8 + (k_)] = clock64(); } while (0)
#else
#define C5_T(k_) do {} while (0)
#endif

struct C5Tile {
    int b, y0, x0, dir;
};

template <int D, int J, int NB = B2F_C5_BATCH>
__device__ __forceinline__ void corr5_unit(const float4 *__restrict__ hb, const float4 *__restrict__ rb, float (&acc)[18])
{
    using namespace v5;
    constexpr int NE = J == 9 ? 9 : 8;
    // (column qx, row f) pairs this unit reads: every neighbour some slot's upper (f = -qy) or lower (f = 1 - qy) pixel uses
    auto used = [](int col, int f) {
        bool u = false;
        for (int e = 0; e < NE; ++e) {
            const int cp = cprime(D, J, e);
            if (qx_of(cp) == col && (-qy_of(cp) == f || 1 - qy_of(cp) == f)) u = true;
        }
        return u;
    };
#pragma unroll 1
    for (int k4 = 0; k4 < NK4; ++k4) {
        const float4 *hk = hb + k4 * PLN;
        const float4 ru = rb[k4 * (TH * TW)], rl = rb[k4 * (TH * TW) + TW];
        // the step's 10 - 11 neighbour float4 in batches of NB: a batch's LDS reads first, then its FMAs (the other two waves of
        // the SIMD cover the latency; all reads up front cost 48 registers the gather's loads in flight need)
#pragma unroll
        for (int b0 = 0; b0 < 12; b0 += NB) {
            float4 n[NB];
            {
                int i = 0;
#pragma unroll
                for (int col = -4; col <= 4; ++col)
#pragma unroll
                    for (int f = -4; f <= 5; ++f)
                        if (used(col, f)) {
                            if (i >= b0 && i < b0 + NB) n[i - b0] = hk[f * HW - col];     // fwd volume: neighbour at (y - qy, x - qx), CostVolMulti.lua:76-87
                            ++i;
                        }
            }
            __builtin_amdgcn_sched_barrier(0);
            int i = 0;
#pragma unroll
            for (int col = -4; col <= 4; ++col) {
#pragma unroll
                for (int f = -4; f <= 5; ++f) {
                    if (!used(col, f)) continue;
                    const int ii = i++;
                    if (ii < b0 || ii >= b0 + NB) continue;
                    const float4 nv = n[ii - b0];
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const int cp = cprime(D, J, e);
                        if (qx_of(cp) != col) continue;
                        if (-qy_of(cp) == f) {
                            float a = acc[e];
                            a = fmaf(ru.x, nv.x, a); a = fmaf(ru.y, nv.y, a); a = fmaf(ru.z, nv.z, a); a = fmaf(ru.w, nv.w, a);
                            acc[e] = a;
                        }
                        if (1 - qy_of(cp) == f) {
                            float a = acc[9 + e];
                            a = fmaf(rl.x, nv.x, a); a = fmaf(rl.y, nv.y, a); a = fmaf(rl.z, nv.z, a); a = fmaf(rl.w, nv.w, a);
                            acc[9 + e] = a;
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// scale by 1/C (output:div(N), CostVolMulti.lua:100) and store unit j of direction d of the pixel pair.  The accumulator
// layout is the same for every unit, so direction and unit enter as (wave-uniform) run-time values: one copy of the code and
// one address computation (as template parameters the twenty chunk addresses were hoisted in front of the FMA loop)
template <bool POW2>
__device__ __forceinline__ void corr5_store(const CorrLaunch &p, const C5Tile &t, int d, int j, int pr, int lx, float (&acc)[18])
{
    using namespace v5;
    const float cf = (float)p.C, inv = 1.f / cf;
    // the addresses are formed here, after the FMA loop (hoisted in front of it they are spilled for the whole stage)
    asm volatile("" : "+v"(lx), "+v"(pr));
    const int px = t.x0 + lx;
    // wave-uniform bases + 32-bit offsets inside the image (warp_costvol_unit_supported checks that they fit)
    float *ob = p.out + (size_t)t.b * p.out_img_stride;
    float *obc = ob + (size_t)(d * 10 + j) * p.out_chunk_stride, *obl = ob + (size_t)20 * p.out_chunk_stride;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int py = t.y0 + 2 * pr + u;
        float a[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) a[e] = POW2 ? acc[9 * u + e] * inv : acc[9 * u + e] / cf;
        if (py < p.h && px < p.w && !((B2F_C5_ABLATE & 4) && a[0] != 12345.678f)) {
            const int pix = py * p.w + px;
            const unsigned off = (unsigned)(pix * p.out_pix_stride);
            float *oc = obc + off;
            *reinterpret_cast<float4 *>(oc) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4 *>(oc + 4) = make_float4(a[4], a[5], a[6], a[7]);
            if (j == 9) {   // last chunk: [fwd80, bwd80, u, v, ub, vb, 0, 0]
                float *ol = obl + off;
                if (d == 0) {
                    ol[0] = a[8];
                } else {
                    const size_t fp = ((size_t)t.b * p.h * p.w + pix) * 2;
                    float2 f = make_float2(0.f, 0.f), fb = make_float2(0.f, 0.f);
                    if (p.flow) f = *reinterpret_cast<const float2 *>(p.flow + fp);
                    if (p.flow_b) fb = *reinterpret_cast<const float2 *>(p.flow_b + fp);
                    ol[1] = a[8];
                    ol[2] = f.x; ol[3] = f.y;
                    *reinterpret_cast<float4 *>(ol + 4) = make_float4(fb.x, fb.y, 0.f, 0.f);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 18; ++e) acc[e] = 0.f;
}

// LDS-DMA of one 1-KB piece: lane i's 16 bytes at g land at LDS byte address lds + 16 i.  Not counted by the compiler's
// s_waitcnt bookkeeping: every stage ends with an explicit vmcnt(0) before its barrier.
__device__ __forceinline__ void c5_dma(const float *g, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}

// bounding box of a tile-direction's taps (written by the two aux waves) -> window origin, does it fit the window
// (the origin is pulled back so that the whole window lies inside the image where the image is at least that large: the DMA of
// such a window needs no clamping)
template <int NW = 2, bool WIN = (B2F_C5_WINDOW != 0)>
__device__ __forceinline__ void c5_bbox(const CorrLaunch &p, const int *bb, int &wx0, int &wy0, bool &fits)
{
    using namespace v5;
    if (!WIN) { wx0 = wy0 = 0; fits = false; return; }
    int x0 = bb[0], y0 = bb[1], x1 = bb[2], y1 = bb[3];
#pragma unroll
    for (int w = 1; w < NW; ++w) { x0 = min(x0, bb[4 * w]); y0 = min(y0, bb[4 * w + 1]); x1 = max(x1, bb[4 * w + 2]); y1 = max(y1, bb[4 * w + 3]); }
    wx0 = min(__builtin_amdgcn_readfirstlane(x0), max(p.w - WC, 0));
    wy0 = min(__builtin_amdgcn_readfirstlane(y0), max(p.h - WR, 0));
    fits = __builtin_amdgcn_readfirstlane((x1 - x0 + 1 <= WC && y1 - y0 + 1 <= WR) ? 1 : 0) != 0 && !(B2F_C5_ABLATE & 32);
}

// min / max over the 64 lanes of a wave in DPP steps (row_shr 1, 2, 4, 8, row_bcast 15, 31): the result is in lane 63
template <bool MAX>
__device__ __forceinline__ int c5_wave_reduce(int v)
{
#define C5_DPP_STEP(ctrl_, rmask_)                                                              \
    do {                                                                                        \
        const int o__ = __builtin_amdgcn_update_dpp(v, v, (ctrl_), (rmask_), 0xf, false);       \
        v = MAX ? max(v, o__) : min(v, o__);                                                    \
    } while (0)
    C5_DPP_STEP(0x111, 0xf); C5_DPP_STEP(0x112, 0xf); C5_DPP_STEP(0x114, 0xf); C5_DPP_STEP(0x118, 0xf);
    C5_DPP_STEP(0x142, 0xa); C5_DPP_STEP(0x143, 0xc);
#undef C5_DPP_STEP
    return v;
}

// flows of the three halo pixels an aux thread owns (hp = ar + 128 j)
template <int NJ = 3, int STR = 128>
__device__ __forceinline__ void c5_rec_issue(const CorrLaunch &p, const C5Tile &t, int ar, float2 &f0, float2 &f1, float2 &f2)
{
    using namespace v5;
    float2 fl[3];
    fl[2] = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int hp = ar + STR * j;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int y = t.y0 - R + hy, x = t.x0 - R + hx;
        const bool in = y >= 0 && y < p.h && x >= 0 && x < p.w && hp < NHP;
        fl[j] = make_float2(0.f, 0.f);
        if (p.flow) fl[j] = *reinterpret_cast<const float2 *>(p.flow + ((size_t)t.b * p.h * p.w + (in ? (size_t)y * p.w + x : 0)) * 2);
    }
    f0 = fl[0]; f1 = fl[1]; f2 = fl[2];
}
// sampling records: clamped top-left tap (x, y: 12 bits each), "right / bottom neighbour exists" flags, valid bit, and the
// fractional weights wx, wy (getTopLeft, BilinearSamplerBHWD.cu:6-20); + this wave's bounding box of the taps
template <int NJ = 3, int STR = 128, bool WIN = (B2F_C5_WINDOW != 0)>
__device__ __forceinline__ void c5_rec_finish(const CorrLaunch &p, const C5Tile &t, int ar, float2 f0, float2 f1, float2 f2, float4 *rec, int *bbw)
{
    using namespace v5;
    const float2 fl[3] = {f0, f1, f2};
    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = -1, by1 = -1;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int hp = ar + STR * j;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int y = t.y0 - R + hy, x = t.x0 - R + hx;
        int packed = 0;
        float wx = 0.f, wy = 0.f;
        if (NJ * STR > NHP && hp >= NHP) continue;
        if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
            const float k = t.dir == 0 ? p.k : -p.k;   // nn.MulConstant(20*(f-ref)/2^(l-2)), pwc.lua:404
            const float u = fl[j].x * k, v = fl[j].y * k;
            int xl, yt;
            bhwd_top_left(u + (float)x, p.w, xl, wx);
            bhwd_top_left(v + (float)y, p.h, yt, wy);
            const int fx = (xl + 1 <= p.w - 1) ? 1 : 0, fy = (yt + 1 <= p.h - 1) ? 1 : 0;
            packed = xl | yt << 12 | fx << 24 | fy << 25 | 1 << 26;
            if (WIN) {
                bx0 = min(bx0, xl); by0 = min(by0, yt);
                bx1 = max(bx1, xl + fx); by1 = max(by1, yt + fy);
            }
        }
        rec[hp] = make_float4(__int_as_float(packed), wx, wy, 0.f);
    }
    if (WIN) {
        bx0 = c5_wave_reduce<false>(bx0); by0 = c5_wave_reduce<false>(by0);
        bx1 = c5_wave_reduce<true>(bx1); by1 = c5_wave_reduce<true>(by1);
        if ((ar & 63) == 63) { bbw[0] = bx0; bbw[1] = by0; bbw[2] = bx1; bbw[3] = by1; }
    }
}

// taps of a stage's warped halo: thread = (halo pixel hp, 16-byte half hh), its NCC chunks, four taps each -- from the LDS
// window, or from memory (buffer loads: one scalar 128-bit resource based at the image's neighbour map, the chunk as scalar
// offset, four 32-bit lane offsets; 4 NCC loads in flight)
struct C5Samp {
    float4 w4;          // blend weights (all 0 for a halo pixel outside the image: CostVolMulti's out-of-range -> 0)
};
__device__ __forceinline__ void c5_taps(const CorrLaunch &p, const C5Tile &t, int cg, const float4 *rec, bool fits, int wx0, int wy0,
                                        const float4 *win, int hp, int hh, float4 (&tp)[4 * v5::NCC], C5Samp &sm)
{
    using namespace v5;
    const float4 r = rec[hp];
    const int packed = __float_as_int(r.x);
    const float wx = r.y, wy = r.z;
    const int xl = packed & 0xfff, yt = (packed >> 12) & 0xfff, fx = (packed >> 24) & 1, fy = (packed >> 25) & 1;
    sm.w4 = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
    if (!((packed >> 26) & 1)) sm.w4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fits) {
        const int wi = ((packed >> 26) & 1) ? ((yt - wy0) * WC + (xl - wx0)) * 2 + hh : hh;
        const int dx = 2 * fx, dy = 2 * WC * fy;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            const float4 *wc = win + cc * (WPX * 2) + wi;
            tp[4 * cc + 0] = wc[0]; tp[4 * cc + 1] = wc[dx]; tp[4 * cc + 2] = wc[dy]; tp[4 * cc + 3] = wc[dy + dx];
        }
    } else {
        // a neighbour outside the image has weight exactly 0 (coordinates are clamped first): its address is folded onto the
        // clamped pixel instead of branching around the load
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>((t.dir == 0 ? p.nbr_fut : p.nbr_past) + (size_t)t.b * p.img_stride), 0, 0x7fffffff, 0x00020000);
        const int o_tl = ((yt * p.w + xl) * p.pix_stride + 4 * hh) * 4;
        const int dx = fx ? p.pix_stride * 4 : 0, dy = fy ? p.w * p.pix_stride * 4 : 0;
#pragma unroll
        for (int cc = 0; cc < NCC; ++cc) {
            const int so = (int)((long)(cg * NCC + cc) * p.chunk_stride * 4);
            tp[4 * cc + 0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_tl, so, 0));
            tp[4 * cc + 1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_tl + dx, so, 0));
            tp[4 * cc + 2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_tl + dy, so, 0));
            tp[4 * cc + 3] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_tl + dy + dx, so, 0));
        }
    }
}
// the blend is the sampler's arithmetic ((wtl tl + wtr tr) + wbl bl) + wbr br as one mul + three fma, weights as the products
// the other kernels form -> warped halo planes
__device__ __forceinline__ void c5_blend(float4 *hdst, int hp, int hh, const float4 (&tp)[4 * v5::NCC], const C5Samp &sm)
{
    using namespace v5;
    if (B2F_C5_ABLATE & 8) return;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 WX = {sm.w4.x, sm.w4.x}, WY = {sm.w4.y, sm.w4.y}, WZ = {sm.w4.z, sm.w4.z}, WW = {sm.w4.w, sm.w4.w};
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
        const float4 tl = tp[4 * cc], tr = tp[4 * cc + 1], bl = tp[4 * cc + 2], br = tp[4 * cc + 3];
        // two components per instruction (v_pk_mul_f32 / v_pk_fma_f32 with the weight broadcast): the same roundings
        f32x2 lo = WX * (f32x2){tl.x, tl.y}, hi = WX * (f32x2){tl.z, tl.w};
        lo = __builtin_elementwise_fma(WY, (f32x2){tr.x, tr.y}, lo); hi = __builtin_elementwise_fma(WY, (f32x2){tr.z, tr.w}, hi);
        lo = __builtin_elementwise_fma(WZ, (f32x2){bl.x, bl.y}, lo); hi = __builtin_elementwise_fma(WZ, (f32x2){bl.z, bl.w}, hi);
        lo = __builtin_elementwise_fma(WW, (f32x2){br.x, br.y}, lo); hi = __builtin_elementwise_fma(WW, (f32x2){br.z, br.w}, hi);
        const float4 v = make_float4(lo.x, lo.y, hi.x, hi.y);
        hdst[(2 * cc + hh) * PLN + hp] = v;
    }
}

// LDS-DMA of a stage's source window: 35 pieces over the 12 waves; item = 64 piece + lane -> (chunk, row, col, half), the source
// pixel clamped into the image (columns / rows past the edge re-read the edge: never used)
template <int NW = 12>
__device__ __forceinline__ void c5_win_dma(const CorrLaunch &p, const C5Tile &t, int cg, int wx0, int wy0, unsigned lds_win, int wave, int lane)
{
    using namespace v5;
    if (B2F_C5_ABLATE & 1) return;
    const float *base = (t.dir == 0 ? p.nbr_fut : p.nbr_past) + (size_t)t.b * p.img_stride + (size_t)(cg * NCC) * p.chunk_stride;
#pragma unroll
    for (int j = 0; j < (WPIECES + NW - 1) / NW; ++j) {
        const int pc = wave + NW * j;
        if (pc < WPIECES) {
            const int item = 64 * pc + lane;
            const int half = item & 1, pxi = item >> 1;
            const int cc = pxi / WPX, rr = pxi - cc * WPX;
            const int row = rr / WC, col = rr - row * WC;
            const int gy = min(wy0 + row, p.h - 1), gx = min(wx0 + col, p.w - 1);
            c5_dma(base + (size_t)cc * p.chunk_stride + (size_t)(gy * p.w + gx) * p.pix_stride + 4 * half, lds_win + 1024u * pc);
        }
    }
}
// LDS-DMA of a stage's reference tile: 8 pieces over NW waves; piece = (k4 plane, half of the tile's 128 pixels)
template <int NW = 2>
__device__ __forceinline__ void c5_ref_dma(const CorrLaunch &p, const C5Tile &t, int cg, unsigned lds_ref, int auxw, int lane)
{
    using namespace v5;
    const float *base = p.ref + (size_t)t.b * p.img_stride + (size_t)(cg * NCC) * p.chunk_stride;
#pragma unroll
    for (int j = 0; j < (2 * NK4 + NW - 1) / NW; ++j) {
        const int pc = auxw + NW * j, k4 = pc >> 1, px = 64 * (pc & 1) + lane;
        if (pc < 2 * NK4) {
            const int y = min(t.y0 + (px >> 4), p.h - 1), x = min(t.x0 + (px & 15), p.w - 1);   // ragged tiles: clamped, never stored
            c5_dma(base + (size_t)(k4 >> 1) * p.chunk_stride + (size_t)(y * p.w + x) * p.pix_stride + 4 * (k4 & 1), lds_ref + 1024u * pc);
        }
    }
}

// the same two with everything per-lane precomputed (variant 6): a window that lies inside the image / a whole tile is a scalar
// base (image, chunk pair, origin pixel) + a per-lane byte offset that only depends on the launch
__device__ __forceinline__ void c5_dma_s(const float *sbase, int voff, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
template <int NW>
__device__ __forceinline__ void c6_win_offsets(const CorrLaunch &p, int gw, int lane, int (&woff)[(v5::WPIECES + NW - 1) / NW])
{
    using namespace v5;
#pragma unroll
    for (int j = 0; j < (WPIECES + NW - 1) / NW; ++j) {
        const int item = 64 * (gw + NW * j) + lane;
        const int half = item & 1, pxi = item >> 1;
        const int cc = pxi / WPX, rr = pxi - cc * WPX;
        const int row = rr / WC, col = rr - row * WC;
        woff[j] = (cc * p.chunk_stride + (row * p.w + col) * p.pix_stride + 4 * half) * 4;
    }
}
template <int NW>
__device__ __forceinline__ void c6_win_dma(const CorrLaunch &p, const C5Tile &t, int cg, int wx0, int wy0, unsigned lds_win, int gw,
                                           const int (&woff)[(v5::WPIECES + NW - 1) / NW])
{
    using namespace v5;
    if (B2F_C5_ABLATE & 1) return;
    const float *sbase = (t.dir == 0 ? p.nbr_fut : p.nbr_past) + (size_t)t.b * p.img_stride + (size_t)(cg * NCC) * p.chunk_stride +
                         (size_t)(wy0 * p.w + wx0) * p.pix_stride;
#pragma unroll
    for (int j = 0; j < (WPIECES + NW - 1) / NW; ++j) {
        const int pc = gw + NW * j;
        if (pc < WPIECES) c5_dma_s(sbase, woff[j], lds_win + 1024u * pc);
    }
}
template <int NW>
__device__ __forceinline__ void c6_ref_offsets(const CorrLaunch &p, int gw, int lane, int (&roff)[(2 * v5::NK4 + NW - 1) / NW])
{
    using namespace v5;
#pragma unroll
    for (int j = 0; j < (2 * NK4 + NW - 1) / NW; ++j) {
        const int pc = gw + NW * j, k4 = pc >> 1, px = 64 * (pc & 1) + lane;
        roff[j] = ((k4 >> 1) * p.chunk_stride + ((px >> 4) * p.w + (px & 15)) * p.pix_stride + 4 * (k4 & 1)) * 4;
    }
}
template <int NW>
__device__ __forceinline__ void c6_ref_dma(const CorrLaunch &p, const C5Tile &t, int cg, unsigned lds_ref, int gw, int lane,
                                           const int (&roff)[(2 * v5::NK4 + NW - 1) / NW])
{
    using namespace v5;
    if (B2F_C5_ABLATE & 16) return;
    if (t.y0 + TH <= p.h && t.x0 + TW <= p.w) {
        const float *sbase = p.ref + (size_t)t.b * p.img_stride + (size_t)(cg * NCC) * p.chunk_stride + (size_t)(t.y0 * p.w + t.x0) * p.pix_stride;
#pragma unroll
        for (int j = 0; j < (2 * NK4 + NW - 1) / NW; ++j) {
            const int pc = gw + NW * j;
            if (pc < 2 * NK4) c5_dma_s(sbase, roff[j], lds_ref + 1024u * pc);
        }
    } else {
        c5_ref_dma<NW>(p, t, cg, lds_ref, gw, lane);
    }
}

template <int D>
__device__ __forceinline__ void c5_units(int wave, const float4 *hb, const float4 *rb, float (&acc)[18])
{
    switch (wave) {
    case 0: corr5_unit<D, 0>(hb, rb, acc); break;
    case 1: corr5_unit<D, 1>(hb, rb, acc); break;
    case 2: corr5_unit<D, 2>(hb, rb, acc); break;
    case 3: corr5_unit<D, 3>(hb, rb, acc); break;
    case 4: corr5_unit<D, 4>(hb, rb, acc); break;
    case 5: corr5_unit<D, 5>(hb, rb, acc); break;
    case 6: corr5_unit<D, 6>(hb, rb, acc); break;
    case 7: corr5_unit<D, 7>(hb, rb, acc); break;
    case 8: corr5_unit<D, 8>(hb, rb, acc); break;
    default: corr5_unit<D, 9>(hb, rb, acc); break;
    }
}

// -DB2F_C5_BLOCKS=2: two blocks per CU (gather mode only: 59 KB of LDS each, 80 registers per thread)
#ifndef B2F_C5_BLOCKS
#define B2F_C5_BLOCKS 1
#endif
#if B2F_C5_BLOCKS == 2
#define C5_OCC __attribute__((amdgpu_waves_per_eu(6, 6)))
#else
#define C5_OCC
#endif
template <bool POW2>
__global__ __launch_bounds__(768) C5_OCC void warp_costvol_unit_kernel(const CorrLaunch p, const int ntd, const int tiles_x, const int tiles_y)
{
    using namespace v5;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *L = reinterpret_cast<float4 *>(smem);
    // (without the window -- B2F_C5_WINDOW 0 -- its buffers are not allocated: 59 KB, two blocks per CU fit)
    constexpr int WOFF = B2F_C5_WINDOW ? 0 : 2 * WIN_F4;
    constexpr int OFF_HALO = v5::OFF_HALO - WOFF, OFF_REF = v5::OFF_REF - WOFF, OFF_REC = v5::OFF_REC - WOFF, OFF_BB = v5::OFF_BB - WOFF;
    float4 *win = L + OFF_WIN;                                   // [2][WIN_F4]          source windows
    float4 *halo = L + OFF_HALO;                                 // [2][NK4][PLN]        warped neighbour halo of a stage
    float4 *refb = L + OFF_REF;                                  // [2][NK4][TH * TW]    reference tile of a stage
    float4 *recs = L + OFF_REC;                                  // [NREC][NHP]          sampling records of a tile-direction
    int *bbox = reinterpret_cast<int *>(L + OFF_BB);             // [NREC][2][4]
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int ncg = p.C / CGC;
    const int nmine = (ntd - (int)blockIdx.x + G - 1) / G;      // tile-directions of this block: v = blockIdx.x + k G
    const int S = nmine * ncg;
    const bool aux = wave >= NUNIT;                               // waves 10, 11
    const int ar = tid - NUNIT * 64;                              // 0..127 for the aux waves
    const int ghp = tid >> 1, ghh = tid & 1;                      // blend item: halo pixel, 16-byte half of its 32-byte chunk
    const int pr = lane >> 4, lx = lane & 15;                     // unit lane: pixel-pair row, column
#if B2F_C5_TRACE
    const bool tr_on = blockIdx.x == 40 && lane == 0 && (wave == 0 || wave == 4 || wave == 9 || wave == 10) && nmine >= 50;
    const int tslot = wave == 0 ? 0 : wave == 4 ? 1 : wave == 9 ? 2 : 3;
#endif

    // ---- tile-directions k .. k + 4 of this block.  The logical index is XCD-banded like every persistent kernel here (an XCD's
    // 32 CUs walk one contiguous band of tiles, fwd / bwd of a tile on neighbouring slots: halo overlap and the shared reference
    // tile are found in that XCD's L2); consecutive tile-directions of a block are G / 8 apart, stepped without divisions.
    const int step = G >> 3;
    int q_lin, q_tx, q_ty, q_b, q_k = 0;                          // newest decoded tile-direction (index q_k)
    C5Tile t0, t1, t2, t3, t4;
    {
        q_lin = xcd_remap((int)blockIdx.x, ntd);
        const int tile = q_lin >> 1;
        q_tx = tile % tiles_x;
        q_ty = (tile / tiles_x) % tiles_y;
        q_b = tile / (tiles_x * tiles_y);
    }
#define C5_NEWEST(t_) do { t_.b = q_b; t_.y0 = q_ty * TH; t_.x0 = q_tx * TW; t_.dir = q_lin & 1; } while (0)
#define C5_ADVANCE()                                                              \
    do {                                                                          \
        if (q_k + 1 < nmine) {                                                    \
            ++q_k;                                                                \
            const int nl__ = q_lin + step;                                        \
            q_tx += (nl__ >> 1) - (q_lin >> 1);                                   \
            q_lin = nl__;                                                         \
            while (q_tx >= tiles_x) { q_tx -= tiles_x; ++q_ty; }                  \
            while (q_ty >= tiles_y) { q_ty -= tiles_y; ++q_b; }                   \
        }                                                                         \
    } while (0)
    C5_NEWEST(t0); C5_ADVANCE(); C5_NEWEST(t1); C5_ADVANCE(); C5_NEWEST(t2); C5_ADVANCE(); C5_NEWEST(t3); C5_ADVANCE(); C5_NEWEST(t4);
    // tile-direction k + d (selected with scalar compares: an indexed array would live in scratch memory)
#define C5_PICK(t_, d_)                                                                                   \
    do {                                                                                                  \
        const int d__ = (d_);                                                                             \
        t_.b = d__ == 0 ? t0.b : d__ == 1 ? t1.b : d__ == 2 ? t2.b : d__ == 3 ? t3.b : t4.b;              \
        t_.y0 = d__ == 0 ? t0.y0 : d__ == 1 ? t1.y0 : d__ == 2 ? t2.y0 : d__ == 3 ? t3.y0 : t4.y0;        \
        t_.x0 = d__ == 0 ? t0.x0 : d__ == 1 ? t1.x0 : d__ == 2 ? t2.x0 : d__ == 3 ? t3.x0 : t4.x0;        \
        t_.dir = d__ == 0 ? t0.dir : d__ == 1 ? t1.dir : d__ == 2 ? t2.dir : d__ == 3 ? t3.dir : t4.dir;  \
    } while (0)
    // (tile-direction offset, channel group) of the stage d stages after (tile-direction k, group cg)
#define C5_AHEAD(d_, cg_, dk_, cgo_)                               \
    do {                                                           \
        int c__ = (cg_) + (d_), dk__ = 0;                          \
        while (c__ >= ncg) { c__ -= ncg; ++dk__; }                 \
        dk_ = dk__; cgo_ = c__;                                    \
    } while (0)
#define C5_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define C5_VM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

    // ---- prologue ----
    // records of the tile-directions whose turn (three stages before their first stage) lies before stage 0; flows of the one
    // whose turn is stage 0
    float2 fa = make_float2(0.f, 0.f), fb = fa, fc = fa;          // flows in flight for the next records (aux waves)
    if (aux) {
        for (int q = 0; q < nmine && q * ncg <= 2; ++q) {
            C5Tile tt;
            C5_PICK(tt, q);
            c5_rec_issue(p, tt, ar, fa, fb, fc);
            c5_rec_finish(p, tt, ar, fa, fb, fc, recs + (q % NREC) * NHP, bbox + (q % NREC) * 8 + (wave - NUNIT) * 4);
        }
        if (3 % ncg == 0 && 3 / ncg < nmine) {
            C5Tile tt;
            C5_PICK(tt, 3 / ncg);
            c5_rec_issue(p, tt, ar, fa, fb, fc);
        }
    }
    __syncthreads();
    {   // windows of stages 0 and 1, reference tile of stage 0
        int wx0, wy0, dk, cgo;
        bool fits;
        c5_bbox(p, bbox, wx0, wy0, fits);
        if (fits) c5_win_dma(p, t0, 0, wx0, wy0, lds0 + 16u * OFF_WIN, wave, lane);
        if (1 < S) {
            C5_AHEAD(1, 0, dk, cgo);
            C5Tile tt;
            C5_PICK(tt, dk);
            c5_bbox(p, bbox + (dk % NREC) * 8, wx0, wy0, fits);
            if (fits) c5_win_dma(p, tt, cgo, wx0, wy0, lds0 + 16u * (OFF_WIN + WIN_F4), wave, lane);
        }
        if (aux) c5_ref_dma(p, t0, 0, lds0 + 16u * OFF_REF, wave - NUNIT, lane);
        C5_VM_DRAIN();
    }
    __syncthreads();
    {   // warped halo of stage 0
        int wx0, wy0;
        bool fits;
        c5_bbox(p, bbox, wx0, wy0, fits);
        float4 tp[4 * NCC];
        C5Samp sm;
        c5_taps(p, t0, 0, recs, fits, wx0, wy0, win, ghp, ghh, tp, sm);
        c5_blend(halo, ghp, ghh, tp, sm);
    }
    __syncthreads();

    // the tile-direction queue is advanced by both copies of the loop in the same way
    if (aux) {
#define C5_AUX 1
#include "b2f_corr5_loop.inc"
#undef C5_AUX
    } else {
#define C5_AUX 0
#include "b2f_corr5_loop.inc"
#undef C5_AUX
    }
}

// ---- the role-specialised form (corr_variant 6) ------------------------------------------------------------------------------
// Same stages, same LDS buffers, same arithmetic; what changes is who does what.  In the kernel above every wave runs every
// phase of a stage at the same time (gather issue | FMAs | blend | stores), so the phases add up.  Here a block is 8 waves:
//   F waves 0..3: cost-volume channels 20 w .. 20 w + 19 (wave 3: .. 80) of the stage's direction for the lane's vertical pixel
//                 pair (2 x 20 sums; 160 products per ~25 LDS float4), LDS -> FMA only, plus the stores of a finished
//                 tile-direction.  They never wait for memory.
//   G waves 4..7: everything that feeds them, one stage ahead: LDS-DMA of the source window of stage s + 2 and of the reference
//                 tile of s + 1, the blend of stage s + 1's warped halo from its window (LDS -> LDS; gathered from memory when
//                 the tile-direction's taps do not fit the window), the sampling records + bounding boxes of the
//                 tile-directions ahead.
// The hardware places wave w on SIMD (w mod 4) of the block's rotation (tools/simd_probe.hip): every SIMD holds one F and one G
// wave -- the same load on all four, and a SIMD interleaves the G wave's latency chains with its F wave's FMAs.
#if B2F_EXPERIMENTS   // variant 6 (role-specialised form): measured slower at every level, profiles/r03_corr_unit_kernel.txt
namespace v6 {
constexpr int NF = 4, NG = 4;
constexpr int NTHR = 64 * (NF + NG);            // 512
constexpr int GT = 64 * NG;                     // gather threads
constexpr int NIT = 2 * v5::NHP / GT;           // blend items (halo pixel, 16-byte half) per gather thread and stage: 3
constexpr int NRJ = (v5::NHP + GT - 1) / GT;    // sampling records per gather thread and tile-direction: 2 (the second for half the threads)
static_assert(2 * v5::NHP % GT == 0 && GT % 2 == 0, "items divide over the gather threads");
constexpr int OFF_BB = v5::OFF_REC + v5::NREC * v5::NHP;
constexpr int LDS_BYTES = 16 * (OFF_BB + v5::NREC * NG);   // bounding boxes: NREC x NG waves x {min x, min y, max x, max y}
constexpr int NCW = 20;                         // cost-volume channels of a direction per F wave (wave 3: 21, with channel 80)
__host__ __device__ constexpr int cp6(int D, int c) { return D ? 80 - c : c; }   // bwd channel c = fwd arithmetic of channel 80 - c
}  // namespace v6

#ifndef B2F_C6_BATCH
#define B2F_C6_BATCH 13
#endif
// 0 (shipped): one sum per output in channel order, bit-identical with the other variants.
// 1: two partial sums per output (even / odd channel of a float4 half, one v_pk_fma_f32 per two products), added at the end -- not
// the bit pattern of the other variants (max difference 3.6e-7 on sums of 0.8).  Measured 0.87 against 0.99 ms at level 3: a wave
// issues a VALU instruction every ~5 cycles whatever its width (tools/fma_rate.hip), so with one FMA wave per SIMD the packed form
// only shortens an issue-latency-bound stream.
#ifndef B2F_C6_PACKED
#define B2F_C6_PACKED 0
#endif
typedef float c6_f32x2 __attribute__((ext_vector_type(2)));
template <bool PK> struct C6Acc;
template <> struct C6Acc<false> { float a[2][21]; };
template <> struct C6Acc<true> { c6_f32x2 a[2][21]; };

// channels C0 .. C0 + NC - 1 of direction D for the lane's vertical pixel pair, one stage (NK4 float4 planes): the pair shares
// neighbour rows (row f serves qy = -f of the upper and qy = 1 - f of the lower pixel) and consecutive channels share columns:
// ~23 neighbour + 2 reference float4 from LDS per 160 products
template <int D, int C0, int NC, int NB, bool PK>
__device__ __forceinline__ void corr6_range(const float4 *__restrict__ hb, const float4 *__restrict__ rb, C6Acc<PK> &A)
{
    using namespace v5;
    auto used = [](int col, int f) {
        bool u = false;
        for (int e = 0; e < NC; ++e) {
            const int cp = v6::cp6(D, C0 + e);
            if (qx_of(cp) == col && (-qy_of(cp) == f || 1 - qy_of(cp) == f)) u = true;
        }
        return u;
    };
    constexpr int NREAD = [&] { int n = 0; for (int col = -4; col <= 4; ++col) for (int f = -4; f <= 5; ++f) if (used(col, f)) ++n; return n; }();
#pragma unroll 1
    for (int k4 = 0; k4 < NK4; ++k4) {
        const float4 *hk = hb + k4 * PLN;
        const float4 ru = rb[k4 * (TH * TW)], rl = rb[k4 * (TH * TW) + TW];
#pragma unroll
        for (int b0 = 0; b0 < NREAD; b0 += NB) {
            float4 n[NB];
            {
                int i = 0;
#pragma unroll
                for (int col = -4; col <= 4; ++col)
#pragma unroll
                    for (int f = -4; f <= 5; ++f)
                        if (used(col, f)) {
                            if (i >= b0 && i < b0 + NB) n[i - b0] = hk[f * HW - col];     // neighbour at (y - qy, x - qx), CostVolMulti.lua:76-87
                            ++i;
                        }
            }
            __builtin_amdgcn_sched_barrier(0);
            int i = 0;
#pragma unroll
            for (int col = -4; col <= 4; ++col) {
#pragma unroll
                for (int f = -4; f <= 5; ++f) {
                    if (!used(col, f)) continue;
                    const int ii = i++;
                    if (ii < b0 || ii >= b0 + NB) continue;
                    const float4 nv = n[ii - b0];
#pragma unroll
                    for (int e = 0; e < NC; ++e) {
                        const int cp = v6::cp6(D, C0 + e);
                        if (qx_of(cp) != col) continue;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            if ((u == 0 ? -qy_of(cp) : 1 - qy_of(cp)) != f) continue;
                            const float4 r = u == 0 ? ru : rl;
                            if constexpr (PK) {
                                c6_f32x2 a = A.a[u][e];
                                a = __builtin_elementwise_fma((c6_f32x2){r.x, r.y}, (c6_f32x2){nv.x, nv.y}, a);
                                a = __builtin_elementwise_fma((c6_f32x2){r.z, r.w}, (c6_f32x2){nv.z, nv.w}, a);
                                A.a[u][e] = a;
                            } else {
                                float a = A.a[u][e];
                                a = fmaf(r.x, nv.x, a); a = fmaf(r.y, nv.y, a); a = fmaf(r.z, nv.z, a); a = fmaf(r.w, nv.w, a);
                                A.a[u][e] = a;
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <bool PK>
__device__ __forceinline__ void c6_zero(C6Acc<PK> &A)
{
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 21; ++e) {
            if constexpr (PK) A.a[u][e] = (c6_f32x2){0.f, 0.f};
            else A.a[u][e] = 0.f;
        }
}
template <int D, bool PK>
__device__ __forceinline__ void c6_ranges(int wave, const float4 *hb, const float4 *rb, C6Acc<PK> &A)
{
    constexpr int NB = B2F_C6_BATCH;
    switch (wave) {
    case 0: corr6_range<D, 0, 20, NB, PK>(hb, rb, A); break;
    case 1: corr6_range<D, 20, 20, NB, PK>(hb, rb, A); break;
    case 2: corr6_range<D, 40, 20, NB, PK>(hb, rb, A); break;
    default: corr6_range<D, 60, 21, NB, PK>(hb, rb, A); break;
    }
}
// scale by 1/C (output:div(N), CostVolMulti.lua:100) and store the wave's channels 20 w .. 20 w + 19 (+ 80) of direction d of the
// pixel pair: five 16-byte pieces of the record's 32-byte chunks per pixel
template <bool POW2, bool PK>
__device__ __forceinline__ void corr6_store(const CorrLaunch &p, const C5Tile &t, int d, int w, int pr, int lx, C6Acc<PK> &A)
{
    using namespace v5;
    const float cf = (float)p.C, inv = 1.f / cf;
    // the addresses are formed here, after the FMA loops (hoisted in front of them they are spilled for the whole stage)
    asm volatile("" : "+v"(lx), "+v"(pr));
    const int px = t.x0 + lx;
    float *ob = p.out + (size_t)t.b * p.out_img_stride;
    float *obl = ob + (size_t)20 * p.out_chunk_stride;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int py = t.y0 + 2 * pr + u;
        float a[21];
#pragma unroll
        for (int e = 0; e < 21; ++e) {
            float v;
            if constexpr (PK) v = A.a[u][e].x + A.a[u][e].y;
            else v = A.a[u][e];
            a[e] = POW2 ? v * inv : v / cf;
        }
        if (py < p.h && px < p.w && !((B2F_C5_ABLATE & 4) && a[0] != 12345.678f)) {
            const int pix = py * p.w + px;
            const unsigned off = (unsigned)(pix * p.out_pix_stride);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int c = v6::NCW * w + 4 * i;                 // wave-uniform
                float *oc = ob + (size_t)(d * 10 + (c >> 3)) * p.out_chunk_stride + (c & 7) + off;
                *reinterpret_cast<float4 *>(oc) = make_float4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
            }
            if (w == 3) {   // last chunk: [fwd80, bwd80, u, v, ub, vb, 0, 0]
                float *ol = obl + off;
                if (d == 0) {
                    ol[0] = a[20];
                } else {
                    const size_t fp = ((size_t)t.b * p.h * p.w + pix) * 2;
                    float2 f = make_float2(0.f, 0.f), fb = make_float2(0.f, 0.f);
                    if (p.flow) f = *reinterpret_cast<const float2 *>(p.flow + fp);
                    if (p.flow_b) fb = *reinterpret_cast<const float2 *>(p.flow_b + fp);
                    ol[1] = a[20];
                    ol[2] = f.x; ol[3] = f.y;
                    *reinterpret_cast<float4 *>(ol + 4) = make_float4(fb.x, fb.y, 0.f, 0.f);
                }
            }
        }
    }
    c6_zero<PK>(A);
}

template <bool POW2>
__global__ __launch_bounds__(512) void warp_costvol_spec_kernel(const CorrLaunch p, const int ntd, const int tiles_x, const int tiles_y)
{
    using namespace v5;
    using v6::NF;
    using v6::NG;
    using v6::GT;
    using v6::NIT;
    using v6::NRJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *L = reinterpret_cast<float4 *>(smem);
    float4 *win = L + OFF_WIN;
    float4 *halo = L + OFF_HALO;
    float4 *refb = L + OFF_REF;
    float4 *recs = L + OFF_REC;
    int *bbox = reinterpret_cast<int *>(L + v6::OFF_BB);         // [NREC][NG][4]
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int ncg = p.C / CGC;
    const int nmine = (ntd - (int)blockIdx.x + G - 1) / G;
    const int S = nmine * ncg;
    const bool gth = wave >= NF;                                  // gather waves
    const int gw = wave - NF, gt = tid - NF * 64;                 // gather wave / thread index
    const int ghp = gt >> 1, ghh = gt & 1;                        // blend items of a gather thread: halo pixels ghp + (GT / 2) i, half ghh
    const int pr = lane >> 4, lx = lane & 15;
#if B2F_C5_TRACE
    const bool tr_on = blockIdx.x == 40 && lane == 0 && (wave == 0 || wave == 3 || wave == 4 || wave == 7) && nmine >= 50;
    const int tslot = wave == 0 ? 0 : wave == 3 ? 1 : wave == 4 ? 2 : 3;
#endif

    const int step = G >> 3;
    int q_lin, q_tx, q_ty, q_b, q_k = 0;
    C5Tile t0, t1, t2, t3, t4;
    {
        q_lin = xcd_remap((int)blockIdx.x, ntd);
        const int tile = q_lin >> 1;
        q_tx = tile % tiles_x;
        q_ty = (tile / tiles_x) % tiles_y;
        q_b = tile / (tiles_x * tiles_y);
    }
    C5_NEWEST(t0); C5_ADVANCE(); C5_NEWEST(t1); C5_ADVANCE(); C5_NEWEST(t2); C5_ADVANCE(); C5_NEWEST(t3); C5_ADVANCE(); C5_NEWEST(t4);
#define C6_BB(slot_) (bbox + ((slot_) % NREC) * (4 * NG))
#define C6_BLEND(tt_, cgn_, k1_, hbuf_)                                                                              \
    do {                                                                                                             \
        int wx0__, wy0__;                                                                                            \
        bool fits__;                                                                                                 \
        c5_bbox<NG, true>(p, C6_BB(k1_), wx0__, wy0__, fits__);                                                      \
        fits__ = fits__ && win_ok;                                                                                   \
        float4 tp__[NIT][4 * NCC];                                                                                   \
        C5Samp sm__[NIT];                                                                                            \
        _Pragma("unroll") for (int i__ = 0; i__ < NIT; ++i__)                                                        \
            c5_taps(p, tt_, cgn_, recs + ((k1_) % NREC) * NHP, fits__, wx0__, wy0__, win + (hbuf_) * WIN_F4, ghp + (GT / 2) * i__, ghh, tp__[i__], sm__[i__]); \
        _Pragma("unroll") for (int i__ = 0; i__ < NIT; ++i__)                                                        \
            c5_blend(halo + (hbuf_) * HALO_F4, ghp + (GT / 2) * i__, ghh, tp__[i__], sm__[i__]);                     \
    } while (0)

    // ---- prologue (gather waves): records, windows of stages 0 and 1, reference tile and warped halo of stage 0
    const bool win_ok = p.w >= WC && p.h >= WR;                   // smaller maps: always the gather
    int woff[(WPIECES + NG - 1) / NG], roff[(2 * NK4 + NG - 1) / NG];
    if (gth) {
        c6_win_offsets<NG>(p, gw, lane, woff);
        c6_ref_offsets<NG>(p, gw, lane, roff);
    }
    float2 fa = make_float2(0.f, 0.f), fb = fa, fc = fa;
    if (gth) {
        for (int q = 0; q < nmine && q * ncg <= 2; ++q) {
            C5Tile tt;
            C5_PICK(tt, q);
            c5_rec_issue<NRJ, GT>(p, tt, gt, fa, fb, fc);
            c5_rec_finish<NRJ, GT, true>(p, tt, gt, fa, fb, fc, recs + (q % NREC) * NHP, C6_BB(q) + gw * 4);
        }
        if (3 % ncg == 0 && 3 / ncg < nmine) {
            C5Tile tt;
            C5_PICK(tt, 3 / ncg);
            c5_rec_issue<NRJ, GT>(p, tt, gt, fa, fb, fc);
        }
    }
    __syncthreads();
    if (gth) {
        int wx0, wy0, dk, cgo;
        bool fits;
        c5_bbox<NG, true>(p, C6_BB(0), wx0, wy0, fits);
        if (fits && win_ok) c6_win_dma<NG>(p, t0, 0, wx0, wy0, lds0 + 16u * OFF_WIN, gw, woff);
        if (1 < S) {
            C5_AHEAD(1, 0, dk, cgo);
            C5Tile tt;
            C5_PICK(tt, dk);
            c5_bbox<NG, true>(p, C6_BB(dk), wx0, wy0, fits);
            if (fits && win_ok) c6_win_dma<NG>(p, tt, cgo, wx0, wy0, lds0 + 16u * (OFF_WIN + WIN_F4), gw, woff);
        }
        c6_ref_dma<NG>(p, t0, 0, lds0 + 16u * OFF_REF, gw, lane, roff);
        C5_VM_DRAIN();
    }
    __syncthreads();
    if (gth) C6_BLEND(t0, 0, 0, 0);
    __syncthreads();

    if (gth) {
        // ---- gather waves: stage s prepares s + 1 (halo, reference tile), s + 2 (window), s + 3 / s + 4 (records, flows)
        int s = 0;
        for (int k = 0; k < nmine; ++k) {
            for (int cg = 0; cg < ncg; ++cg, ++s) {
                const bool last_cg = cg + 1 == ncg;
                C5_T(0);
                if (s + 2 < S) {     // window of stage s + 2 -> win[s & 1] (read by the blend of the previous stage)
                    int dk, cgo, wx0, wy0;
                    bool fits;
                    C5_AHEAD(2, cg, dk, cgo);
                    C5Tile tt;
                    C5_PICK(tt, dk);
                    c5_bbox<NG, true>(p, C6_BB(k + dk), wx0, wy0, fits);
                    if (fits && win_ok) c6_win_dma<NG>(p, tt, cgo, wx0, wy0, lds0 + 16u * (OFF_WIN + (s & 1) * WIN_F4), gw, woff);
                }
                float2 na = fa, nb = fb, nc = fc;
                {
                    int dk, cgo;
                    C5_AHEAD(4, cg, dk, cgo);
                    if (cgo == 0 && k + dk < nmine) {
                        C5Tile tt;
                        C5_PICK(tt, dk);
                        c5_rec_issue<NRJ, GT>(p, tt, gt, na, nb, nc);
                    }
                }
                C5_T(1);
                if (s + 1 < S) {
                    C5Tile tt;
                    C5_PICK(tt, last_cg ? 1 : 0);
                    c6_ref_dma<NG>(p, tt, last_cg ? 0 : cg + 1, lds0 + 16u * (OFF_REF + ((s + 1) & 1) * REF_F4), gw, lane, roff);
                    C5_T(2);
                    const int k1 = last_cg ? k + 1 : k;
                    C6_BLEND(tt, last_cg ? 0 : cg + 1, k1, (s + 1) & 1);
                }
                C5_T(3);
                {
                    int dk, cgo;
                    C5_AHEAD(3, cg, dk, cgo);
                    if (cgo == 0 && k + dk < nmine && (k + dk) * ncg > 2 && !(B2F_C5_ABLATE & 64)) {
                        C5Tile tt;
                        C5_PICK(tt, dk);
                        c5_rec_finish<NRJ, GT, true>(p, tt, gt, fa, fb, fc, recs + ((k + dk) % NREC) * NHP, C6_BB(k + dk) + gw * 4);
                    }
                    fa = na; fb = nb; fc = nc;
                }
                C5_T(4);
                C5_VM_DRAIN();
                C5_T(5);
                C5_LDS_BARRIER();
                C5_T(6);
            }
            t0 = t1; t1 = t2; t2 = t3; t3 = t4;
            C5_ADVANCE();
            C5_NEWEST(t4);
        }
    } else {
        // ---- F waves: the stage's FMAs from LDS, the stores of a finished tile-direction
        constexpr bool PK = B2F_C6_PACKED != 0;
        C6Acc<PK> A;
        c6_zero<PK>(A);
        int s = 0;
        for (int k = 0; k < nmine; ++k) {
            for (int cg = 0; cg < ncg; ++cg, ++s) {
                const float4 *hb = halo + (s & 1) * HALO_F4 + (2 * pr + R) * HW + (lx + R);
                const float4 *rb = refb + (s & 1) * REF_F4 + (2 * pr) * TW + lx;
                C5_T(0); C5_T(1); C5_T(2);
                if (!(B2F_C5_ABLATE & 2)) {
                    if (t0.dir == 0) c6_ranges<0, PK>(wave, hb, rb, A);
                    else c6_ranges<1, PK>(wave, hb, rb, A);
                }
                C5_T(3); C5_T(4);
                if (cg + 1 == ncg) corr6_store<POW2, PK>(p, t0, t0.dir, wave, pr, lx, A);
                C5_T(5);
                C5_LDS_BARRIER();
                C5_T(6);
            }
            t0 = t1; t1 = t2; t2 = t3; t3 = t4;
            C5_ADVANCE();
            C5_NEWEST(t4);
        }
    }
#undef C6_BLEND
#undef C6_BB
}

// ---- sixteen waves (corr_variant 7): ten unit waves + six gather waves ----------------------------------------------------
// The two lessons of variants 5 and 6 put together: the FMAs want the ten 8-channel unit waves of variant 5 (a SIMD needs three
// issuing waves to retire an FMA every 1.67 cycles; one wave issues one every 5.5), and everything that waits for memory wants
// to sit in waves that do nothing else.  Block = 1 024 threads: waves 0..9 compute one unit per stage from LDS (and store a
// finished tile-direction), waves 10..15 gather the taps of the NEXT stage's warped halo straight from memory (two items = 16
// loads per thread in flight, no window), blend them into the other halo buffer, DMA the reference tile and compute the sampling
// records a stage ahead.  128 registers per thread; one LDS-only barrier per stage.
#endif  // B2F_EXPERIMENTS

namespace v7 {
constexpr int NF = 10, NG = 6;
constexpr int NTHR = 64 * (NF + NG);            // 1 024
constexpr int GT = 64 * NG;                     // 384 gather threads
constexpr int NIT = 2 * v5::NHP / GT;           // blend items per gather thread and stage: 2
constexpr int NRJ = v5::NHP / GT;               // sampling records per gather thread and tile-direction: 1
static_assert(2 * v5::NHP % GT == 0 && v5::NHP % GT == 0 && GT % 2 == 0, "items divide over the gather threads");
constexpr int OFF_HALO = 0, OFF_REF = OFF_HALO + 2 * v5::HALO_F4, OFF_REC = OFF_REF + 2 * v5::REF_F4;
constexpr int LDS_BYTES = 16 * (OFF_REC + v5::NREC * v5::NHP);   // 84 480
}  // namespace v7

template <bool POW2>
__global__ __launch_bounds__(1024) void warp_costvol_gw_kernel(const CorrLaunch p, const int ntd, const int tiles_x, const int tiles_y)
{
    using namespace v5;
    using v7::NF;
    using v7::NG;
    using v7::GT;
    using v7::NIT;
    using v7::NRJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *L = reinterpret_cast<float4 *>(smem);
    float4 *halo = L + v7::OFF_HALO;
    float4 *refb = L + v7::OFF_REF;
    float4 *recs = L + v7::OFF_REC;
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int ncg = p.C / CGC;
    const int nmine = (ntd - (int)blockIdx.x + G - 1) / G;
    const int S = nmine * ncg;
    const bool gth = wave >= NF;
    const int gw = wave - NF, gt = tid - NF * 64;
    const int ghp = gt >> 1, ghh = gt & 1;                        // blend items of a gather thread: halo pixels ghp + (GT / 2) i, half ghh
    const int pr = lane >> 4, lx = lane & 15;

    const int step = G >> 3;
    int q_lin, q_tx, q_ty, q_b, q_k = 0;
    C5Tile t0, t1, t2, t3, t4;
    {
        q_lin = xcd_remap((int)blockIdx.x, ntd);
        const int tile = q_lin >> 1;
        q_tx = tile % tiles_x;
        q_ty = (tile / tiles_x) % tiles_y;
        q_b = tile / (tiles_x * tiles_y);
    }
    C5_NEWEST(t0); C5_ADVANCE(); C5_NEWEST(t1); C5_ADVANCE(); C5_NEWEST(t2); C5_ADVANCE(); C5_NEWEST(t3); C5_ADVANCE(); C5_NEWEST(t4);
#define C7_TAPS(tt_, cgn_, k1_)                                                                                      \
    _Pragma("unroll") for (int i__ = 0; i__ < NIT; ++i__)                                                            \
        if (B2F_C5_ABLATE & 1) { _Pragma("unroll") for (int q__ = 0; q__ < 4 * NCC; ++q__) tp[i__][q__] = make_float4(1.f, 2.f, 3.f, 4.f); sm[i__].w4 = make_float4(.25f, .25f, .25f, .25f); } else \
        c5_taps(p, tt_, cgn_, recs + ((k1_) % NREC) * NHP, false, 0, 0, nullptr, ghp + (GT / 2) * i__, ghh, tp[i__], sm[i__])
#define C7_BLEND(hbuf_)                                                                                              \
    _Pragma("unroll") for (int i__ = 0; i__ < NIT; ++i__)                                                            \
        c5_blend(halo + (hbuf_) * HALO_F4, ghp + (GT / 2) * i__, ghh, tp[i__], sm[i__])

    // ---- prologue (gather waves): records of the first tile-directions, reference tile and warped halo of stage 0
    float2 fa = make_float2(0.f, 0.f), fb = fa, fc = fa;
    if (gth) {
        for (int q = 0; q < nmine && q * ncg <= 2; ++q) {
            C5Tile tt;
            C5_PICK(tt, q);
            c5_rec_issue<NRJ, GT>(p, tt, gt, fa, fb, fc);
            c5_rec_finish<NRJ, GT, false>(p, tt, gt, fa, fb, fc, recs + (q % NREC) * NHP, nullptr);
        }
        if (3 % ncg == 0 && 3 / ncg < nmine) {
            C5Tile tt;
            C5_PICK(tt, 3 / ncg);
            c5_rec_issue<NRJ, GT>(p, tt, gt, fa, fb, fc);
        }
        c5_ref_dma<NG>(p, t0, 0, lds0 + 16u * v7::OFF_REF, gw, lane);
    }
    __syncthreads();
    if (gth) {
        float4 tp[NIT][4 * NCC];
        C5Samp sm[NIT];
        C7_TAPS(t0, 0, 0);
        C7_BLEND(0);
        C5_VM_DRAIN();
    }
    __syncthreads();

    if (gth) {
        int s = 0;
        for (int k = 0; k < nmine; ++k) {
            for (int cg = 0; cg < ncg; ++cg, ++s) {
                const bool last_cg = cg + 1 == ncg;
                // flows of the tile-direction that stage s + 4 opens (consumed a stage later), reference tile of stage s + 1
                float2 na = fa, nb = fb, nc = fc;
                {
                    int dk, cgo;
                    C5_AHEAD(4, cg, dk, cgo);
                    if (cgo == 0 && k + dk < nmine) {
                        C5Tile tt;
                        C5_PICK(tt, dk);
                        c5_rec_issue<NRJ, GT>(p, tt, gt, na, nb, nc);
                    }
                }
                float4 tp[NIT][4 * NCC];
                C5Samp sm[NIT];
                if (s + 1 < S) {
                    C5Tile tt;
                    C5_PICK(tt, last_cg ? 1 : 0);
                    c5_ref_dma<NG>(p, tt, last_cg ? 0 : cg + 1, lds0 + 16u * (v7::OFF_REF + ((s + 1) & 1) * REF_F4), gw, lane);
                    const int k1 = last_cg ? k + 1 : k;
                    C7_TAPS(tt, last_cg ? 0 : cg + 1, k1);            // 16 loads per thread in flight under the record pass
                }
                {   // sampling records of the tile-direction that stage s + 3 opens (flows issued a stage ago)
                    int dk, cgo;
                    C5_AHEAD(3, cg, dk, cgo);
                    if (cgo == 0 && k + dk < nmine && (k + dk) * ncg > 2) {
                        C5Tile tt;
                        C5_PICK(tt, dk);
                        c5_rec_finish<NRJ, GT, false>(p, tt, gt, fa, fb, fc, recs + ((k + dk) % NREC) * NHP, nullptr);
                    }
                    fa = na; fb = nb; fc = nc;
                }
                if (s + 1 < S) C7_BLEND((s + 1) & 1);
                C5_VM_DRAIN();
                C5_LDS_BARRIER();
            }
            t0 = t1; t1 = t2; t2 = t3; t3 = t4;
            C5_ADVANCE();
            C5_NEWEST(t4);
        }
    } else {
        float acc[18];
#pragma unroll
        for (int e = 0; e < 18; ++e) acc[e] = 0.f;
        int s = 0;
        for (int k = 0; k < nmine; ++k) {
            for (int cg = 0; cg < ncg; ++cg, ++s) {
                const float4 *hb = halo + (s & 1) * HALO_F4 + (2 * pr + R) * HW + (lx + R);
                const float4 *rb = refb + (s & 1) * REF_F4 + (2 * pr) * TW + lx;
                if (!(B2F_C5_ABLATE & 2)) {
                    if (t0.dir == 0) c5_units<0>(wave, hb, rb, acc);
                    else c5_units<1>(wave, hb, rb, acc);
                }
                if (cg + 1 == ncg) corr5_store<POW2>(p, t0, t0.dir, wave, pr, lx, acc);
                C5_LDS_BARRIER();
            }
            t0 = t1; t1 = t2; t2 = t3; t3 = t4;
            C5_ADVANCE();
            C5_NEWEST(t4);
        }
    }
#undef C7_BLEND
#undef C7_TAPS
#undef C5_VM_DRAIN
#undef C5_LDS_BARRIER
#undef C5_AHEAD
#undef C5_PICK
#undef C5_ADVANCE
#undef C5_NEWEST
}

bool warp_costvol_unit_supported(const CorrLaunch &p)
{
    // 12-bit tap coordinates in the sampling records; 32-bit byte offsets inside an image (fallback's buffer loads)
    return p.C >= v5::CGC && p.C % v5::CGC == 0 && p.w <= 4096 && p.h <= 4096 && (double)p.img_stride * 4.0 < 2147483648.0 &&
           (double)(p.C / 8) * (double)p.chunk_stride * 4.0 + (double)p.h * p.w * p.pix_stride * 4.0 < 2147483648.0 &&
           (double)p.h * p.w * p.out_pix_stride * 4.0 < 2147483648.0;
}

#if B2F_EXPERIMENTS
hipError_t launch_warp_costvol_spec(const CorrLaunch &p, hipStream_t s)
{
    using namespace v5;
    if (!warp_costvol_unit_supported(p)) return hipErrorInvalidValue;
    static bool attr_done_dev[64][2] = {{false}};
    static int n_cu_dev[64] = {0};
    const int slot = attr_slot();
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    if (!attr_done_dev[slot][pow2 ? 1 : 0]) {
        const void *fn = pow2 ? reinterpret_cast<const void *>(&warp_costvol_spec_kernel<true>) : reinterpret_cast<const void *>(&warp_costvol_spec_kernel<false>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, v6::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done_dev[slot][pow2 ? 1 : 0] = true;
    }
    if (!n_cu_dev[slot]) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n &= ~7;
        n_cu_dev[slot] = n < 8 ? 8 : n;
    }
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    const int ntd = 2 * tiles_x * tiles_y * p.B;
    int grid = n_cu_dev[slot] < ntd ? n_cu_dev[slot] : ntd;
    if (grid >= 8) grid &= ~7;
    if (pow2) hipLaunchKernelGGL((warp_costvol_spec_kernel<true>), dim3((unsigned)grid), dim3(v6::NTHR), v6::LDS_BYTES, s, p, ntd, tiles_x, tiles_y);
    else hipLaunchKernelGGL((warp_costvol_spec_kernel<false>), dim3((unsigned)grid), dim3(v6::NTHR), v6::LDS_BYTES, s, p, ntd, tiles_x, tiles_y);
#if B2F_C5_TRACE
    static int traced = 0;
    if (ntd / grid >= 50 && traced++ == 2) {
        static long long h[4 * 64 * 8];
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(c5_trace_buf), sizeof h);
        const char *nm[4] = {"F wave 0", "F wave 3", "G wave 4", "G wave 7"};
        for (int w = 0; w < 4; ++w) {
            fprintf(stderr, "c6 trace %s (C %d, %d x %d): per stage, cycles: top | G: window DMA + flows | ref DMA | blend (F: FMAs) | records | vmcnt(0) (F: stores) | barrier\n", nm[w], p.C, p.h, p.w);
            for (int st = 1; st < 24; ++st) {
                const long long *t = h + (w * 64 + st) * 8, *tp = h + (w * 64 + st - 1) * 8;
                fprintf(stderr, "  stage %2d  %6lld |", st, t[0] - tp[6]);
                for (int q = 1; q <= 6; ++q) fprintf(stderr, " %6lld", t[q] - t[q - 1]);
                fprintf(stderr, "   total %6lld\n", t[6] - tp[6]);
            }
        }
    }
#endif
    return hipGetLastError();
}

#endif  // B2F_EXPERIMENTS

hipError_t launch_warp_costvol_gw(const CorrLaunch &p, hipStream_t s)
{
    using namespace v5;
    if (!warp_costvol_unit_supported(p)) return hipErrorInvalidValue;
    static bool attr_done_dev[64][2] = {{false}};
    static int n_cu_dev[64] = {0};
    const int slot = attr_slot();
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    if (!attr_done_dev[slot][pow2 ? 1 : 0]) {
        const void *fn = pow2 ? reinterpret_cast<const void *>(&warp_costvol_gw_kernel<true>) : reinterpret_cast<const void *>(&warp_costvol_gw_kernel<false>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, v7::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done_dev[slot][pow2 ? 1 : 0] = true;
    }
    if (!n_cu_dev[slot]) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n &= ~7;
        n_cu_dev[slot] = n < 8 ? 8 : n;
    }
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    const int ntd = 2 * tiles_x * tiles_y * p.B;
    int grid = n_cu_dev[slot] < ntd ? n_cu_dev[slot] : ntd;
    if (grid >= 8) grid &= ~7;
    if (pow2) hipLaunchKernelGGL((warp_costvol_gw_kernel<true>), dim3((unsigned)grid), dim3(v7::NTHR), v7::LDS_BYTES, s, p, ntd, tiles_x, tiles_y);
    else hipLaunchKernelGGL((warp_costvol_gw_kernel<false>), dim3((unsigned)grid), dim3(v7::NTHR), v7::LDS_BYTES, s, p, ntd, tiles_x, tiles_y);
    return hipGetLastError();
}

hipError_t launch_warp_costvol_unit(const CorrLaunch &p, hipStream_t s)
{
    using namespace v5;
    constexpr int U_LDS = LDS_BYTES - (B2F_C5_WINDOW ? 0 : 16 * 2 * WIN_F4);
    if (!warp_costvol_unit_supported(p)) return hipErrorInvalidValue;
    static bool attr_done_dev[64][2] = {{false}};
    static int n_cu_dev[64] = {0};
    const int slot = attr_slot();
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    if (!attr_done_dev[slot][pow2 ? 1 : 0]) {
        const void *fn = pow2 ? reinterpret_cast<const void *>(&warp_costvol_unit_kernel<true>) : reinterpret_cast<const void *>(&warp_costvol_unit_kernel<false>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, U_LDS);
        if (e != hipSuccess) return e;
        attr_done_dev[slot][pow2 ? 1 : 0] = true;
    }
    if (!n_cu_dev[slot]) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        n &= ~7;                                  // the XCD banding of the logical index wants a multiple of 8
        n_cu_dev[slot] = n < 8 ? 8 : n;
    }
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    const int ntd = 2 * tiles_x * tiles_y * p.B;
    int grid = B2F_C5_BLOCKS * n_cu_dev[slot] < ntd ? B2F_C5_BLOCKS * n_cu_dev[slot] : ntd;
    if (grid >= 8) grid &= ~7;
    if (pow2) hipLaunchKernelGGL((warp_costvol_unit_kernel<true>), dim3((unsigned)grid), dim3(NTHR), U_LDS, s, p, ntd, tiles_x, tiles_y);
    else hipLaunchKernelGGL((warp_costvol_unit_kernel<false>), dim3((unsigned)grid), dim3(NTHR), U_LDS, s, p, ntd, tiles_x, tiles_y);
#if B2F_C5_TRACE
    static int traced = 0;
    if (ntd / grid >= 50 && traced++ == 2) {
        static long long h[4 * 64 * 8];
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(c5_trace_buf), sizeof h);
        const char *nm[4] = {"wave 0", "wave 4", "wave 9", "wave 10 (aux)"};
        for (int w = 0; w < 4; ++w) {
            fprintf(stderr, "c5 trace %s (C %d, %d x %d): per stage, cycles: top | DMA issue | blend | FMAs / records | vmcnt(0) | stores | barrier\n", nm[w], p.C, p.h, p.w);
            for (int st = 1; st < 24; ++st) {
                const long long *t = h + (w * 64 + st) * 8, *tp = h + (w * 64 + st - 1) * 8;
                fprintf(stderr, "  stage %2d  %6lld |", st, t[0] - tp[6]);
                for (int q = 1; q <= 6; ++q) fprintf(stderr, " %6lld", t[q] - t[q - 1]);
                fprintf(stderr, "   total %6lld\n", t[6] - tp[6]);
            }
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace b2f
