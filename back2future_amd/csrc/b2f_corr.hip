// Fused bilinear warp + 9x9 multi-frame cost volume for gfx950 (MI355X).
//
// Replaces, per pyramid level, the reference's
//   ws[f][l]  = warpingUnit(cs[f][l], ufs[l+1] * 20(f-2)/2^(l-1))     pwc.lua:68-73,393-409
//               -> nn.BilinearSamplerBHWD CUDA kernel                   BilinearSamplerBHWD.cu:41-115
//   cvs_fwd   = nn.CostVolMulti(9, true ){cs[2][l], ws[3][l]}          pwc.lua:246-252
//   cvs_bwd   = nn.CostVolMulti(9, false){cs[2][l], ws[1][l]}          pwc.lua:257-263
//   JoinTable(2){fwd, bwd}                                              pwc.lua:267
// (4 transposing copies + 2 sampler launches + 2 x >=243 tensor-op launches in the
// reference, CostVolMulti.lua:62-100) by ONE launch that never materializes the warped
// maps: HBM-bound, algorithmic traffic (3C + 2 + 162) * 4 B per level pixel.
//
// Maths (channel order inside a volume is x-major, c = (qx+4)*9 + (qy+4), CostVolMulti.lua:66-67,92):
//   fwd[c] = 1/C * sum_k ref[y,x,k] * W3[y-qy, x-qx, k]      (out of range -> 0)
//   bwd[c] = 1/C * sum_k ref[y,x,k] * W1[y+qy, x+qx, k]
// with W3/W1 = neighbour map sampled at (x + k*u, y + k*v) / (x - k*u, y - k*v), coordinates
// clamped to the border, top-left weight 1 - frac (BilinearSamplerBHWD.cu:6-20).
// Output: one 168-float record per pixel, slot order [fwd 0..79 | bwd 0..79 | fwd80 bwd80 u v ub vb 0 0]
// (b2f_internal.h), stored chunk-planar so that a wave's stores fill whole cache lines.
//
// Block = 256 threads, output tile 8 x 16 pixels.  Threads 0..127 own one pixel of the fwd
// volume each, threads 128..255 the same pixels of the bwd volume: 81 accumulators in VGPRs.
// Channels are walked in 8-channel chunks (= one plane of the chunk-planar feature maps): the
// block gathers the warped 16 x 24 halo of both neighbour maps into LDS (layout [k4][row pitch
// 32] of float4: a wave's ds_read_b128 are conflict-free and every displacement is an immediate
// offset; consecutive halo pixels are 32 B apart in HBM, so a wave's gather touches ~8 lines
// instead of 32), then every thread does 81 x (1 ds_read_b128 + 4 FMA) per float4 of its
// reference pixel, software-pipelined one 9-displacement column ahead.  The bwd thread runs the
// same code on the mirrored window (bwd channel c uses offset +q = fwd offset of channel 80 - c).
// Blocks are remapped so that each XCD (private L2) works on one contiguous band of tiles.
//
// Measured dead ends (round 1, MI355X, kept out of the tree): 3-pixel register
// blocking (fewer LDS reads: slower, LDS was never the limiter), 3 threads per pixel-direction
// (more loads in flight: +6 %), software-pipelining the gather across chunks (no gain), producer /
// consumer wave specialization (2x slower: the gather is VALU-issue-bound, not latency-bound),
// v_pk_fma_f32 on (fwd, bwd) pairs with interleaved maps (slower: packed FMA issues at half rate).
// What did pay: chunk-planar inputs/outputs (whole-line gathers and stores), XCD remap, a cheap
// gather (per-pixel state in LDS, one thread fetches both float4 of a chunk, fma blend).
#include "b2f_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace b2f {

namespace {
constexpr int TH = 8, TW = 16, R = 4;
constexpr int HH = TH + 2 * R;      // 16 halo rows
constexpr int HWD = TW + 2 * R;     // 24 halo cols
constexpr int HP = 32;              // LDS row pitch in pixels (multiple of 16 -> conflict-free b128)
constexpr int NHALO = HH * HWD;     // 384

// Bilinear sampling record of one halo pixel of one neighbour map: the four blend
// weights (all 0 for a halo pixel outside the image: CostVolMulti's out-of-range -> 0) and
// the top-left pixel index with the offsets of the right / bottom neighbours.  A neighbour
// outside the image has weight exactly 0 (coordinates are clamped first), so its address is
// folded onto the clamped pixel instead of branching around the load.
struct SampIdx {
    int idx;           // top-left pixel index (y*w + x)
    int flags;         // bit0: right neighbour is x+1 (else folded onto x), bit1: bottom is y+1
};
}  // namespace

// Profiling builds only (-DB2F_CORR_TRACE): clock64() stamps of block 0's waves for their first 64 items
#ifdef B2F_CORR_TRACE
__device__ long long g_corr_trace[8 * 64 * 4];
#define CORR_T(item_, k_) do { if (blockIdx.x == 0 && lane == 0 && (item_) < 64) g_corr_trace[((wave * 64) + (item_)) * 4 + (k_)] = clock64(); } while (0)
#else
#define CORR_T(item_, k_) do {} while (0)
#endif

// one-pixel / two-pixel VALU kernels: stamps of block 0, wave 0 (slot = chunk)
#ifdef B2F_CORR_TRACE
#define CORR_TV(ch_, k_) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (ch_) < 64) g_corr_trace[(ch_) * 8 + (k_)] = clock64(); } while (0)
#else
#define CORR_TV(ch_, k_) do {} while (0)
#endif

__device__ __forceinline__ void top_left(float coord, int size, int &pt, float &wt)
{
    // getTopLeft, BilinearSamplerBHWD.cu:6-20
    float c = coord;
    if (c < 0.f) c = 0.f;
    if (c > (float)(size - 1)) c = (float)(size - 1);
    const float fl = floorf(c);
    pt = (int)fl;
    wt = 1.f - (c - fl);
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8 (8 XCDs, private L2s), so
// with the natural order the 3x halo overlap of neighbouring tiles is re-fetched from the fabric
// by up to 8 different L2s.  Give every XCD one contiguous band of tiles instead (bijective for
// any grid size; placement only affects speed, never results).
// LAT: variant for launches that cannot fill the chip anyway (a single triplet, the coarse levels): two blocks per CU
// instead of three buy the registers to issue all 24 gather loads of a chunk at once -- one memory round trip per
// chunk instead of three, which is most of what a lone block's run time consists of.  (Also tried: one block per CU
// with the next chunk's loads issued before the FMA phase -- slower, the 96 extra live registers go through AGPRs.)
template <bool POW2, bool LAT>
__global__ __launch_bounds__(256, LAT ? 2 : 3) void warp_costvol_kernel(const CorrLaunch p)
{
    __shared__ __attribute__((aligned(16))) float4 nb[2][2][HH * HP];   // [map][k4][pixel] 32 KB
    __shared__ float4 samp_w[2][NHALO];                                  // 12 KB
    __shared__ SampIdx samp_i[2][NHALO];                                 // 6 KB

    const int tid = threadIdx.x;
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    int bid = (p.ablate & 8) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx_i * TW, y0 = ty_i * TH;

    const float *ref = p.ref + (size_t)b * p.img_stride;
    const float *nbr[2] = {p.nbr_fut + (size_t)b * p.img_stride, p.nbr_past + (size_t)b * p.img_stride};

    // ---- sampling records for the halo (once per block) ----
    // 2 * NHALO = 768 = 3 records per thread; the three flow loads (clamped address, no branch) are issued
    // together: behind a condition they were three serialized memory round trips per block
    {
        float2 fl[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + j * 256;
            const int map = i / NHALO, hp = i - map * NHALO;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int y = y0 - R + hy, x = x0 - R + hx;
            const bool in = y >= 0 && y < p.h && x >= 0 && x < p.w;
            fl[j] = make_float2(0.f, 0.f);
            if (p.flow) fl[j] = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * p.h * p.w + (in ? (size_t)y * p.w + x : 0)) * 2);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + j * 256;
            const int map = i / NHALO, hp = i - map * NHALO;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int y = y0 - R + hy, x = x0 - R + hx;
            float4 wgt = make_float4(0.f, 0.f, 0.f, 0.f);
            SampIdx si;
            si.idx = 0; si.flags = 0;
            if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
                const float k = map == 0 ? p.k : -p.k;   // nn.MulConstant(20*(f-ref)/2^(l-2)), pwc.lua:404
                const float u = fl[j].x * k, v = fl[j].y * k;
                int xl, yt;
                float wx, wy;
                top_left(u + (float)x, p.w, xl, wx);
                top_left(v + (float)y, p.h, yt, wy);
                si.idx = (yt * p.w + xl) * p.pix_stride;   // float offset of the top-left tap inside the image plane
                si.flags = ((xl + 1 <= p.w - 1) ? 1 : 0) | ((yt + 1 <= p.h - 1) ? 2 : 0);
                wgt = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
            }
            si.flags |= (hy * HP + hx) << 2;               // LDS slot of this halo pixel
            samp_w[map][hp] = wgt;
            samp_i[map][hp] = si;
        }
    }

    float acc[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) acc[i] = 0.f;

    const int pl = tid & 127, dir = tid >> 7;
    const int ly = pl >> 4, lx = pl & 15;
    const int py = y0 + ly, px = x0 + lx;
    const bool pvalid = py < p.h && px < p.w;
    const float *refp = ref + (size_t)(pvalid ? (py * p.w + px) : 0) * p.pix_stride;
    const float4 *myn = &nb[dir][0][(ly + R) * HP + (lx + R)];

    __syncthreads();
    const int nchunk = p.C >> 3;
    for (int ch = 0; ch < nchunk; ++ch) {
        const size_t coff = (size_t)ch * p.chunk_stride;
        // ---- gather + blend the warped halo chunk into LDS ----
        // 2 maps x 384 halo pixels = 768 = 3 per thread; a thread fetches both float4 of the chunk
        // for its halo pixel (the sampling record, its only per-pixel state, lives in LDS), 8 loads
        // in flight, blend = 1 mul + 3 fma per component.
        if constexpr (!LAT) {
#pragma unroll 1
            for (int j = 0; j < 3; ++j) {
                const int q = tid + j * 256;
                const int map = q >= NHALO;
                const float4 wg = (&samp_w[0][0])[q];
                const SampIdx si = (&samp_i[0][0])[q];
                const float *src = nbr[map] + coff + si.idx;
                const int dx = (si.flags & 1) * p.pix_stride, dy = (si.flags & 2) ? p.w * p.pix_stride : 0;
                float4 t[8];
                if (!(p.ablate & 1)) {
#pragma unroll
                    for (int k4 = 0; k4 < 2; ++k4) {
                        t[k4 * 4 + 0] = *reinterpret_cast<const float4 *>(src + 4 * k4);
                        t[k4 * 4 + 1] = *reinterpret_cast<const float4 *>(src + dx + 4 * k4);
                        t[k4 * 4 + 2] = *reinterpret_cast<const float4 *>(src + dy + 4 * k4);
                        t[k4 * 4 + 3] = *reinterpret_cast<const float4 *>(src + dy + dx + 4 * k4);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = wg;
                }
                float4 *dst = &nb[map][0][si.flags >> 2];
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4) {
                    const float4 tl = t[k4 * 4], tr = t[k4 * 4 + 1], bl = t[k4 * 4 + 2], br = t[k4 * 4 + 3];
                    float4 v;
                    v.x = fmaf(wg.w, br.x, fmaf(wg.z, bl.x, fmaf(wg.y, tr.x, wg.x * tl.x)));
                    v.y = fmaf(wg.w, br.y, fmaf(wg.z, bl.y, fmaf(wg.y, tr.y, wg.x * tl.y)));
                    v.z = fmaf(wg.w, br.z, fmaf(wg.z, bl.z, fmaf(wg.y, tr.z, wg.x * tl.z)));
                    v.w = fmaf(wg.w, br.w, fmaf(wg.z, bl.w, fmaf(wg.y, tr.w, wg.x * tl.w)));
                    dst[k4 * (HH * HP)] = v;
                }
            }
        } else {
            float4 tt[3][8], wgs[3];
            int slot[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int q = tid + j * 256;
                const int map = q >= NHALO;
                wgs[j] = (&samp_w[0][0])[q];
                const SampIdx si = (&samp_i[0][0])[q];
                const float *src = nbr[map] + coff + si.idx;
                const int dx = (si.flags & 1) * p.pix_stride, dy = (si.flags & 2) ? p.w * p.pix_stride : 0;
                slot[j] = map * (2 * HH * HP) + (si.flags >> 2);
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4) {
                    tt[j][k4 * 4 + 0] = *reinterpret_cast<const float4 *>(src + 4 * k4);
                    tt[j][k4 * 4 + 1] = *reinterpret_cast<const float4 *>(src + dx + 4 * k4);
                    tt[j][k4 * 4 + 2] = *reinterpret_cast<const float4 *>(src + dy + 4 * k4);
                    tt[j][k4 * 4 + 3] = *reinterpret_cast<const float4 *>(src + dy + dx + 4 * k4);
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float4 wg = wgs[j];
                float4 *dst = &nb[0][0][0] + slot[j];
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4) {
                    const float4 tl = tt[j][k4 * 4], tr = tt[j][k4 * 4 + 1], bl = tt[j][k4 * 4 + 2], br = tt[j][k4 * 4 + 3];
                    float4 v;
                    v.x = fmaf(wg.w, br.x, fmaf(wg.z, bl.x, fmaf(wg.y, tr.x, wg.x * tl.x)));
                    v.y = fmaf(wg.w, br.y, fmaf(wg.z, bl.y, fmaf(wg.y, tr.y, wg.x * tl.y)));
                    v.z = fmaf(wg.w, br.z, fmaf(wg.z, bl.z, fmaf(wg.y, tr.z, wg.x * tl.z)));
                    v.w = fmaf(wg.w, br.w, fmaf(wg.z, bl.w, fmaf(wg.y, tr.w, wg.x * tl.w)));
                    dst[k4 * (HH * HP)] = v;
                }
            }
        }
        // reference pixel chunk (address is clamped to a valid pixel for out-of-image lanes)
        const float4 r01[2] = {*reinterpret_cast<const float4 *>(refp + coff), *reinterpret_cast<const float4 *>(refp + coff + 4)};
        __syncthreads();
        // ---- correlate ----
        // 18 steps = 2 k4 x 9 qx columns; the 9 ds_read_b128 of step s+1 are issued before the
        // 36 FMAs of step s (two register sets), so LDS latency hides under the FMAs of the
        // same wave instead of relying on other waves.
        if (!(p.ablate & 2)) {
            float4 va[9], vb[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) va[j] = myn[-(j - 4) * HP + 4];
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                const int k4 = st / 9, g = st - 9 * k4;
                const float4 r = r01[k4];
                float4 *cur = (st & 1) ? vb : va, *nxt = (st & 1) ? va : vb;
                if (st + 1 < 18) {
                    const int k4n = (st + 1) / 9, gn = (st + 1) - 9 * k4n;
#pragma unroll
                    for (int j = 0; j < 9; ++j) nxt[j] = myn[k4n * (HH * HP) - (j - 4) * HP - (gn - 4)];
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float a = acc[g * 9 + j];
                    a = fmaf(r.x, cur[j].x, a);
                    a = fmaf(r.y, cur[j].y, a);
                    a = fmaf(r.z, cur[j].z, a);
                    a = fmaf(r.w, cur[j].w, a);
                    acc[g * 9 + j] = a;
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the steps in order (bounds live ds_read results)
            }
        }
        __syncthreads();
    }

    if (!pvalid) return;
    if ((p.ablate & 4) && acc[0] != 12345.678f) return;   // profiling only: drop the stores, keep acc live
    // ---- scale by 1/C (output:div(N), CostVolMulti.lua:100) and store the record slots ----
    const float cf = (float)p.C, inv = 1.f / cf;
#pragma unroll
    for (int c = 0; c < 81; ++c) acc[c] = POW2 ? acc[c] * inv : acc[c] / cf;
    const size_t pix = (size_t)py * p.w + px;
    float *o = p.out + (size_t)b * p.out_img_stride + pix * p.out_pix_stride;
    // slots dir*80 .. dir*80+79 = ten whole chunks: channel c of this direction.  The bwd thread
    // accumulated the mirrored index, acc[c'] holds bwd channel 80 - c'.
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        float4 lo, hi;
        if (dir == 0) {
            lo = make_float4(acc[8 * j], acc[8 * j + 1], acc[8 * j + 2], acc[8 * j + 3]);
            hi = make_float4(acc[8 * j + 4], acc[8 * j + 5], acc[8 * j + 6], acc[8 * j + 7]);
        } else {
            lo = make_float4(acc[80 - 8 * j], acc[79 - 8 * j], acc[78 - 8 * j], acc[77 - 8 * j]);
            hi = make_float4(acc[76 - 8 * j], acc[75 - 8 * j], acc[74 - 8 * j], acc[73 - 8 * j]);
        }
        float *oc = o + (size_t)(dir * 10 + j) * p.out_chunk_stride;
        *reinterpret_cast<float4 *>(oc) = lo;
        *reinterpret_cast<float4 *>(oc + 4) = hi;
    }
    // last chunk: [fwd80, bwd80, u, v, ub, vb, 0, 0]
    float *ol = o + (size_t)20 * p.out_chunk_stride;
    if (dir == 0) {
        ol[0] = acc[80];
    } else {
        const size_t fp = ((size_t)b * p.h * p.w + pix) * 2;
        float2 f = make_float2(0.f, 0.f), fb = make_float2(0.f, 0.f);
        if (p.flow) f = *reinterpret_cast<const float2 *>(p.flow + fp);
        if (p.flow_b) fb = *reinterpret_cast<const float2 *>(p.flow_b + fp);
        ol[1] = acc[0];
        ol[2] = f.x; ol[3] = f.y;
        *reinterpret_cast<float4 *>(ol + 4) = make_float4(fb.x, fb.y, 0.f, 0.f);
    }
}


// ======================================================================================================
// Two-pixel variant (round 2).  Round 1's correlation phase turned out to be LDS-bandwidth-bound, not VALU-bound:
// one ds_read_b128 per 4 v_fmac is 16 LDS cycles per 8 VALU cycles on each of the four SIMDs sharing the 256 B/clk
// LDS -- 2592 cycles per 8 x 16 tile and chunk against 1296 of VALU issue (PMC: VALU 56 % busy, one instruction
// per 4 cycles).  Here a thread owns TWO vertically adjacent pixels: a neighbour row serves qy of the upper pixel and
// qy + 1 of the lower one, so 10 rows x 9 reads feed 2 x 81 accumulators (one read per 7.2 FMAs, LDS 1440 cycles per
// 128 pixels and chunk; lanes still walk consecutive float4, conflict-free).  Tile 16 x 16 (halo 24 x 24: 2.25
// gathered pixels per output pixel instead of 3), 256 threads = 2 directions x 8 pixel-pair rows x 16 columns, two
// blocks per CU (162 accumulators).  Same arithmetic in the same order as the one-pixel kernel: identical results.
namespace v2 {
constexpr int TH2 = 16, TW2 = 16;
constexpr int HH2 = TH2 + 2 * R, HW2 = TW2 + 2 * R;   // 24 x 24 halo
constexpr int HP2 = 32;                              // LDS row pitch in pixels
constexpr int NH2 = HH2 * HW2;                       // 576 halo pixels per map
constexpr int NG = (2 * NH2 + 255) / 256;            // gather rounds: 5 (the last one half full)
}  // namespace v2

template <bool POW2>
__global__ __launch_bounds__(256, 2) void warp_costvol_2px_kernel(const CorrLaunch p)
{
    using namespace v2;
    __shared__ __attribute__((aligned(16))) float4 nb[2][2][HH2 * HP2];   // [map][k4][pixel] 48 KB
    __shared__ float4 samp_w[2 * NH2];                                     // 18 KB
    __shared__ SampIdx samp_i[2 * NH2];                                    // 9 KB

    const int tid = threadIdx.x;
    const int tiles_x = (p.w + TW2 - 1) / TW2, tiles_y = (p.h + TH2 - 1) / TH2;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx_i * TW2, y0 = ty_i * TH2;

    const float *ref = p.ref + (size_t)b * p.img_stride;
    const float *nbr[2] = {p.nbr_fut + (size_t)b * p.img_stride, p.nbr_past + (size_t)b * p.img_stride};

    // ---- sampling records for the halo (once per block), flow loads issued together
    {
        float2 fl[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int i = min(tid + j * 256, 2 * NH2 - 1);
            const int map = i / NH2, hp = i - map * NH2;
            const int hy = hp / HW2, hx = hp - hy * HW2;
            const int y = y0 - R + hy, x = x0 - R + hx;
            const bool in = y >= 0 && y < p.h && x >= 0 && x < p.w;
            fl[j] = make_float2(0.f, 0.f);
            if (p.flow) fl[j] = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * p.h * p.w + (in ? (size_t)y * p.w + x : 0)) * 2);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int i = tid + j * 256;
            if (i < 2 * NH2) {
                const int map = i / NH2, hp = i - map * NH2;
                const int hy = hp / HW2, hx = hp - hy * HW2;
                const int y = y0 - R + hy, x = x0 - R + hx;
                float4 wgt = make_float4(0.f, 0.f, 0.f, 0.f);
                SampIdx si;
                si.idx = 0; si.flags = 0;
                if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
                    const float k = map == 0 ? p.k : -p.k;   // nn.MulConstant(20*(f-ref)/2^(l-2)), pwc.lua:404
                    const float u = fl[j].x * k, v = fl[j].y * k;
                    int xl, yt;
                    float wx, wy;
                    top_left(u + (float)x, p.w, xl, wx);
                    top_left(v + (float)y, p.h, yt, wy);
                    si.idx = (yt * p.w + xl) * p.pix_stride;
                    si.flags = ((xl + 1 <= p.w - 1) ? 1 : 0) | ((yt + 1 <= p.h - 1) ? 2 : 0);
                    wgt = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
                }
                si.flags |= (map * (2 * HH2 * HP2) + hy * HP2 + hx) << 2;   // LDS slot of this halo pixel (k4 = 0 plane)
                samp_w[i] = wgt;
                samp_i[i] = si;
            }
        }
    }

    float acc0[81], acc1[81];   // upper / lower pixel of the pair
#pragma unroll
    for (int i = 0; i < 81; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int pl = tid & 127, dir = tid >> 7;
    const int tyr = pl >> 4, lx = pl & 15;
    const int py0 = y0 + 2 * tyr, px = x0 + lx;
    const bool v0 = py0 < p.h && px < p.w, v1 = py0 + 1 < p.h && px < p.w;
    const float *refp0 = ref + (size_t)(v0 ? (py0 * p.w + px) : 0) * p.pix_stride;
    const float *refp1 = ref + (size_t)(v1 ? ((py0 + 1) * p.w + px) : 0) * p.pix_stride;
    // neighbour (row f, qx) of the pair: LDS pixel (2 tyr + R + f, lx + R - qx)
    const float4 *myn = &nb[dir][0][(2 * tyr + R) * HP2 + (lx + R)];

    __syncthreads();
    const int nchunk = p.C >> 3;
    for (int ch = 0; ch < nchunk; ++ch) {
        CORR_TV(ch, 0);
        const size_t coff = (size_t)ch * p.chunk_stride;
        // ---- gather + blend the warped halo chunk into LDS (as the one-pixel kernel: one thread fetches both float4 of
        // the chunk for its halo pixel, 8 loads in flight)
#pragma unroll 1
        for (int j = 0; j < NG; ++j) {
            const int q = tid + j * 256;
            if (q < 2 * NH2) {
                const int map = q >= NH2;
                const float4 wg = samp_w[q];
                const SampIdx si = samp_i[q];
                const float *src = nbr[map] + coff + si.idx;
                const int dx = (si.flags & 1) * p.pix_stride, dy = (si.flags & 2) ? p.w * p.pix_stride : 0;
                float4 t[8];
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4) {
                    t[k4 * 4 + 0] = *reinterpret_cast<const float4 *>(src + 4 * k4);
                    t[k4 * 4 + 1] = *reinterpret_cast<const float4 *>(src + dx + 4 * k4);
                    t[k4 * 4 + 2] = *reinterpret_cast<const float4 *>(src + dy + 4 * k4);
                    t[k4 * 4 + 3] = *reinterpret_cast<const float4 *>(src + dy + dx + 4 * k4);
                }
                float4 *dst = &nb[0][0][0] + (si.flags >> 2);
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4) {
                    const float4 tl = t[k4 * 4], tr = t[k4 * 4 + 1], bl = t[k4 * 4 + 2], br = t[k4 * 4 + 3];
                    float4 v;
                    v.x = fmaf(wg.w, br.x, fmaf(wg.z, bl.x, fmaf(wg.y, tr.x, wg.x * tl.x)));
                    v.y = fmaf(wg.w, br.y, fmaf(wg.z, bl.y, fmaf(wg.y, tr.y, wg.x * tl.y)));
                    v.z = fmaf(wg.w, br.z, fmaf(wg.z, bl.z, fmaf(wg.y, tr.z, wg.x * tl.z)));
                    v.w = fmaf(wg.w, br.w, fmaf(wg.z, bl.w, fmaf(wg.y, tr.w, wg.x * tl.w)));
                    dst[k4 * (HH2 * HP2)] = v;
                }
            }
        }
        const float4 ra[2] = {*reinterpret_cast<const float4 *>(refp0 + coff), *reinterpret_cast<const float4 *>(refp0 + coff + 4)};
        const float4 rb[2] = {*reinterpret_cast<const float4 *>(refp1 + coff), *reinterpret_cast<const float4 *>(refp1 + coff + 4)};
        CORR_TV(ch, 1);
        __syncthreads();
        CORR_TV(ch, 2);
        // ---- correlate: 20 steps = 2 k4 x 10 neighbour rows f = -4..5; row f is qy = -f of the upper pixel and
        // qy = 1 - f of the lower one (fwd volume: neighbour at (y - qy, x - qx), CostVolMulti.lua:76-87)
#pragma unroll
        for (int st = 0; st < 20; ++st) {
            const int k4 = st / 10, f = st - 10 * k4 - 4;
            float4 cur[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) cur[j] = myn[k4 * (HH2 * HP2) + f * HP2 - (j - 4)];   // qx = j - 4
            if (f <= 4) {
                const float4 r = ra[k4];
                const int qy = -f;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float a = acc0[j * 9 + qy + 4];
                    a = fmaf(r.x, cur[j].x, a);
                    a = fmaf(r.y, cur[j].y, a);
                    a = fmaf(r.z, cur[j].z, a);
                    a = fmaf(r.w, cur[j].w, a);
                    acc0[j * 9 + qy + 4] = a;
                }
            }
            if (f >= -3) {
                const float4 r = rb[k4];
                const int qy = 1 - f;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float a = acc1[j * 9 + qy + 4];
                    a = fmaf(r.x, cur[j].x, a);
                    a = fmaf(r.y, cur[j].y, a);
                    a = fmaf(r.z, cur[j].z, a);
                    a = fmaf(r.w, cur[j].w, a);
                    acc1[j * 9 + qy + 4] = a;
                }
            }
        }
        CORR_TV(ch, 3);
        __syncthreads();
        CORR_TV(ch, 4);
    }

    // ---- scale by 1/C (output:div(N), CostVolMulti.lua:100) and store the record slots of both pixels
    const float cf = (float)p.C, inv = 1.f / cf;
    auto store_px = [&](float *acc, bool valid, int py) {
        if (!valid) return;
#pragma unroll
        for (int c = 0; c < 81; ++c) acc[c] = POW2 ? acc[c] * inv : acc[c] / cf;
        const size_t pix = (size_t)py * p.w + px;
        float *o = p.out + (size_t)b * p.out_img_stride + pix * p.out_pix_stride;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            float4 lo, hi;
            if (dir == 0) {
                lo = make_float4(acc[8 * j], acc[8 * j + 1], acc[8 * j + 2], acc[8 * j + 3]);
                hi = make_float4(acc[8 * j + 4], acc[8 * j + 5], acc[8 * j + 6], acc[8 * j + 7]);
            } else {   // the bwd thread accumulated the mirrored window: acc[c'] holds bwd channel 80 - c'
                lo = make_float4(acc[80 - 8 * j], acc[79 - 8 * j], acc[78 - 8 * j], acc[77 - 8 * j]);
                hi = make_float4(acc[76 - 8 * j], acc[75 - 8 * j], acc[74 - 8 * j], acc[73 - 8 * j]);
            }
            float *oc = o + (size_t)(dir * 10 + j) * p.out_chunk_stride;
            *reinterpret_cast<float4 *>(oc) = lo;
            *reinterpret_cast<float4 *>(oc + 4) = hi;
        }
        float *ol = o + (size_t)20 * p.out_chunk_stride;   // last chunk: [fwd80, bwd80, u, v, ub, vb, 0, 0]
        if (dir == 0) {
            ol[0] = acc[80];
        } else {
            const size_t fp = ((size_t)b * p.h * p.w + pix) * 2;
            float2 f = make_float2(0.f, 0.f), fb = make_float2(0.f, 0.f);
            if (p.flow) f = *reinterpret_cast<const float2 *>(p.flow + fp);
            if (p.flow_b) fb = *reinterpret_cast<const float2 *>(p.flow_b + fp);
            ol[1] = acc[0];
            ol[2] = f.x; ol[3] = f.y;
            *reinterpret_cast<float4 *>(ol + 4) = make_float4(fb.x, fb.y, 0.f, 0.f);
        }
    };
    CORR_TV(nchunk, 0);
    store_px(acc0, v0, py0);
    store_px(acc1, v1, py0 + 1);
    CORR_TV(nchunk, 1);
}

// ======================================================================================================
// MFMA variant (round 2): the same function, bit for bit, restructured around what the measurements showed:
//  (1) round-1 ablation: gather, correlation (VALU issue + one ds_read_b128 per 4 v_fmac) and record stores each
//      took about a third of the kernel and ran strictly one after the other;
//  (2) the gather was bound by the L1's cache-LINE rate, not its byte rate: 4 taps x 2 half-chunks per halo pixel as
//      per-lane 16-byte loads touch ~18 lines per wave instruction, 1700 line visits per 8 x 16 tile and chunk for
//      ~150 distinct lines (11.6 B/clk/CU delivered of the 64 the L1 can do).
//
//  * The 9 x 9 banded correlation runs on v_mfma_f32_4x4x1_16b_f32: 16 independent 4 x 4 outer products per
//    instruction, K = 1 channel, exact fp32 (an fmaf chain in channel order -- the same chain the VALU version
//    builds, so the results are identical).  Block b of a wave = one 2 x 2 patch of reference pixels (B operand:
//    lane 4b + j = reference pixel j of the patch, i.e. every lane owns ONE reference pixel) against the 2 x 2
//    patch of warped neighbour pixels displaced by (2 dY, 2 dX), dY, dX in -2..2 (A operand: lane 4b + i =
//    neighbour pixel i of that patch): 25 accumulators of 4 VGPRs hold the 10 x 10 window of displacements
//    around the patch, of which each pixel uses 9 x 9 (81 % of the products; 1-D 4-pixel patches would use 75 %,
//    32 x 32 tiles 25 %).  One ds_read_b128 feeds 4 MFMAs = 1024 products, against 256 for the VALU form: LDS
//    reads drop 3x, VALU issue slots of the correlation drop to zero, and lane j ends up with all 81 values of
//    its own pixel (which register holds which displacement depends on the pixel's parity inside the patch:
//    two v_cndmask stages at the end instead of a transposition through LDS).  tools/mfma4x4_probe.hip: 8.2
//    cycles per instruction and wave, operand / result layout as used here.
//  * The unwarped source window of the tile (bounding box of all bilinear taps of its 16 x 24 halo, <= 22 x 32
//    pixels per neighbour map, found per tile by corr_window_kernel) is staged in LDS by LDS-DMA
//    (global_load_lds_dwordx4: one window row = one wave instruction = 1 KB of whole cache lines, no VGPRs, no
//    ds_write), and the four taps are blended from LDS.  A tile whose flow spreads the taps over more than the
//    window (motion boundaries) gathers that map straight from memory as round 1 did -- same arithmetic, same bits.
//  * Wave specialization inside a persistent block of 512 threads: waves 4..7 stage chunk n + 2, blend chunk n + 1
//    into one of two warped-halo slots (and copy the reference pixels' chunk) while waves 0..3 run the 200 MFMAs of
//    chunk n; one LDS-only barrier per chunk.  The consumers issue no loads at all, so nothing ever makes them
//    wait for their own record stores.
//  * Persistent over tiles (each XCD walks one contiguous band of tiles): the record stores of tile t drain while
//    the producers already stage tile t + 1 and the MFMAs of its first chunks run.
namespace mf {
constexpr int WR = 22, WC = 32;             // raw source window per map: rows x pixels (one row = 64 float4 = one LDS-DMA instruction)
constexpr int RAWMAP_F4 = WR * WC * 2;      // 1408 float4
constexpr int RAWREF_F4 = 2 * RAWMAP_F4;    // reference chunk of the tile: [k4][consumer wave half][lane]
constexpr int RAW_F4 = RAWREF_F4 + 2 * TH * TW;   // 3072 float4 = 48 KB per raw buffer
constexpr int HPITCH = HWD;                 // 24 = 8 (mod 16): the 2 x 2-patch ds_read_b128 pattern is conflict-free
constexpr int K4_F4 = NHALO + 4;            // 388 = 4 (mod 8): a lane pair (pixel, k4 = 0 | 1) writes distinct banks
constexpr int MAP_F4 = 2 * K4_F4;
constexpr int REF_F4 = 2 * MAP_F4;
constexpr int SLOT_F4 = REF_F4 + 2 * TH * TW;     // 1808 float4 per warped slot
constexpr int LDS_F4 = 2 * RAW_F4 + 2 * SLOT_F4;  // 9760 float4 = 156 160 B
constexpr int NTHREADS = 512;               // waves 0..3 consumers, 4..7 producers
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
}  // namespace mf

#define CORR_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")


// Window of one tile and map: origin (oy, ox) = smallest top-left tap row / column over the in-image halo pixels,
// rows = number of window rows the taps touch, fits = the taps stay inside WR x WC.  One wave per tile.
__global__ __launch_bounds__(64) void corr_window_kernel(const CorrLaunch p, int4 *winfo)
{
    const int lane = threadIdx.x;
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    int t = blockIdx.x;
    const int x0 = (t % tiles_x) * TW;
    t /= tiles_x;
    const int y0 = (t % tiles_y) * TH;
    const int b = t / tiles_y;
    for (int map = 0; map < 2; ++map) {
        int ymin = 1 << 30, ymax = -1, xmin = 1 << 30, xmax = -1;
        const float k = map == 0 ? p.k : -p.k;
        for (int hp = lane; hp < NHALO; hp += 64) {
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int y = y0 - R + hy, x = x0 - R + hx;
            if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
                float2 fl = make_float2(0.f, 0.f);
                if (p.flow) fl = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * p.h * p.w + (size_t)y * p.w + x) * 2);
                int xl, yt;
                float wx, wy;
                top_left(fl.x * k + (float)x, p.w, xl, wx);
                top_left(fl.y * k + (float)y, p.h, yt, wy);
                ymin = min(ymin, yt); ymax = max(ymax, yt);
                xmin = min(xmin, xl); xmax = max(xmax, xl);
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            ymin = min(ymin, __shfl_xor(ymin, o)); ymax = max(ymax, __shfl_xor(ymax, o));
            xmin = min(xmin, __shfl_xor(xmin, o)); xmax = max(xmax, __shfl_xor(xmax, o));
        }
        if (lane == 0) {
            const int rows = ymax - ymin + 2, cols = xmax - xmin + 2;   // + the bottom / right taps
            winfo[(size_t)blockIdx.x * 2 + map] = make_int4(ymin, xmin, min(rows, mf::WR), (rows <= mf::WR && cols <= mf::WC) ? 1 : 0);
        }
    }
}

// ABL: profiling builds only (-DB2F_CORR_ABLATE_VARIANTS, wrong results): 1 no staging / gather loads, 2 no MFMAs,
// 4 no stores.  Compile-time on purpose: a wave-uniform branch around the MFMA groups made the allocator spill in the loop.
template <bool POW2, int ABL>
__global__ __launch_bounds__(mf::NTHREADS, 2) void warp_costvol_mfma_kernel(const CorrLaunch p, const int4 *__restrict__ winfo)
{
    using mf::f32x4;
    // [raw 0][raw 1][warped slot 0][warped slot 1]
    __shared__ __attribute__((aligned(16))) f32x4 lds[mf::LDS_F4];
    f32x4 *const raw0 = lds, *const warped0 = lds + 2 * mf::RAW_F4;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave index in an SGPR: LDS-DMA bases must be scalar
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * p.B;
    // tiles of this block: XCD x (= blockIdx % 8, where the dispatcher puts the block) owns one contiguous band of
    // tiles, its blocks walk the band with stride (blocks on that XCD); placement only affects speed
    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;
    const int nbx = ((int)gridDim.x - xcd + 7) >> 3;
    const int tq = ntiles >> 3, trem = ntiles & 7;
    const int band0 = xcd * tq + (xcd < trem ? xcd : trem), bandn = tq + (xcd < trem ? 1 : 0);
    const int nchunk = p.C >> 3;
    const int my_tiles = rank < bandn ? (bandn - rank + nbx - 1) / nbx : 0;
    const int nitems = my_tiles * nchunk;   // item n = chunk n % nchunk of my tile n / nchunk
    if (my_tiles == 0) return;
    auto tile_id = [&](int ti) { return band0 + rank + ti * nbx; };
    auto tile_coords = [&](int ti, int &b, int &y0, int &x0) {
        int t = tile_id(ti);
        x0 = (t % tiles_x) * TW;
        t /= tiles_x;
        y0 = (t % tiles_y) * TH;
        b = t / tiles_y;
    };

    if (wave >= 4) {
        // ================= producers =================
        // Phase n (between barrier n and barrier n + 1): issue the LDS-DMA of item n + 2 into raw[n & 1] (free since
        // barrier n: item n was blended from it in phase n - 1), blend item n + 1 from raw[(n + 1) & 1] (its DMA was
        // issued in phase n - 1 and awaited by every wave before barrier n) into warped[(n + 1) & 1], await the DMA.
        const int pw = wave - 4, ptid = tid - 256;
        const int k4 = ptid & 1;
        // reference pixel whose half-chunk this thread stages: LDS index ptid = rk4 * 128 + half * 64 + consumer lane
        const int rk4 = ptid >> 7, rl = ptid & 63, rhalf = (ptid >> 6) & 1;
        const int rly = 4 * rhalf + 2 * (rl >> 5) + ((rl & 3) >> 1), rlx = 2 * ((rl >> 2) & 7) + (rl & 1);

        // ---- state of the blend cursor's tile: sampling records of this thread's six (halo pixel, k4) items.
        // Items s = 0..2 belong to the future map, 3..5 to the past map; halo pixel = (ptid >> 1) + 128 (s % 3).
        f32x4 wg[6];
        int woff[6];      // fitting map: float4 index of the top-left tap inside the raw window (incl. k4);
                          // other: byte offset of the tap in the feature plane (incl. k4)
        int dxo[6], dyo[6];
        int bfits[2] = {1, 1};
        int b_b = 0;      // image of the blend tile
        float2 flN[3];    // flow at this thread's three halo pixels of the NEXT tile (prefetched)
        // ---- state of the DMA cursor's tile
        int4 wi_dma[2], wi_prev[2];   // window info (per map) of the DMA cursor's tile and of the tile before it
        int4 wiN[2];      // prefetched for the tile after the DMA cursor's
        wi_dma[0] = wi_dma[1] = make_int4(0, 0, 0, 0);
        int d_b = 0, d_y0 = 0, d_x0 = 0, d_roff = 0, dma_entered = -1;
        int want_winfo = -1, want_flow = -1;   // prefetches to issue AFTER this phase's DMA (so a counted vmcnt can leave them in flight)

        auto load_winfo = [&](int ti) {
            if (ti < my_tiles) {
                wiN[0] = winfo[(size_t)tile_id(ti) * 2];
                wiN[1] = winfo[(size_t)tile_id(ti) * 2 + 1];
            }
        };
        auto load_flow = [&](int ti) {
            if (ti >= my_tiles) return;
            int b, y0, x0;
            tile_coords(ti, b, y0, x0);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int hp = (ptid >> 1) + 128 * s;
                const int hy = hp / HWD, hx = hp - hy * HWD;
                const int y = y0 - R + hy, x = x0 - R + hx;
                const bool in = y >= 0 && y < p.h && x >= 0 && x < p.w;
                flN[s] = make_float2(0.f, 0.f);
                if (p.flow) flN[s] = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * p.h * p.w + (in ? (size_t)y * p.w + x : 0)) * 2);
            }
        };
        // blend cursor enters tile ti: sampling records from the prefetched flow and the tile's window
        auto enter_blend_tile = [&](int ti, int dma_ti) {
            int y0, x0;
            tile_coords(ti, b_b, y0, x0);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int4 wi = dma_ti == ti ? wi_dma[m] : wi_prev[m];   // the DMA cursor is at most one tile ahead
                bfits[m] = wi.w;
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const int s = m * 3 + s3;
                    const int hp = (ptid >> 1) + 128 * s3;
                    const int hy = hp / HWD, hx = hp - hy * HWD;
                    const int y = y0 - R + hy, x = x0 - R + hx;
                    wg[s] = f32x4{0.f, 0.f, 0.f, 0.f};
                    woff[s] = wi.w ? k4 : 4 * k4 * 4;
                    dxo[s] = 0; dyo[s] = 0;
                    if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
                        const float k = m == 0 ? p.k : -p.k;   // nn.MulConstant(20*(f-ref)/2^(l-2)), pwc.lua:404
                        int xl, yt;
                        float wx, wy;
                        top_left(flN[s3].x * k + (float)x, p.w, xl, wx);
                        top_left(flN[s3].y * k + (float)y, p.h, yt, wy);
                        const int hasx = (xl + 1 <= p.w - 1), hasy = (yt + 1 <= p.h - 1);   // a neighbour outside the image has
                        if (wi.w) {                                                          // weight 0: fold it onto the clamped pixel
                            woff[s] = ((yt - wi.x) * mf::WC + (xl - wi.y)) * 2 + k4;
                            dxo[s] = hasx ? 2 : 0;
                            dyo[s] = hasy ? 2 * mf::WC : 0;
                        } else {
                            woff[s] = ((yt * p.w + xl) * p.pix_stride + 4 * k4) * 4;
                            dxo[s] = hasx ? p.pix_stride * 4 : 0;
                            dyo[s] = hasy ? p.w * p.pix_stride * 4 : 0;
                        }
                        wg[s] = f32x4{wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy)};
                    }
                }
            }
            want_flow = ti + 1;
        };
        auto enter_dma_tile = [&](int ti) {
            wi_prev[0] = wi_dma[0]; wi_prev[1] = wi_dma[1];
#pragma unroll
            for (int m = 0; m < 2; ++m)   // every lane loaded the same record: keep it in SGPRs (uniform loop bounds and branches)
                wi_dma[m] = make_int4(__builtin_amdgcn_readfirstlane(wiN[m].x), __builtin_amdgcn_readfirstlane(wiN[m].y),
                                      __builtin_amdgcn_readfirstlane(wiN[m].z), __builtin_amdgcn_readfirstlane(wiN[m].w));
            dma_entered = ti;
            want_winfo = ti + 1;
            tile_coords(ti, d_b, d_y0, d_x0);
            const int py = d_y0 + rly, px = d_x0 + rlx;
            d_roff = (py < p.h && px < p.w) ? (py * p.w + px) * p.pix_stride + 4 * rk4 : 0;
        };
        // LDS-DMA of item (tile ti, chunk ch) into raw buffer `rb`: window rows of both maps, one row per instruction,
        // rows spread over the four producer waves; plus the reference chunk (4 instructions, one per wave)
        auto issue_dma = [&](int ch, f32x4 *rb) {
            if (ABL & 1) return;
            const size_t plane = (size_t)d_b * p.img_stride + (size_t)ch * p.chunk_stride;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int4 wi = wi_dma[m];
                if (!wi.w) continue;
                const float *src = (m ? p.nbr_past : p.nbr_fut) + plane;
                int gx = wi.y + (lane >> 1);
                gx = gx > p.w - 1 ? p.w - 1 : gx;
                const int coff = gx * p.pix_stride + 4 * (lane & 1);
                for (int r = pw; r < wi.z; r += 4) {
                    int gy = wi.x + r;
                    gy = gy > p.h - 1 ? p.h - 1 : gy;
                    __builtin_amdgcn_global_load_lds((mf::glb_void *)(src + (size_t)gy * p.w * p.pix_stride + coff),
                                                     (mf::lds_void *)(rb + m * mf::RAWMAP_F4 + r * (2 * mf::WC)), 16, 0, 0);
                }
            }
            __builtin_amdgcn_global_load_lds((mf::glb_void *)(p.ref + plane + d_roff), (mf::lds_void *)(rb + mf::RAWREF_F4 + pw * 64), 16, 0, 0);
        };
        auto blend = [&](int ch, const f32x4 *rb, f32x4 *slot) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x4 tl[3], tr[3], bl[3], br[3];
                if (bfits[m]) {
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3) {
                        const f32x4 *q = rb + m * mf::RAWMAP_F4 + woff[m * 3 + s3];
                        const int dx = dxo[m * 3 + s3], dy = dyo[m * 3 + s3];
                        tl[s3] = q[0]; tr[s3] = q[dx]; bl[s3] = q[dy]; br[s3] = q[dy + dx];
                    }
                } else {
                    // this map of this tile does not fit the window: gather from memory (round-1 path)
                    const float *src = (m ? p.nbr_past : p.nbr_fut) + (size_t)b_b * p.img_stride + (size_t)ch * p.chunk_stride;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 0x7fffffff, 0x00020000);
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3) {
                        const int o = woff[m * 3 + s3], dx = dxo[m * 3 + s3], dy = dyo[m * 3 + s3];
                        if (ABL & 1) { tl[s3] = tr[s3] = bl[s3] = br[s3] = wg[m * 3 + s3]; continue; }
                        tl[s3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
                        tr[s3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o + dx, 0, 0));
                        bl[s3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o + dy, 0, 0));
                        br[s3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o + dy + dx, 0, 0));
                    }
                }
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const f32x4 w4 = wg[m * 3 + s3];
                    f32x4 v;
                    v.x = fmaf(w4.w, br[s3].x, fmaf(w4.z, bl[s3].x, fmaf(w4.y, tr[s3].x, w4.x * tl[s3].x)));
                    v.y = fmaf(w4.w, br[s3].y, fmaf(w4.z, bl[s3].y, fmaf(w4.y, tr[s3].y, w4.x * tl[s3].y)));
                    v.z = fmaf(w4.w, br[s3].z, fmaf(w4.z, bl[s3].z, fmaf(w4.y, tr[s3].z, w4.x * tl[s3].z)));
                    v.w = fmaf(w4.w, br[s3].w, fmaf(w4.z, bl[s3].w, fmaf(w4.y, tr[s3].w, w4.x * tl[s3].w)));
                    slot[m * mf::MAP_F4 + k4 * mf::K4_F4 + (ptid >> 1) + 128 * s3] = v;
                }
            }
            slot[mf::REF_F4 + ptid] = (ABL & 1) ? wg[0] : rb[mf::RAWREF_F4 + ptid];
        };

        // end of a phase: issue the pending prefetches (next tile's window record: 2 loads, next tile's flow: 3 loads)
        // behind the DMA, then wait for the DMA only -- vmcnt counts in order, the prefetches stay in flight
        auto end_phase = [&](int tn) {
            const int nw = (want_winfo >= 0 && want_winfo < my_tiles) ? 2 : 0, nf = (want_flow >= 0 && want_flow < my_tiles) ? 3 : 0;
            if (nw) load_winfo(want_winfo);
            if (nf) load_flow(want_flow);
            want_winfo = want_flow = -1;
            if (nw + nf == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (nw + nf == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (nw + nf == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            if (tn >= 0) CORR_T(tn, 3);
            CORR_LDS_BARRIER();
        };
        // ---- prologue: items 0 and 1
        load_winfo(0);
        load_flow(0);
        enter_dma_tile(0);
        int dti = 0, dch = 0;          // DMA cursor = item n + 2 in phase n
        issue_dma(0, raw0);
        end_phase(-1);                 // prologue barrier: raw[0] holds item 0
        int bti = 0, bch = 0;          // blend cursor = item n + 1 in phase n
        enter_blend_tile(0, dma_entered);
        if (++dch == nchunk) { dch = 0; ++dti; if (dti < my_tiles) enter_dma_tile(dti); }
        if (nitems > 1) issue_dma(dch, raw0 + mf::RAW_F4);
        blend(0, raw0, warped0);
        end_phase(-1);                 // barrier 0
        for (int n = 0; n + 1 < nitems; ++n) {
            CORR_T(n, 0);
            if (++dch == nchunk) { dch = 0; ++dti; if (dti < my_tiles) enter_dma_tile(dti); }
            if (n + 2 < nitems) issue_dma(dch, raw0 + (n & 1) * mf::RAW_F4);
            CORR_T(n, 1);
            if (++bch == nchunk) { bch = 0; ++bti; enter_blend_tile(bti, dma_entered); }
            blend(bch, raw0 + ((n + 1) & 1) * mf::RAW_F4, warped0 + ((n + 1) & 1) * mf::SLOT_F4);
            CORR_T(n, 2);
            end_phase(n);               // barrier n + 1
        }
        return;
    }

    // ================= consumers: banded correlation on the 4 x 4 x 1 MFMA, record stores =================
    const int dir = wave >> 1, half = wave & 1;
    const int blk = lane >> 2, jj = lane & 3;
    const int jx = jj & 1, jy = jj >> 1;
    const int ly = 4 * half + 2 * (blk >> 3) + jy, lx = 2 * (blk & 7) + jx;
    // A operand of displacement block (dY, dX): warped neighbour pixel at this lane's own position + (2 dY, 2 dX)
    const f32x4 *abase = warped0 + dir * mf::MAP_F4 + (ly + R) * mf::HPITCH + (lx + R);
    const f32x4 *rbase = warped0 + mf::REF_F4 + half * 64 + lane;
    const float cf = (float)p.C, inv = 1.f / cf;

    CORR_LDS_BARRIER();   // prologue barrier
    int item = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        int b, ty0, tx0;
        tile_coords(ti, b, ty0, tx0);
        const int py = ty0 + ly, px = tx0 + lx;
        const bool pvalid = py < p.h && px < p.w;

        // the flows copied into the record's last chunk: loaded now (bwd waves), a whole tile before their use -- behind the
        // record stores they would make the wave wait for all of those (vmcnt counts in order)
        float2 fcopy = make_float2(0.f, 0.f), fbcopy = make_float2(0.f, 0.f);
        if (dir == 1 && pvalid) {
            const size_t fp = ((size_t)b * p.h * p.w + (size_t)py * p.w + px) * 2;
            if (p.flow) fcopy = *reinterpret_cast<const float2 *>(p.flow + fp);
            if (p.flow_b) fbcopy = *reinterpret_cast<const float2 *>(p.flow_b + fp);
        }

        f32x4 acc[5][5];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int ch = 0; ch < nchunk; ++ch, ++item) {
            CORR_LDS_BARRIER();   // barrier `item`: the item is in its slot
            CORR_T(item, 0);
            const f32x4 *ab = abase + (item & 1) * mf::SLOT_F4;
            const f32x4 r0 = rbase[(item & 1) * mf::SLOT_F4], r1 = rbase[(item & 1) * mf::SLOT_F4 + 128];
            // 10 steps = 2 k4 x 5 dY rows; the 5 A operands of step s + 1 are read under the 20 MFMAs of step s
            f32x4 va[5], vb[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) va[c] = ab[(-4) * mf::HPITCH + 2 * (c - 2)];
#pragma unroll
            for (int st = 0; st < 10; ++st) {
                const int k4 = st / 5, a = st - 5 * k4;
                const f32x4 r = k4 ? r1 : r0;
                f32x4 *cur = (st & 1) ? vb : va, *nxt = (st & 1) ? va : vb;
                if (st + 1 < 10) {
                    const int k4n = (st + 1) / 5, an = (st + 1) - 5 * k4n;
#pragma unroll
                    for (int c = 0; c < 5; ++c) nxt[c] = ab[k4n * mf::K4_F4 + 2 * (an - 2) * mf::HPITCH + 2 * (c - 2)];
                }
                if (!(ABL & 2)) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int c = 0; c < 5; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(cur[c][k], r[k], acc[a][c], 0, 0, 0);
                } else {
#pragma unroll
                    for (int c = 0; c < 5; ++c) acc[a][c].x += cur[c].x + r.x;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            CORR_T(item, 1);
        }

        // ---- output stage.  T[ty][tx], ty, tx in -4..5: product with the neighbour at patch origin + (ty, tx), held in
        // acc[ty >> 1 + 2][tx >> 1 + 2][(ty & 1) * 2 + (tx & 1)].  This lane's pixel sits at (jy, jx) inside the patch,
        // so its displacement (qy, qx) (fwd volume: neighbour at (y - qy, x - qx), CostVolMulti.lua:76-87) is
        // T[jy - qy][jx - qx]: select by the lane's parity, x first, then y.  The bwd wave accumulated the mirrored
        // window: bwd channel c is the fwd-style value of (-qx, -qy).  Channels leave in record order, one qx column
        // (9 channels) at a time, so that only a column and the pending part of a chunk are live beside the accumulators.
        const bool st_on = pvalid && !((ABL & 4) && acc[0][0][0] != 12345.678f);
        const size_t pix = pvalid ? (size_t)py * p.w + px : 0;
        float *o = p.out + (size_t)b * p.out_img_stride + pix * p.out_pix_stride + (size_t)(dir * 10) * p.out_chunk_stride;
        auto out_stage = [&](auto dir_c) {
            constexpr int DIR = decltype(dir_c)::value;
            float pend[17];
            int np = 0, chunk = 0;
#pragma unroll
            for (int qxo = -4; qxo <= 4; ++qxo) {
                const int qx = DIR ? -qxo : qxo;
                const int t0 = -qx, t1 = 1 - qx;   // tx for jx = 0 / 1
                float ucol[10];                    // [ty + 4]
#pragma unroll
                for (int ty = -4; ty <= 5; ++ty) {
                    const float v0 = acc[(ty + 4) >> 1][(t0 + 4) >> 1][((ty + 4) & 1) * 2 + ((t0 + 4) & 1)];
                    const float v1 = acc[(ty + 4) >> 1][(t1 + 4) >> 1][((ty + 4) & 1) * 2 + ((t1 + 4) & 1)];
                    ucol[ty + 4] = jx ? v1 : v0;
                }
#pragma unroll
                for (int qyo = -4; qyo <= 4; ++qyo) {
                    const int qy = DIR ? -qyo : qyo;
                    const float v = jy ? ucol[1 - qy + 4] : ucol[-qy + 4];
                    pend[np++] = POW2 ? v * inv : v / cf;   // output:div(N), CostVolMulti.lua:100
                }
#pragma unroll
                for (int rep = 0; rep < 2; ++rep) {
                    if (np >= 8 && chunk < 10) {
                        if (st_on) {
                            float *oc = o + (size_t)chunk * p.out_chunk_stride;
                            *reinterpret_cast<float4 *>(oc) = make_float4(pend[0], pend[1], pend[2], pend[3]);
                            *reinterpret_cast<float4 *>(oc + 4) = make_float4(pend[4], pend[5], pend[6], pend[7]);
                        }
#pragma unroll
                        for (int i = 8; i < 17; ++i) pend[i - 8] = pend[i];
                        np -= 8;
                        ++chunk;
                    }
                }
                if (qxo == 0) CORR_T(item - 1, 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            // 81 = 10 chunks + 1: channel 80 of this direction goes to the last chunk [fwd80, bwd80, u, v, ub, vb, 0, 0]
            if (st_on) {
                float *ol = p.out + (size_t)b * p.out_img_stride + pix * p.out_pix_stride + (size_t)20 * p.out_chunk_stride;
                if (DIR == 0) {
                    ol[0] = pend[0];
                } else {
                    ol[1] = pend[0];
                    ol[2] = fcopy.x; ol[3] = fcopy.y;
                    *reinterpret_cast<float4 *>(ol + 4) = make_float4(fbcopy.x, fbcopy.y, 0.f, 0.f);
                }
            }
        };
        if (dir == 0) out_stage(std::integral_constant<int, 0>());
        else out_stage(std::integral_constant<int, 1>());
        CORR_T(item - 1, 2);
    }
}

static void corr_trace_dump(const CorrLaunch &p, hipStream_t s)
{
#ifdef B2F_CORR_TRACE
    static int traced = 0;
    if (p.h * p.w >= 256 * 480 && traced++ == 3) {
        (void)hipStreamSynchronize(s);
        std::vector<long long> t(8 * 64 * 4);
        (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_corr_trace), t.size() * 8);
        const long long t0 = t[(0 * 64 + 0) * 4];
        fprintf(stderr, "corr trace (block 0; cycles since consumer wave 0 started item 0)\n item | cons0: start mfma_done out_half out_done | prod4: start dma_issued blended waited | prod7: start waited\n");
        for (int i = 0; i < 24; ++i)
            fprintf(stderr, " %3d | %7lld %7lld %7lld %7lld | %7lld %7lld %7lld %7lld | %7lld %7lld\n", i, t[i * 4] - t0, t[i * 4 + 1] - t0, t[i * 4 + 3] - t0, t[i * 4 + 2] - t0,
                    t[(4 * 64 + i) * 4] - t0, t[(4 * 64 + i) * 4 + 1] - t0, t[(4 * 64 + i) * 4 + 2] - t0, t[(4 * 64 + i) * 4 + 3] - t0,
                    t[(7 * 64 + i) * 4] - t0, t[(7 * 64 + i) * 4 + 3] - t0);
    }
#else
    (void)p; (void)s;
#endif
}

size_t corr_winfo_bytes(int B, int h, int w)
{
    return (size_t)B * ((h + TH - 1) / TH) * ((w + TW - 1) / TW) * 2 * sizeof(int4);
}

hipError_t launch_warp_costvol(const CorrLaunch &p_in, hipStream_t s)
{
    if (p_in.C % 8 != 0 || p_in.pix_stride % 4 != 0 || p_in.chunk_stride % 4 != 0 || p_in.out_pix_stride % 4 != 0 ||
        p_in.out_chunk_stride % 4 != 0)
        return hipErrorInvalidValue;
    CorrLaunch p = p_in;
    const int ablate = p.ablate;
    (void)ablate;
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    dim3 grid((unsigned)(tiles_x * tiles_y * p.B));
    if ((p.variant == 2 || p.variant < 0) && p.winfo) {
        // persistent MFMA variant: one 512-thread block per CU (at most one per tile), after the per-tile window pass
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
            if (ncu <= 0) ncu = 256;
        }
        const unsigned nblk = std::min<unsigned>((unsigned)ncu, grid.x);
        int4 *wi = reinterpret_cast<int4 *>(p.winfo);
        hipLaunchKernelGGL(corr_window_kernel, grid, dim3(64), 0, s, p, wi);
#ifdef B2F_CORR_ABLATE_VARIANTS
        if (pow2 && (ablate & 7)) {
            switch (ablate & 7) {
#define B2F_ABL_CASE(n_) case n_: hipLaunchKernelGGL((warp_costvol_mfma_kernel<true, n_>), dim3(nblk), dim3(mf::NTHREADS), 0, s, p, wi); break;
                B2F_ABL_CASE(1) B2F_ABL_CASE(2) B2F_ABL_CASE(3) B2F_ABL_CASE(4) B2F_ABL_CASE(5) B2F_ABL_CASE(6) B2F_ABL_CASE(7)
#undef B2F_ABL_CASE
            }
            corr_trace_dump(p, s);
            return hipGetLastError();
        }
#endif
        if (pow2) hipLaunchKernelGGL((warp_costvol_mfma_kernel<true, 0>), dim3(nblk), dim3(mf::NTHREADS), 0, s, p, wi);
        else hipLaunchKernelGGL((warp_costvol_mfma_kernel<false, 0>), dim3(nblk), dim3(mf::NTHREADS), 0, s, p, wi);
        corr_trace_dump(p, s);
        return hipGetLastError();
    }
    if (p.variant == 3) {
        const int t2x = (p.w + v2::TW2 - 1) / v2::TW2, t2y = (p.h + v2::TH2 - 1) / v2::TH2;
        const dim3 g2((unsigned)(t2x * t2y * p.B));
        if (pow2) hipLaunchKernelGGL((warp_costvol_2px_kernel<true>), g2, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((warp_costvol_2px_kernel<false>), g2, dim3(256), 0, s, p);
#ifdef B2F_CORR_TRACE
        static int traced2 = 0;
        if (p.h * p.w >= 256 * 480 && traced2++ == 3) {
            (void)hipStreamSynchronize(s);
            std::vector<long long> t(8 * 64 * 4);
            (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_corr_trace), t.size() * 8);
            fprintf(stderr, "2px kernel trace, block 0 wave 0: chunk | gather_start blend_done barrier1 fma_done barrier2 (cycles since chunk 0 start)\n");
            for (int c = 0; c <= p.C / 8; ++c)
                fprintf(stderr, " %2d | %7lld %7lld %7lld %7lld %7lld\n", c, t[c * 8] - t[0], t[c * 8 + 1] - t[0], t[c * 8 + 2] - t[0], t[c * 8 + 3] - t[0], t[c * 8 + 4] - t[0]);
        }
#endif
        return hipGetLastError();
    }
    // round-1 VALU kernels (kept for A/B runs): at most one round of two blocks per CU -> the latency variant
    const bool lat = p.variant != 0 ? (p.variant == 1) : false;
    if (lat) {
        if (pow2) hipLaunchKernelGGL((warp_costvol_kernel<true, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((warp_costvol_kernel<false, true>), grid, dim3(256), 0, s, p);
    } else {
        if (pow2) hipLaunchKernelGGL((warp_costvol_kernel<true, false>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((warp_costvol_kernel<false, false>), grid, dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

// Generic single-direction cost volume for windows other than the shipped 9x9
// (createModelMulti(nil) uses win 5, pwc.lua:88).  One thread per output element;
// not on the hot path.  NHWC in, B x h x w x win*win out.
__global__ void costvol_generic_kernel(const float *ref, const float *frm, int B, int C, int h, int w,
                                       int win, int fwd, float *out)
{
    const size_t total = (size_t)B * h * w * win * win;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = (win - 1) / 2;
    const int c = (int)(i % (win * win));
    size_t pix = i / (win * win);
    const int x = (int)(pix % w);
    pix /= w;
    const int y = (int)(pix % h);
    const int b = (int)(pix / h);
    int qx = c / win - n, qy = c % win - n;
    if (!fwd) { qx = -qx; qy = -qy; }
    const int yy = y - qy, xx = x - qx;
    float acc = 0.f;
    if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
        const float *r = ref + ((size_t)(b * h + y) * w + x) * C;
        const float *g = frm + ((size_t)(b * h + yy) * w + xx) * C;
        for (int k = 0; k < C; ++k) acc = fmaf(r[k], g[k], acc);
    }
    out[i] = acc / (float)C;
}

hipError_t launch_costvol_generic(const float *ref, const float *frm, int B, int C, int h, int w,
                                  int win, int fwd, float *out, hipStream_t s)
{
    const size_t total = (size_t)B * h * w * win * win;
    hipLaunchKernelGGL(costvol_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       ref, frm, B, C, h, w, win, fwd, out);
    return hipGetLastError();
}

}  // namespace b2f
