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

#include <cstdlib>

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
                bhwd_top_left(u + (float)x, p.w, xl, wx);
                bhwd_top_left(v + (float)y, p.h, yt, wy);
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
constexpr int PL2 = HH2 * HP2 + 4;                   // float4 per (map, k4) plane: 772 = 4 (mod 8), so that the lane pair
                                                     // (pixel, k4 = 0 | 1) of the gather writes distinct LDS banks
}  // namespace v2

// NDIR = 2: 256 threads compute both directions of a tile (one gather pass covers both neighbour maps' halos).
// NDIR = 1: 128 threads compute ONE direction (block parity): four independent blocks per CU instead of two, whose
//           memory phases (gather, record stores -- HBM-bound) and compute phases overlap better.
template <bool POW2, int NDIR>
__global__ __launch_bounds__(128 * NDIR, 2) void warp_costvol_2px_kernel(const CorrLaunch p)
{
    using namespace v2;
    constexpr int NTHR = 128 * NDIR;
    constexpr int NPX = NDIR * NH2;                    // halo pixels this block gathers
    constexpr int NG = (NPX + NTHR - 1) / NTHR;        // sampling-record rounds: 5 (the last one half full)
    constexpr int NIT = NPX * 2 / NTHR;                // gather items (halo pixel, k4) per thread and chunk: 9
    __shared__ __attribute__((aligned(16))) float4 nb[NDIR][2][PL2];      // [map][k4][pixel] 24 KB per map
    __shared__ float4 samp_w[NPX];                                         // 9 KB per map
    __shared__ SampIdx samp_i[NPX];                                        // 4.5 KB per map
    __shared__ float pf_sink[NDIR * 192];                                  // landing zone of the L2 prefetch (never read)

    const int tid = threadIdx.x;
    const int tiles_x = (p.w + TW2 - 1) / TW2, tiles_y = (p.h + TH2 - 1) / TH2;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bdir = NDIR == 1 ? (bid & 1) : 0;        // NDIR = 1: direction of this block (0 fwd / future map, 1 bwd / past map)
    if (NDIR == 1) bid >>= 1;
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
            const int i = min(tid + j * NTHR, NPX - 1);
            const int hp = i % NH2;
            const int hy = hp / HW2, hx = hp - hy * HW2;
            const int y = y0 - R + hy, x = x0 - R + hx;
            const bool in = y >= 0 && y < p.h && x >= 0 && x < p.w;
            fl[j] = make_float2(0.f, 0.f);
            if (p.flow) fl[j] = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * p.h * p.w + (in ? (size_t)y * p.w + x : 0)) * 2);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int i = tid + j * NTHR;
            if (i < NPX) {
                const int lmap = i / NH2, hp = i - lmap * NH2;
                const int map = NDIR == 1 ? bdir : lmap;
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
                    bhwd_top_left(u + (float)x, p.w, xl, wx);
                    bhwd_top_left(v + (float)y, p.h, yt, wy);
                    si.idx = (yt * p.w + xl) * p.pix_stride;
                    si.flags = ((xl + 1 <= p.w - 1) ? 1 : 0) | ((yt + 1 <= p.h - 1) ? 2 : 0);
                    wgt = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
                }
                si.flags |= (lmap * (2 * PL2) + hy * HP2 + hx) << 2;   // LDS slot of this halo pixel (k4 = 0 plane)
                samp_w[i] = wgt;
                samp_i[i] = si;
            }
        }
    }

    float acc0[81], acc1[81];   // upper / lower pixel of the pair
#pragma unroll
    for (int i = 0; i < 81; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int pl = tid & 127, dir = NDIR == 1 ? bdir : (tid >> 7);
    const int tyr = pl >> 4, lx = pl & 15;
    const int py0 = y0 + 2 * tyr, px = x0 + lx;
    const bool v0 = py0 < p.h && px < p.w, v1 = py0 + 1 < p.h && px < p.w;
    const float *refp0 = ref + (size_t)(v0 ? (py0 * p.w + px) : 0) * p.pix_stride;
    const float *refp1 = ref + (size_t)(v1 ? ((py0 + 1) * p.w + px) : 0) * p.pix_stride;
    // neighbour (row f, qx) of the pair: LDS pixel (2 tyr + R + f, lx + R - qx)
    const float4 *myn = &nb[NDIR == 1 ? 0 : dir][0][(2 * tyr + R) * HP2 + (lx + R)];

    __syncthreads();
    const int nchunk = p.C >> 3;
    for (int ch = 0; ch < nchunk; ++ch) {
        const size_t coff = (size_t)ch * p.chunk_stride;
        // ---- gather + blend the warped halo chunk into LDS.  Item = (halo pixel, k4): lanes 2i and 2i + 1 fetch the two
        // 16-byte halves of the same pixel's chunk, so a wave's tap load covers 32 pixels x 32 contiguous bytes -- whole
        // cache lines (8 - 10 per instruction instead of 16 - 20 when a lane fetched both halves 32 bytes apart from
        // its neighbour's).  Three items = 12 loads in flight.
        const int gk4 = tid & 1;
#pragma unroll 1
        for (int j0 = 0; j0 < NIT; j0 += 3) {
            float4 t[3][4], wg[3];
            int slot[3];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int q = (tid >> 1) + (NTHR / 2) * (j0 + jj);
                const int map = NDIR == 1 ? bdir : (q >= NH2);
                wg[jj] = samp_w[q];
                const SampIdx si = samp_i[q];
                const float *src = nbr[map] + coff + si.idx + 4 * gk4;
                const int dx = (si.flags & 1) * p.pix_stride, dy = (si.flags & 2) ? p.w * p.pix_stride : 0;
                slot[jj] = (si.flags >> 2) + gk4 * PL2;
                t[jj][0] = *reinterpret_cast<const float4 *>(src);
                t[jj][1] = *reinterpret_cast<const float4 *>(src + dx);
                t[jj][2] = *reinterpret_cast<const float4 *>(src + dy);
                t[jj][3] = *reinterpret_cast<const float4 *>(src + dy + dx);
            }
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const float4 w4 = wg[jj], tl = t[jj][0], tr = t[jj][1], bl = t[jj][2], br = t[jj][3];
                float4 v;
                v.x = fmaf(w4.w, br.x, fmaf(w4.z, bl.x, fmaf(w4.y, tr.x, w4.x * tl.x)));
                v.y = fmaf(w4.w, br.y, fmaf(w4.z, bl.y, fmaf(w4.y, tr.y, w4.x * tl.y)));
                v.z = fmaf(w4.w, br.z, fmaf(w4.z, bl.z, fmaf(w4.y, tr.z, w4.x * tl.z)));
                v.w = fmaf(w4.w, br.w, fmaf(w4.z, bl.w, fmaf(w4.y, tr.w, w4.x * tl.w)));
                (&nb[0][0][0])[slot[jj]] = v;
            }
        }
        const float4 ra[2] = {*reinterpret_cast<const float4 *>(refp0 + coff), *reinterpret_cast<const float4 *>(refp0 + coff + 4)};
        const float4 rb[2] = {*reinterpret_cast<const float4 *>(refp1 + coff), *reinterpret_cast<const float4 *>(refp1 + coff + 4)};
        __syncthreads();
        // ---- L2 prefetch of the NEXT chunk's source window, issued under this chunk's FMAs: the kernel is HBM-bound for
        // 65 % of its time and the FMA phase holds the registers (162 accumulators), so the gather cannot be software-
        // pipelined through VGPRs.  One LDS-DMA dword per cache line instead -- no registers, no wait: lane i of the first
        // wave(s) touches line i % 7 of halo row i / 7 (7 lines cover a row's 24 x 32 bytes wherever it starts; row 24 = the
        // bottom taps of the last row) at the address its sampling record points to; the data lands in pf_sink and is
        // never read.  The gather of the next chunk then finds its lines in L2.
        // (the reference pixels' loads are awaited first: vmcnt counts in order, and a wait placed after the DMA would
        // drain it before the first FMA)
        asm volatile("" :: "v"(ra[0].x), "v"(ra[1].x), "v"(rb[0].x), "v"(rb[1].x));
        if (ch + 1 < nchunk && (tid >> 6) < NDIR) {
            const int lmap = tid >> 6;
            const int map = NDIR == 1 ? bdir : lmap;
            const float *nbase = nbr[map] + coff + p.chunk_stride;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int i = (tid & 63) + 64 * it;
                const int r = min(i / 7, HH2), sg = i - (i / 7) * 7;
                const SampIdx si = samp_i[lmap * NH2 + min(r, HH2 - 1) * HW2 + min(4 * sg, HW2 - 1)];
                const float *a = nbase + si.idx + (r == HH2 ? p.w * p.pix_stride : 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)a,
                                                 (__attribute__((address_space(3))) void *)(pf_sink + lmap * 192 + it * 64), 4, 0, 0);
            }
        }
        // ---- correlate: 20 steps = 2 k4 x 10 neighbour rows f = -4..5; row f is qy = -f of the upper pixel and
        // qy = 1 - f of the lower one (fwd volume: neighbour at (y - qy, x - qx), CostVolMulti.lua:76-87)
#pragma unroll
        for (int st = 0; st < 20; ++st) {
            const int k4 = st / 10, f = st - 10 * k4 - 4;
            float4 cur[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) cur[j] = myn[k4 * PL2 + f * HP2 - (j - 4)];   // qx = j - 4
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
        __syncthreads();
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
    store_px(acc0, v0, py0);
    store_px(acc1, v1, py0 + 1);
}


#if B2F_EXPERIMENTS   // variant 8 (four pixels x three qx columns per thread): measured slower than variant 3, profiles/r05_corr_notes.txt
// ======================================================================================================
// Four-pixel variant (round 5, variant 8).  What the two-pixel kernel is short of is issue slots and LDS bandwidth, both for the
// same reason: 2 x 81 accumulators per thread leave two waves per SIMD (one VALU instruction every ~2.75 cycles where the pipe takes
// one every 1.67: tools/fma_rate.hip) and one ds_read_b128 per 7.2 FMAs (levels 3 + 4 read 34 GB of LDS per step: 0.5 ms of the
// 0.97 at 128 B/clk).  Here a thread owns FOUR vertically adjacent pixels but only THREE of the nine qx columns -- wave g of a
// 192-thread block takes qx = 3g - 4 .. 3g - 2 for all 64 pixel quads of the 16 x 16 tile: 4 x 27 = 108 accumulators (three waves
// per SIMD, four blocks = twelve waves per CU), a neighbour row serves up to four pixels (12 rows x 3 reads feed 432 FMAs per channel
// quad: one read per 12 FMAs).  Halo, sampling records, gather, L2 prefetch and the order of every accumulator's operations are the
// two-pixel kernel's: identical results.  One direction per block.  A thread's 27 channels are not a whole number of 8-float
// records: the records shared by two column groups (3 and 6 of a direction's ten) are written in pieces by both.
// MEASURED (batch 16, levels 3 / 4 / 5): 0.826 / 0.337 / 0.179 ms against 0.695 / 0.301 / 0.121 of the two-pixel kernel -- the premise
// was wrong.  Ablations at level 3 (p.ablate: 1 no tap loads, 2 no FMAs, 4 no stores, 16 no reference loads, 32 no L2 prefetch):
// stores 0.32 ms, reference loads 0.18 (read by three waves here), tap loads 0.12, FMAs 0.06, skeleton 0.17 -- the parts ADD UP to
// the total, and starting the co-resident blocks out of phase changes nothing: one shared resource, the CU's vector-memory path
// (loads and stores together), not issue slots or LDS bandwidth.  Experiments build only.
template <bool POW2>
__global__ __launch_bounds__(192, 3) void warp_costvol_4px_kernel(const CorrLaunch p)
{
    using namespace v2;
    constexpr int NTHR = 192;
    constexpr int NG = NH2 / NTHR;                     // sampling-record rounds: 3 (576 halo pixels)
    constexpr int NIT = NH2 * 2 / NTHR;                // gather items (halo pixel, k4) per thread and chunk: 6
    static_assert(NG * NTHR == NH2 && NIT * NTHR == 2 * NH2, "halo must divide evenly");
    __shared__ __attribute__((aligned(16))) float4 nb[2][PL2];             // [k4][pixel] 24 KB
    __shared__ float4 samp_w[NH2];                                         // 9 KB
    __shared__ SampIdx samp_i[NH2];                                        // 4.5 KB
    __shared__ float pf_sink[192];                                         // landing zone of the L2 prefetch (never read)

    const int tid = threadIdx.x;
    const int tiles_x = (p.w + TW2 - 1) / TW2, tiles_y = (p.h + TH2 - 1) / TH2;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int dir = bid & 1;                           // 0 fwd / future map, 1 bwd / past map
    bid >>= 1;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx_i * TW2, y0 = ty_i * TH2;

    const float *ref = p.ref + (size_t)b * p.img_stride;
    const float *nbr = (dir == 0 ? p.nbr_fut : p.nbr_past) + (size_t)b * p.img_stride;

    // ---- sampling records for the halo (once per block), flow loads issued together
    {
        float2 fl[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int hp = tid + j * NTHR;
            const int hy = hp / HW2, hx = hp - hy * HW2;
            const int y = y0 - R + hy, x = x0 - R + hx;
            const bool in = y >= 0 && y < p.h && x >= 0 && x < p.w;
            fl[j] = make_float2(0.f, 0.f);
            if (p.flow) fl[j] = *reinterpret_cast<const float2 *>(p.flow + ((size_t)b * p.h * p.w + (in ? (size_t)y * p.w + x : 0)) * 2);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int hp = tid + j * NTHR;
            const int hy = hp / HW2, hx = hp - hy * HW2;
            const int y = y0 - R + hy, x = x0 - R + hx;
            float4 wgt = make_float4(0.f, 0.f, 0.f, 0.f);
            SampIdx si;
            si.idx = 0; si.flags = 0;
            if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
                const float k = dir == 0 ? p.k : -p.k;   // nn.MulConstant(20*(f-ref)/2^(l-2)), pwc.lua:404
                const float u = fl[j].x * k, v = fl[j].y * k;
                int xl, yt;
                float wx, wy;
                bhwd_top_left(u + (float)x, p.w, xl, wx);
                bhwd_top_left(v + (float)y, p.h, yt, wy);
                si.idx = (yt * p.w + xl) * p.pix_stride;
                si.flags = ((xl + 1 <= p.w - 1) ? 1 : 0) | ((yt + 1 <= p.h - 1) ? 2 : 0);
                wgt = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
            }
            si.flags |= (hy * HP2 + hx) << 2;   // LDS slot of this halo pixel (k4 = 0 plane)
            samp_w[hp] = wgt;
            samp_i[hp] = si;
        }
    }

    float acc[4][27];   // [pixel of the quad][(qx - (3 grp - 4)) * 9 + qy + 4]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 27; ++i) acc[r][i] = 0.f;

    const int grp = __builtin_amdgcn_readfirstlane(tid >> 6);   // qx column group of this wave
    const int lane = tid & 63;
    const int tyq = lane >> 4, lx = lane & 15;
    const int py0 = y0 + 4 * tyq, px = x0 + lx;
    const float *refp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) refp[r] = ref + (size_t)((py0 + r < p.h && px < p.w) ? ((py0 + r) * p.w + px) : 0) * p.pix_stride;
    // neighbour (row f, column jj of the group) of the quad: LDS pixel (4 tyq + R + f, lx + R - qx), qx = 3 grp - 4 + jj
    const float4 *myn = &nb[0][(4 * tyq + R) * HP2 + (lx + R + 4 - 3 * grp)];

    __syncthreads();
    const int nchunk = p.C >> 3;
    for (int ch = 0; ch < nchunk; ++ch) {
        const size_t coff = (size_t)ch * p.chunk_stride;
        // ---- gather + blend the warped halo chunk into LDS (the two-pixel kernel's: lanes 2i / 2i + 1 fetch the two 16-byte
        // halves of a pixel's chunk); two items = 8 loads in flight per thread, 1 536 per block as there
        const int gk4 = tid & 1;
#pragma unroll 1
        for (int j0 = 0; j0 < NIT; j0 += 2) {
            float4 t[2][4], wg[2];
            int slot[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int q = (tid >> 1) + (NTHR / 2) * (j0 + jj);
                wg[jj] = samp_w[q];
                const SampIdx si = samp_i[q];
                const float *src = nbr + coff + si.idx + 4 * gk4;
                const int dx = (si.flags & 1) * p.pix_stride, dy = (si.flags & 2) ? p.w * p.pix_stride : 0;
                slot[jj] = (si.flags >> 2) + gk4 * PL2;
                if (p.ablate & 1) { t[jj][0] = t[jj][1] = t[jj][2] = t[jj][3] = wg[jj]; continue; }
                t[jj][0] = *reinterpret_cast<const float4 *>(src);
                t[jj][1] = *reinterpret_cast<const float4 *>(src + dx);
                t[jj][2] = *reinterpret_cast<const float4 *>(src + dy);
                t[jj][3] = *reinterpret_cast<const float4 *>(src + dy + dx);
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const float4 w4 = wg[jj], tl = t[jj][0], tr = t[jj][1], bl = t[jj][2], br = t[jj][3];
                float4 v;
                v.x = fmaf(w4.w, br.x, fmaf(w4.z, bl.x, fmaf(w4.y, tr.x, w4.x * tl.x)));
                v.y = fmaf(w4.w, br.y, fmaf(w4.z, bl.y, fmaf(w4.y, tr.y, w4.x * tl.y)));
                v.z = fmaf(w4.w, br.z, fmaf(w4.z, bl.z, fmaf(w4.y, tr.z, w4.x * tl.z)));
                v.w = fmaf(w4.w, br.w, fmaf(w4.z, bl.w, fmaf(w4.y, tr.w, w4.x * tl.w)));
                (&nb[0][0])[slot[jj]] = v;
            }
        }
        float4 rf[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (p.ablate & 16) { rf[r][0] = rf[r][1] = make_float4((float)ch, 1.f, 2.f, (float)r); continue; }
            rf[r][0] = *reinterpret_cast<const float4 *>(refp[r] + coff);
            rf[r][1] = *reinterpret_cast<const float4 *>(refp[r] + coff + 4);
        }
        __syncthreads();
        // ---- L2 prefetch of the NEXT chunk's source window under this chunk's FMAs (see the two-pixel kernel)
        asm volatile("" :: "v"(rf[0][0].x), "v"(rf[1][0].x), "v"(rf[2][0].x), "v"(rf[3][0].x), "v"(rf[0][1].x), "v"(rf[1][1].x), "v"(rf[2][1].x), "v"(rf[3][1].x));
        if (ch + 1 < nchunk && tid < 64 && !(p.ablate & 32)) {
            const float *nbase = nbr + coff + p.chunk_stride;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int i = tid + 64 * it;
                const int r = min(i / 7, HH2), sg = i - (i / 7) * 7;
                const SampIdx si = samp_i[min(r, HH2 - 1) * HW2 + min(4 * sg, HW2 - 1)];
                const float *a = nbase + si.idx + (r == HH2 ? p.w * p.pix_stride : 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)a,
                                                 (__attribute__((address_space(3))) void *)(pf_sink + it * 64), 4, 0, 0);
            }
        }
        // ---- correlate: 24 steps = 2 k4 x 12 neighbour rows f = -4..7; row f is qy = r - f of pixel r of the quad
        // (fwd volume: neighbour at (y - qy, x - qx), CostVolMulti.lua:76-87)
        if (!(p.ablate & 2))
#pragma unroll
        for (int st = 0; st < 24; ++st) {
            const int k4 = st / 12, f = st - 12 * k4 - 4;
            float4 cur[3];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) cur[jj] = myn[k4 * PL2 + f * HP2 - jj];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qy = r - f;
                if (qy < -4 || qy > 4) continue;
                const float4 rv = rf[r][k4];
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    float a = acc[r][jj * 9 + qy + 4];
                    a = fmaf(rv.x, cur[jj].x, a);
                    a = fmaf(rv.y, cur[jj].y, a);
                    a = fmaf(rv.z, cur[jj].z, a);
                    a = fmaf(rv.w, cur[jj].w, a);
                    acc[r][jj * 9 + qy + 4] = a;
                }
            }
        }
        __syncthreads();
    }

    // ---- scale by 1/C (output:div(N), CostVolMulti.lua:100) and store this wave's 27 channels of the four pixels.  Channel of
    // accumulator i: 27 grp + i in the fwd volume, 80 - (27 grp + i) in the bwd one (mirrored window); cb = first channel / 27.
    const float cf = (float)p.C, inv = 1.f / cf;
    const int cb = dir == 0 ? grp : 2 - grp;
    if ((p.ablate & 4) && acc[0][0] != 12345.678f) return;   // profiling only: drop the stores, keep acc live
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int py = py0 + r;
        if (!(py < p.h && px < p.w)) continue;
        float v[27];                                   // ascending channel order: v[i] = channel 27 cb + i
#pragma unroll
        for (int i = 0; i < 27; ++i) {
            const float a = dir == 0 ? acc[r][i] : acc[r][26 - i];
            v[i] = POW2 ? a * inv : a / cf;
        }
        const size_t pix = (size_t)py * p.w + px;
        float *o = p.out + (size_t)b * p.out_img_stride + pix * p.out_pix_stride + (size_t)(dir * 10) * p.out_chunk_stride;
        auto rec = [&](int j) { return o + (size_t)j * p.out_chunk_stride; };
        auto full = [&](int j, int i0) {               // record j = v[i0 .. i0 + 7]
            *reinterpret_cast<float4 *>(rec(j)) = make_float4(v[i0], v[i0 + 1], v[i0 + 2], v[i0 + 3]);
            *reinterpret_cast<float4 *>(rec(j) + 4) = make_float4(v[i0 + 4], v[i0 + 5], v[i0 + 6], v[i0 + 7]);
        };
        if (cb == 0) {                                 // channels 0..26: records 0, 1, 2, slots 0..2 of record 3
            full(0, 0); full(1, 8); full(2, 16);
            *reinterpret_cast<float2 *>(rec(3)) = make_float2(v[24], v[25]);
            rec(3)[2] = v[26];
        } else if (cb == 1) {                          // channels 27..53: slots 3..7 of record 3, records 4, 5, slots 0..5 of record 6
            rec(3)[3] = v[0];
            *reinterpret_cast<float4 *>(rec(3) + 4) = make_float4(v[1], v[2], v[3], v[4]);
            full(4, 5); full(5, 13);
            *reinterpret_cast<float4 *>(rec(6)) = make_float4(v[21], v[22], v[23], v[24]);
            *reinterpret_cast<float2 *>(rec(6) + 4) = make_float2(v[25], v[26]);
        } else {                                       // channels 54..80: slots 6, 7 of record 6, records 7, 8, 9, channel 80
            *reinterpret_cast<float2 *>(rec(6) + 6) = make_float2(v[0], v[1]);
            full(7, 2); full(8, 10); full(9, 18);
            float *ol = p.out + (size_t)b * p.out_img_stride + pix * p.out_pix_stride + (size_t)20 * p.out_chunk_stride;   // [fwd80, bwd80, u, v, ub, vb, 0, 0]
            if (dir == 0) {
                ol[0] = v[26];
            } else {
                const size_t fp = ((size_t)b * p.h * p.w + pix) * 2;
                float2 f = make_float2(0.f, 0.f), fb = make_float2(0.f, 0.f);
                if (p.flow) f = *reinterpret_cast<const float2 *>(p.flow + fp);
                if (p.flow_b) fb = *reinterpret_cast<const float2 *>(p.flow_b + fp);
                ol[1] = v[26];
                ol[2] = f.x; ol[3] = f.y;
                *reinterpret_cast<float4 *>(ol + 4) = make_float4(fb.x, fb.y, 0.f, 0.f);
            }
        }
    }
}
#endif  // B2F_EXPERIMENTS (variant 8)


// ======================================================================================================
// Window-staged variant (round 2, variant 4): the gather of the two-pixel kernel issues four 16-byte tap loads per
// halo pixel and chunk half through the vector memory pipe -- 9x the bytes of the source window they come from -- in
// three dependent rounds per chunk (the 162 accumulators leave registers for 12 loads in flight).  Here the UNWARPED
// source window of the tile's halo (bounding box of the clamped tap coordinates, at most 32 x 32 pixels) is brought
// into LDS by LDS-DMA (global_load_lds_dwordx4: no registers, asynchronous) and the bilinear blend reads its four taps
// from LDS.  Work is cut into half chunks (4 channels): per half chunk  [wait DMA | barrier | blend window -> warped
// halo plane | barrier | issue the DMA of the NEXT half chunk | 2 x 81 x 4 FMAs], so the memory round trip of the next
// window runs under the FMAs of this one, and one halo plane + one 16-byte-per-pixel window + compact sampling records
// (tap coordinates, wx, wy) fit four blocks per CU (37 KB).  One direction per 128-thread block, 16 x 16 tiles, two
// pixels per thread, the same arithmetic in the same order as the other variants: identical results.  A tile whose
// flow spreads the taps over more than 32 x 32 pixels gathers from memory instead (block-uniform fallback).
#if B2F_EXPERIMENTS   // variant 4 (window staged by LDS-DMA): measured slower than variant 3, profiles/r02_corr_experiments.txt (10)
namespace v4 {
constexpr int WP = 32, WROWS = 32;                   // window pitch / rows in pixels
}  // namespace v4

// Profiling only (results are wrong): -DB2F_WIN_ABLATE=bits, 1 no DMA, 2 no blend, 4 no FMA phase, 8 no record stores
#ifndef B2F_WIN_ABLATE
#define B2F_WIN_ABLATE 0
#endif
template <bool POW2>
__global__ __launch_bounds__(128, 2) void warp_costvol_win_kernel(const CorrLaunch p)
{
    using namespace v2;
    using namespace v4;
    constexpr int NTHR = 128;
    constexpr int NG = (NH2 + NTHR - 1) / NTHR;        // 5 halo pixels per thread (the last round half full)
    __shared__ __attribute__((aligned(16))) float4 nb[PL2];              // warped halo, one 4-channel plane
    __shared__ __attribute__((aligned(16))) float4 win[WROWS * WP];      // source window, 4 channels per pixel
    __shared__ int samp_i[NH2];                                           // xl | yt << 12 | flags << 24 | valid << 26
    __shared__ float2 samp_w[NH2];                                        // wx, wy
    __shared__ int samp_f[NH2];                                           // window index | dx << 10 | dy << 11 | halo slot << 12 | valid << 22
    __shared__ int bbox[4];

    const int tid = threadIdx.x;
    const int tiles_x = (p.w + TW2 - 1) / TW2, tiles_y = (p.h + TH2 - 1) / TH2;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int dir = bid & 1;                            // 0 fwd / future map, 1 bwd / past map
    bid >>= 1;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx_i * TW2, y0 = ty_i * TH2;

    const float *ref = p.ref + (size_t)b * p.img_stride;
    const float *nbr = (dir == 0 ? p.nbr_fut : p.nbr_past) + (size_t)b * p.img_stride;

    if (tid < 2) bbox[tid] = 0x7fffffff;
    else if (tid < 4) bbox[tid] = -1;
    // ---- sampling records of the halo (once per block) and the bounding box of their taps
    {
        float2 fl[NG];
        size_t fo[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int hp = min(tid + j * NTHR, NH2 - 1);
            const int hy = hp / HW2, hx = hp - hy * HW2;
            const int y = y0 - R + hy, x = x0 - R + hx;
            const bool in = y >= 0 && y < p.h && x >= 0 && x < p.w;
            fl[j] = make_float2(0.f, 0.f);
            fo[j] = ((size_t)b * p.h * p.w + (in ? (size_t)y * p.w + x : 0)) * 2;
        }
        if (p.flow) {                                   // one uniform branch around all five loads: they are in flight together
#pragma unroll
            for (int j = 0; j < NG; ++j) fl[j] = *reinterpret_cast<const float2 *>(p.flow + fo[j]);
        }
        __syncthreads();                                // bbox initialised
        int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int hp = tid + j * NTHR;
            if (hp < NH2) {
                const int hy = hp / HW2, hx = hp - hy * HW2;
                const int y = y0 - R + hy, x = x0 - R + hx;
                int rec = 0;
                float2 wxy = make_float2(0.f, 0.f);
                if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
                    const float k = dir == 0 ? p.k : -p.k;   // nn.MulConstant(20*(f-ref)/2^(l-2)), pwc.lua:404
                    const float u = fl[j].x * k, v = fl[j].y * k;
                    int xl, yt;
                    float wx, wy;
                    bhwd_top_left(u + (float)x, p.w, xl, wx);
                    bhwd_top_left(v + (float)y, p.h, yt, wy);
                    const int fx = (xl + 1 <= p.w - 1) ? 1 : 0, fy = (yt + 1 <= p.h - 1) ? 1 : 0;
                    rec = xl | yt << 12 | fx << 24 | fy << 25 | 1 << 26;
                    wxy = make_float2(wx, wy);
                    bx0 = min(bx0, xl); by0 = min(by0, yt);
                    bx1 = max(bx1, xl + fx); by1 = max(by1, yt + fy);
                }
                samp_i[hp] = rec;
                samp_w[hp] = wxy;
            }
        }
        atomicMin(&bbox[0], bx0); atomicMin(&bbox[1], by0);
        atomicMax(&bbox[2], bx1); atomicMax(&bbox[3], by1);
    }

    float acc0[81], acc1[81];   // upper / lower pixel of the pair
#pragma unroll
    for (int i = 0; i < 81; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int tyr = tid >> 4, lx = tid & 15;
    const int py0 = y0 + 2 * tyr, px = x0 + lx;
    const bool v0 = py0 < p.h && px < p.w, v1 = py0 + 1 < p.h && px < p.w;
    const float *refp0 = ref + (size_t)(v0 ? (py0 * p.w + px) : 0) * p.pix_stride;
    const float *refp1 = ref + (size_t)(v1 ? ((py0 + 1) * p.w + px) : 0) * p.pix_stride;
    const float4 *myn = &nb[(2 * tyr + R) * HP2 + (lx + R)];

    __syncthreads();
    const int wx0 = bbox[0], wy0 = bbox[1];
    const int wh = bbox[3] - wy0 + 1;
    const bool fits = bbox[2] - wx0 + 1 <= WP && wh <= WROWS;     // block-uniform
    if (fits) {
        // records of the fast path: everything the blend needs in one word (each thread converts its own records)
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int q = tid + j * NTHR;
            if (q < NH2) {
                const int rec = samp_i[q];
                const int valid = (rec >> 26) & 1;
                const int xl = rec & 0xfff, yt = (rec >> 12) & 0xfff;
                const int wi = valid ? (yt - wy0) * WP + (xl - wx0) : 0;
                const int hy = q / HW2, hx = q - hy * HW2;
                samp_f[q] = wi | ((rec >> 24) & 3) << 10 | (hy * HP2 + hx) << 12 | valid << 22;
            }
        }
    }
    // LDS-DMA of one half chunk's window: instruction k moves rows 2k, 2k + 1 (lane = (row parity, column)), wave 0 the
    // even k, wave 1 the odd ones; columns past the image edge re-read the edge pixel (never used)
    const int wv = tid >> 6, ln = tid & 63;
    const int d_gx = min(wx0 + (ln & 31), p.w - 1);
    // (inline asm: behind the builtin the compiler waits vmcnt(0) before the next ds_read of ANY LDS array -- here the first
    // operand read of the FMA phase -- and the round trip would not overlap anything.  The DMA is ordered against the
    // window's readers by the two barriers of a half chunk and awaited by the explicit vmcnt(0) at its top.)
    const unsigned win_lds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>(win + wv * 64)));
#define B2F_WIN_DMA(hc_)                                                                            \
    do {                                                                                            \
        const float *src__ = nbr + (size_t)((hc_) >> 1) * p.chunk_stride + 4 * ((hc_) & 1);         \
        _Pragma("unroll") for (int k2 = 0; k2 < WROWS / 4; ++k2) {                                  \
            const int gy = min(wy0 + 4 * k2 + 2 * wv + (ln >> 5), p.h - 1);                         \
            const float *a = src__ + (size_t)(gy * p.w + d_gx) * p.pix_stride;                      \
            if (4 * k2 < wh && !(B2F_WIN_ABLATE & 1))                                               \
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"     \
                             :: "v"(a), "s"(win_lds + k2 * 2048u) : "memory", "m0");                \
        }                                                                                           \
    } while (0)

    const int nhalf = (p.C >> 3) * 2;
    float4 rc0 = *reinterpret_cast<const float4 *>(refp0), rc1 = *reinterpret_cast<const float4 *>(refp1);
    if (fits) B2F_WIN_DMA(0);
    for (int hc = 0; hc < nhalf; ++hc) {
        // vmcnt(0): this wave's share of the window (and the reference pixels) has landed -- a wait the compiler sees, so
        // that it knows nothing of its own is in flight below and places no vmcnt wait of its own between the DMA issue
        // and the end of the FMA phase (its counts do not include the inline-asm DMA: any such wait would drain it)
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (fits) {
            // ---- blend the warped halo plane from the window (the DMA of this half chunk was issued a phase ago)
            // (no branch inside: the five items' LDS reads are in flight together; only the last item's write is predicated)
#pragma unroll
            for (int j = 0; j < ((B2F_WIN_ABLATE & 2) ? 0 : NG); ++j) {
                const int q = min(tid + j * NTHR, NH2 - 1);
                const int f = samp_f[q];
                const float2 wxy = samp_w[q];
                const int wi = f & 1023, dx = (f >> 10) & 1, dy = ((f >> 11) & 1) * WP;
                const float4 tl = win[wi], tr = win[wi + dx], bl = win[wi + dy], br = win[wi + dy + dx];
                const float wx = wxy.x, wy = wxy.y;
                float4 w4 = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
                if (!((f >> 22) & 1)) w4 = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 v;
                v.x = fmaf(w4.w, br.x, fmaf(w4.z, bl.x, fmaf(w4.y, tr.x, w4.x * tl.x)));
                v.y = fmaf(w4.w, br.y, fmaf(w4.z, bl.y, fmaf(w4.y, tr.y, w4.x * tl.y)));
                v.z = fmaf(w4.w, br.z, fmaf(w4.z, bl.z, fmaf(w4.y, tr.z, w4.x * tl.z)));
                v.w = fmaf(w4.w, br.w, fmaf(w4.z, bl.w, fmaf(w4.y, tr.w, w4.x * tl.w)));
                if (j < NG - 1 || tid + j * NTHR < NH2) nb[(f >> 12) & 1023] = v;
            }
            __syncthreads();
        } else {
            // ---- fallback: taps straight from memory (the flow spreads this tile's taps over more than the window holds)
            const float *src0 = nbr + (size_t)(hc >> 1) * p.chunk_stride + 4 * (hc & 1);
#pragma unroll 1
            for (int j = 0; j < NG; ++j) {
                const int q = tid + j * NTHR;
                if (q < NH2) {
                    const int rec = samp_i[q];
                    const float2 wxy = samp_w[q];
                    const bool valid = (rec >> 26) & 1;
                    const int xl = rec & 0xfff, yt = (rec >> 12) & 0xfff;
                    const float *src = src0 + (size_t)(yt * p.w + xl) * p.pix_stride;
                    const int dx = ((rec >> 24) & 1) * p.pix_stride, dy = ((rec >> 25) & 1) * p.w * p.pix_stride;
                    const float4 tl = *reinterpret_cast<const float4 *>(src), tr = *reinterpret_cast<const float4 *>(src + dx);
                    const float4 bl = *reinterpret_cast<const float4 *>(src + dy), br = *reinterpret_cast<const float4 *>(src + dy + dx);
                    const float wx = wxy.x, wy = wxy.y;
                    float4 w4 = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
                    if (!valid) w4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 v;
                    v.x = fmaf(w4.w, br.x, fmaf(w4.z, bl.x, fmaf(w4.y, tr.x, w4.x * tl.x)));
                    v.y = fmaf(w4.w, br.y, fmaf(w4.z, bl.y, fmaf(w4.y, tr.y, w4.x * tl.y)));
                    v.z = fmaf(w4.w, br.z, fmaf(w4.z, bl.z, fmaf(w4.y, tr.z, w4.x * tl.z)));
                    v.w = fmaf(w4.w, br.w, fmaf(w4.z, bl.w, fmaf(w4.y, tr.w, w4.x * tl.w)));
                    const int hy = q / HW2, hx = q - hy * HW2;
                    nb[hy * HP2 + hx] = v;
                }
            }
            __syncthreads();
        }
        // the window and the reference pixels of the next half chunk: in flight under the FMAs
        const float4 ra = rc0, rb = rc1;
        if (fits && hc + 1 < nhalf) B2F_WIN_DMA(hc + 1);
        {
            const int hn = min(hc + 1, nhalf - 1);      // (the last iteration re-reads its own: no branch around the loads)
            const size_t o = (size_t)(hn >> 1) * p.chunk_stride + 4 * (hn & 1);
            rc0 = *reinterpret_cast<const float4 *>(refp0 + o);
            rc1 = *reinterpret_cast<const float4 *>(refp1 + o);
        }
        // ---- correlate: 10 neighbour rows f = -4..5; row f is qy = -f of the upper pixel and qy = 1 - f of the lower one
#pragma unroll
        for (int st = 0; st < ((B2F_WIN_ABLATE & 4) ? 0 : 10); ++st) {
            const int f = st - 4;
            float4 cur[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) cur[j] = myn[f * HP2 - (j - 4)];   // qx = j - 4
            if (f <= 4) {
                const int qy = -f;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float a = acc0[j * 9 + qy + 4];
                    a = fmaf(ra.x, cur[j].x, a);
                    a = fmaf(ra.y, cur[j].y, a);
                    a = fmaf(ra.z, cur[j].z, a);
                    a = fmaf(ra.w, cur[j].w, a);
                    acc0[j * 9 + qy + 4] = a;
                }
            }
            if (f >= -3) {
                const int qy = 1 - f;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float a = acc1[j * 9 + qy + 4];
                    a = fmaf(rb.x, cur[j].x, a);
                    a = fmaf(rb.y, cur[j].y, a);
                    a = fmaf(rb.z, cur[j].z, a);
                    a = fmaf(rb.w, cur[j].w, a);
                    acc1[j * 9 + qy + 4] = a;
                }
            }
        }
    }
#undef B2F_WIN_DMA

    // ---- scale by 1/C (output:div(N), CostVolMulti.lua:100) and store the record slots of both pixels
    const float cf = (float)p.C, inv = 1.f / cf;
    auto store_px = [&](float *acc, bool valid, int py) {
        if (!valid) return;
        if ((B2F_WIN_ABLATE & 8) && acc[0] != 12345.678f) return;
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
    store_px(acc0, v0, py0);
    store_px(acc1, v1, py0 + 1);
}

#endif  // B2F_EXPERIMENTS

hipError_t launch_warp_costvol(const CorrLaunch &p_in, hipStream_t s)
{
    if (p_in.C % 8 != 0 || p_in.pix_stride % 4 != 0 || p_in.chunk_stride % 4 != 0 || p_in.out_pix_stride % 4 != 0 ||
        p_in.out_chunk_stride % 4 != 0)
        return hipErrorInvalidValue;
    CorrLaunch p = p_in;
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    const dim3 grid((unsigned)(tiles_x * tiles_y * p.B));
    const int t2x = (p.w + v2::TW2 - 1) / v2::TW2, t2y = (p.h + v2::TH2 - 1) / v2::TH2;
    const dim3 g2((unsigned)(t2x * t2y * p.B));
    // Instantiations of the same arithmetic (bit-identical results; p.variant >= 0 forces one):
    //   3  two pixels per thread, 16 x 16 tiles, one direction per 128-thread block (four blocks per CU): launches that
    //      fill the chip -- measured at batch 16, level 3 / 4 / 5: 0.716 / 0.302 / 0.113 ms against 0.821 / 0.328 / 0.152
    //   2  the same with both directions in one 256-thread block (two blocks per CU): 0.757 / 0.318 / 0.115
    //   1  one pixel per thread, all 24 gather loads of a chunk in flight: at most one round of two blocks per CU
    //   0  one pixel per thread, 8 x 16 tiles, three blocks per CU: in between
    //   4  variant 3 with the source window staged in LDS by LDS-DMA and the blend read from LDS (see its header); maps of
    //      up to 4096 x 4096 pixels (12-bit tap coordinates in its sampling records).  Opt-in (corr_variant = 4): measured
    //      0.72 / 0.30 / 0.108 ms at levels 3 / 4 / 5 against 0.69 / 0.29 / 0.113 of variant 3 -- the DMA round trip does
    //      hide under the FMAs (ablating it saves 0.03 ms), but the half-chunk phases double the barriers and the
    //      per-phase LDS latency chains (profiles/r02_corr_experiments.txt (10))
    //   5  the persistent "unit" form of b2f_corr5.hip (C a multiple of 16; other C run variant 3)
    //   6  its role-specialised form (FMA waves / gather waves with the LDS-DMA window), opt-in
    //   8  four pixels x three qx columns per thread, 192-thread blocks, one direction per block (round 5; see its header)
    //   7  ten unit waves + six gather waves per 1 024-thread block: the default for maps of up to 2 048 pixels (levels 6, 7 of a
    //      full-HD triplet: 0.043 / 0.031 ms against 0.050 / 0.038 of variant 5)
    const bool win_ok = p.w <= 4096 && p.h <= 4096;
    // default: by the map size for the small maps (so that a triplet's kernel does not depend on the batch it is computed in; the
    // instantiations are bit-identical anyway): the sixteen-wave unit kernel up to 2 048 pixels (levels 6, 7 of a full-HD triplet:
    // 0.043 / 0.031 ms against 0.060 / 0.081 of the block-per-tile kernels at batch 16), else by the launch size: the two-pixel
    // kernel needs about one full round of its 1 024 resident blocks (a single triplet's 256 x 480 level has 960), anything smaller
    // also runs the sixteen-wave unit kernel (round 5, single triplet at 3x1024x1920: levels 4 / 5 0.027 / 0.020 ms against
    // 0.037 / 0.046 of the one-pixel kernels, level 3 0.046 on the two-pixel kernel against 0.060: 0.196 -> 0.145 ms per call)
#if !B2F_EXPERIMENTS
    if (p.variant == 2 || p.variant == 4 || p.variant == 6 || p.variant == 8) p.variant = 3;     // experiment kernels (tools/experiments): not in this build
#endif
    const unsigned n_cu = (unsigned)device_cu_count(), round2 = 15 * n_cu / 4;   // 960 of the two-pixel kernel's 1 024 resident blocks on 256 CUs
    int variant = p.variant >= 0 ? (p.variant == 4 && !win_ok ? 3 : p.variant)
                  : (p.ablate ? 0 : ((p.h * p.w <= 2048 || 2 * g2.x < round2) && warp_costvol_unit_supported(p)) ? 7 : 2 * g2.x >= round2 ? 3 : grid.x <= 2u * n_cu ? 1 : 0);
    if ((variant == 5 || variant == 6 || variant == 7) && !warp_costvol_unit_supported(p)) variant = 3;
    if (variant == 5) return launch_warp_costvol_unit(p, s);
    if (variant == 7) return launch_warp_costvol_gw(p, s);
#if B2F_EXPERIMENTS
    if (variant == 6) return launch_warp_costvol_spec(p, s);
    if (variant == 4) {
        if (pow2) hipLaunchKernelGGL((warp_costvol_win_kernel<true>), dim3(2 * g2.x), dim3(128), 0, s, p);
        else hipLaunchKernelGGL((warp_costvol_win_kernel<false>), dim3(2 * g2.x), dim3(128), 0, s, p);
    } else if (variant == 2) {
        if (pow2) hipLaunchKernelGGL((warp_costvol_2px_kernel<true, 2>), g2, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((warp_costvol_2px_kernel<false, 2>), g2, dim3(256), 0, s, p);
    } else
#endif
#if B2F_EXPERIMENTS
    if (variant == 8) {
        if (pow2) hipLaunchKernelGGL((warp_costvol_4px_kernel<true>), dim3(2 * g2.x), dim3(192), 0, s, p);
        else hipLaunchKernelGGL((warp_costvol_4px_kernel<false>), dim3(2 * g2.x), dim3(192), 0, s, p);
    } else
#endif
    if (variant == 3) {
        if (pow2) hipLaunchKernelGGL((warp_costvol_2px_kernel<true, 1>), dim3(2 * g2.x), dim3(128), 0, s, p);
        else hipLaunchKernelGGL((warp_costvol_2px_kernel<false, 1>), dim3(2 * g2.x), dim3(128), 0, s, p);
    } else if (variant == 1) {
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
