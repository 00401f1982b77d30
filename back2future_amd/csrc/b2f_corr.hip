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
// Output record per pixel (NHWC): [fwd 81 | bwd 81 | u | v], channel order inside a
// volume is x-major, c = (qx+4)*9 + (qy+4) (CostVolMulti.lua:66-67,92):
//   fwd[c] = 1/C * sum_k ref[y,x,k] * W3[y-qy, x-qx, k]      (out of range -> 0)
//   bwd[c] = 1/C * sum_k ref[y,x,k] * W1[y+qy, x+qx, k]
// with W3/W1 = neighbour map sampled at (x + k*u, y + k*v) / (x - k*u, y - k*v), coordinates
// clamped to the border, top-left weight 1 - frac (BilinearSamplerBHWD.cu:6-20).
//
// Block = 256 threads, output tile 8 x 16 pixels.  Threads 0..127 own one pixel of the
// fwd volume each, threads 128..255 the same pixels of the bwd volume: 81 accumulators in
// VGPRs.  Channels are walked in chunks of 8: the block gathers the warped 16 x 24 halo of
// both neighbour maps into LDS (layout [k4][row pitch 32] of float4: a wave's ds_read_b128
// are conflict-free and every displacement is an immediate offset), then every thread does
// 81 x (1 ds_read_b128 + 4 FMA) per float4 of its reference pixel.  The bwd thread runs the
// same code on the mirrored window (bwd channel c uses offset +q = fwd offset of channel
// 80 - c), so there is one inner loop.
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
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <bool POW2, int NK4, int HPP>
__global__ __launch_bounds__(256, 2) void warp_costvol_kernel(const CorrLaunch p)
{
    __shared__ __attribute__((aligned(16))) float4 nb[2][NK4][HH * HPP];   // [map][k4][pixel]
    __shared__ float4 samp_w[2][NHALO];                                  // 12 KB
    __shared__ SampIdx samp_i[2][NHALO];                                 // 6 KB

    const int tid = threadIdx.x;
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    int bid = p.ablate & 8 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx_i * TW, y0 = ty_i * TH;

    const float *ref = p.ref + (size_t)b * p.img_stride;
    const float *nbr[2] = {p.nbr_fut + (size_t)b * p.img_stride, p.nbr_past + (size_t)b * p.img_stride};

    // ---- sampling records for the halo (once per block) ----
    for (int i = tid; i < 2 * NHALO; i += 256) {
        const int map = i / NHALO, hp = i - map * NHALO;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int y = y0 - R + hy, x = x0 - R + hx;
        float4 wgt = make_float4(0.f, 0.f, 0.f, 0.f);
        SampIdx si;
        si.idx = 0; si.flags = 0;
        if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
            float u = 0.f, v = 0.f;
            if (p.flow) {
                const float2 f = *reinterpret_cast<const float2 *>(p.flow + ((size_t)(b * p.h + y) * p.w + x) * 2);
                const float k = map == 0 ? p.k : -p.k;   // nn.MulConstant(20*(f-ref)/2^(l-2)), pwc.lua:404
                u = f.x * k; v = f.y * k;
            }
            int xl, yt;
            float wx, wy;
            top_left(u + (float)x, p.w, xl, wx);
            top_left(v + (float)y, p.h, yt, wy);
            si.idx = yt * p.w + xl;
            si.flags = ((xl + 1 <= p.w - 1) ? 1 : 0) | ((yt + 1 <= p.h - 1) ? 2 : 0);
            wgt = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
        }
        samp_w[map][hp] = wgt;
        samp_i[map][hp] = si;
    }

    float acc[81];
#pragma unroll
    for (int i = 0; i < 81; ++i) acc[i] = 0.f;

    const int pl = tid & 127, dir = tid >> 7;
    const int ly = pl >> 4, lx = pl & 15;
    const int py = y0 + ly, px = x0 + lx;
    const bool pvalid = py < p.h && px < p.w;
    const float *refp = ref + (size_t)(pvalid ? (py * p.w + px) : 0) * p.pix_stride;
    const float4 *myn = &nb[dir][0][(ly + R) * HPP + (lx + R)];

    __syncthreads();
    for (int c0 = 0; c0 < p.C; c0 += 4 * NK4) {
        // ---- gather + blend the warped halo chunk into LDS ----
#pragma unroll 3   // 12 float4 gathers in flight per thread: enough to cover L2 latency
        for (int i = tid; i < 2 * NK4 * NHALO; i += 256) {
            const int k4 = i % NK4;
            const int rest = i / NK4;
            const int map = rest / NHALO, hp = rest - map * NHALO;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const float4 wg = samp_w[map][hp];
            const SampIdx si = samp_i[map][hp];
            const float *src = nbr[map] + (size_t)((p.ablate & 16) ? 0 : si.idx) * p.pix_stride + c0 + 4 * k4;
            const int dx = si.flags & 1, dy = (si.flags & 2) ? p.w : 0;
            float4 tl = wg, tr = wg, bl = wg, br = wg;
            if (!(p.ablate & 1)) {
                tl = *reinterpret_cast<const float4 *>(src);
                if (!(p.ablate & 32)) {
                    tr = *reinterpret_cast<const float4 *>(src + dx * p.pix_stride);
                    bl = *reinterpret_cast<const float4 *>(src + (size_t)dy * p.pix_stride);
                    br = *reinterpret_cast<const float4 *>(src + (size_t)(dy + dx) * p.pix_stride);
                }
            }
            float4 v;
            v.x = wg.x * tl.x + wg.y * tr.x + wg.z * bl.x + wg.w * br.x;
            v.y = wg.x * tl.y + wg.y * tr.y + wg.z * bl.y + wg.w * br.y;
            v.z = wg.x * tl.z + wg.y * tr.z + wg.z * bl.z + wg.w * br.z;
            v.w = wg.x * tl.w + wg.y * tr.w + wg.z * bl.w + wg.w * br.w;
            nb[map][k4][hy * HPP + hx] = v;
        }
        // reference pixel chunk (address is clamped to a valid pixel for out-of-image lanes)
        float4 r01[NK4];
#pragma unroll
        for (int q = 0; q < NK4; ++q) r01[q] = *reinterpret_cast<const float4 *>(refp + c0 + 4 * q);
        __syncthreads();
        // ---- correlate ----
        if (!(p.ablate & 2))
        // 18 steps = 2 k4 x 9 qx columns; the 9 ds_read_b128 of step s+1 are issued before the
        // 36 FMAs of step s (two register sets), so LDS latency hides under the FMAs of the
        // same wave instead of relying on other waves.
        {
            float4 va[9], vb[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) va[j] = myn[-(j - 4) * HPP + 4];
#pragma unroll
            for (int st = 0; st < 9 * NK4; ++st) {
                const int k4 = st / 9, g = st - 9 * k4;
                const float4 r = r01[k4];
                float4 *cur = (st & 1) ? vb : va, *nxt = (st & 1) ? va : vb;
                if (st + 1 < 9 * NK4) {
                    const int k4n = (st + 1) / 9, gn = (st + 1) - 9 * k4n;
#pragma unroll
                    for (int j = 0; j < 9; ++j) nxt[j] = myn[k4n * (HH * HPP) - (j - 4) * HPP - (gn - 4)];
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
    float *o = p.out + ((size_t)(b * p.h + py) * p.w + px) * p.rec + dir * 81;
    const float cf = (float)p.C, inv = 1.f / cf;
    if (dir == 0) {
#pragma unroll
        for (int c = 0; c < 81; ++c) o[c] = POW2 ? acc[c] * inv : acc[c] / cf;   // output:div(N), CostVolMulti.lua:100
    } else {
#pragma unroll
        for (int c = 0; c < 81; ++c) o[80 - c] = POW2 ? acc[c] * inv : acc[c] / cf;
        const size_t pix = (size_t)(b * p.h + py) * p.w + px;
        float2 f = make_float2(0.f, 0.f), fb = make_float2(0.f, 0.f);
        if (p.flow) f = *reinterpret_cast<const float2 *>(p.flow + pix * 2);
        if (p.flow_b) fb = *reinterpret_cast<const float2 *>(p.flow_b + pix * 2);
        o[81] = f.x;
        o[82] = f.y;
        // floats 164..167: (ub, vb, 0, 0) inside a 168-float record; for 164-float records they
        // belong to the next pixel, except after the very last record, where the decoder's 21st
        // K-chunk still reads them (with zero weights): keep them finite.
        if (p.rec == kCvRecFull || (b == p.B - 1 && py == p.h - 1 && px == p.w - 1)) {
            o[83] = fb.x; o[84] = fb.y; o[85] = 0.f; o[86] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------------
// v2: register-blocked variant.  Same maths, different work split: a thread owns a vertical
// strip of 3 pixels, one direction and one third of the qx range (3 columns x 9 qy = 27
// displacements per pixel, 81 accumulators).  The 3 pixels x 9 qy of one column share the 11
// halo rows y0-4 .. y0+6, so 11 ds_read_b128 feed 27 x 4 FMAs: 0.41 LDS floats per FMA
// instead of 1.0 in the one-pixel-per-thread kernel above.  Block = 6 waves = (2 directions)
// x (3 qx thirds), tile = 12 x 16 pixels (64 strips), halo 20 x 24.
namespace v2 {
constexpr int TH2 = 12, TW2 = 16;
constexpr int HH2 = TH2 + 2 * R;     // 20
constexpr int HWD2 = TW2 + 2 * R;    // 24
constexpr int NHALO2 = HH2 * HWD2;   // 480
constexpr int NT2 = 384;
}  // namespace v2

template <bool POW2>
__global__ __launch_bounds__(384, 2) void warp_costvol_v2_kernel(const CorrLaunch p)
{
    using namespace v2;
    __shared__ __attribute__((aligned(16))) float4 nb[2][2][HH2 * HP];   // [map][k4][pixel] 40 KB
    __shared__ float4 samp_w[2][NHALO2];                                  // 15 KB
    __shared__ SampIdx samp_i[2][NHALO2];                                 // 7.5 KB

    const int tid = threadIdx.x;
    const int tiles_x = (p.w + TW2 - 1) / TW2, tiles_y = (p.h + TH2 - 1) / TH2;
    int bid = p.ablate & 8 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx_i * TW2, y0 = ty_i * TH2;

    const float *ref = p.ref + (size_t)b * p.img_stride;
    const float *nbr[2] = {p.nbr_fut + (size_t)b * p.img_stride, p.nbr_past + (size_t)b * p.img_stride};

    // ---- sampling records for the halo (once per block) ----
    for (int i = tid; i < 2 * NHALO2; i += NT2) {
        const int map = i / NHALO2, hp = i - map * NHALO2;
        const int hy = hp / HWD2, hx = hp - hy * HWD2;
        const int y = y0 - R + hy, x = x0 - R + hx;
        float4 wgt = make_float4(0.f, 0.f, 0.f, 0.f);
        SampIdx si;
        si.idx = 0; si.flags = 0;
        if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
            float u = 0.f, v = 0.f;
            if (p.flow) {
                const float2 f = *reinterpret_cast<const float2 *>(p.flow + ((size_t)(b * p.h + y) * p.w + x) * 2);
                const float k = map == 0 ? p.k : -p.k;   // nn.MulConstant(20*(f-ref)/2^(l-2)), pwc.lua:404
                u = f.x * k; v = f.y * k;
            }
            int xl, yt;
            float wx, wy;
            top_left(u + (float)x, p.w, xl, wx);
            top_left(v + (float)y, p.h, yt, wy);
            si.idx = yt * p.w + xl;
            si.flags = ((xl + 1 <= p.w - 1) ? 1 : 0) | ((yt + 1 <= p.h - 1) ? 2 : 0);
            wgt = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
        }
        samp_w[map][hp] = wgt;
        samp_i[map][hp] = si;
    }

    float acc[81];   // [pixel i 0..2][column dxi 0..2][qy+4 0..8]
#pragma unroll
    for (int i = 0; i < 81; ++i) acc[i] = 0.f;

    // wave -> (direction, qx third); lane -> (strip row, column)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int dir = wave / 3, third = wave - 3 * dir;
    const int sr = lane >> 4, lx = lane & 15;
    const int ls = sr * 3;                       // first tile row of the strip
    const int px = x0 + lx;
    const int qx0 = 3 * third - 4;               // first qx of this thread's three columns
    // reference pixels of the strip (addresses clamped into the image; masked at the store)
    const float *refp[3];
    bool pv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int py = y0 + ls + i;
        pv[i] = py < p.h && px < p.w;
        refp[i] = ref + (size_t)(pv[i] ? (py * p.w + px) : 0) * p.pix_stride;
    }
    // halo pixel of (row ls, displacement qy=+4 of pixel 0 .. ) : row ls + j, column lx + 4 - qx
    const float4 *myn = &nb[dir][0][ls * HP + lx + R - qx0];

    __syncthreads();
    for (int c0 = 0; c0 < p.C; c0 += 8) {
        // ---- gather + blend the warped halo chunk into LDS ----
#pragma unroll
        for (int i = tid; i < 2 * 2 * NHALO2; i += NT2) {
            const int k4 = i & 1;
            const int rest = i >> 1;
            const int map = rest / NHALO2, hp = rest - map * NHALO2;
            const int hy = hp / HWD2, hx = hp - hy * HWD2;
            const float4 wg = samp_w[map][hp];
            const SampIdx si = samp_i[map][hp];
            const float *src = nbr[map] + (size_t)si.idx * p.pix_stride + c0 + 4 * k4;
            const int dx = si.flags & 1, dy = (si.flags & 2) ? p.w : 0;
            const float4 tl = *reinterpret_cast<const float4 *>(src);
            const float4 tr = *reinterpret_cast<const float4 *>(src + dx * p.pix_stride);
            const float4 bl = *reinterpret_cast<const float4 *>(src + (size_t)dy * p.pix_stride);
            const float4 br = *reinterpret_cast<const float4 *>(src + (size_t)(dy + dx) * p.pix_stride);
            float4 v;
            v.x = wg.x * tl.x + wg.y * tr.x + wg.z * bl.x + wg.w * br.x;
            v.y = wg.x * tl.y + wg.y * tr.y + wg.z * bl.y + wg.w * br.y;
            v.z = wg.x * tl.z + wg.y * tr.z + wg.z * bl.z + wg.w * br.z;
            v.w = wg.x * tl.w + wg.y * tr.w + wg.z * bl.w + wg.w * br.w;
            nb[map][k4][hy * HP + hx] = v;
        }
        float4 rr[2][3];
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
            for (int i = 0; i < 3; ++i) rr[k4][i] = *reinterpret_cast<const float4 *>(refp[i] + c0 + 4 * k4);
        __syncthreads();
        // ---- correlate ----
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4) {
            const float4 *base = myn + k4 * (HH2 * HP);
#pragma unroll
            for (int dxi = 0; dxi < 3; ++dxi) {
                float4 v[11];
#pragma unroll
                for (int j = 0; j < 11; ++j) v[j] = base[j * HP - dxi];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float4 r = rr[k4][i];
#pragma unroll
                    for (int q = 0; q < 9; ++q) {         // q = qy + 4, halo row j = i + 4 - qy = i + 8 - q
                        float a = acc[(i * 3 + dxi) * 9 + q];
                        const float4 w = v[i + 8 - q];
                        a = fmaf(r.x, w.x, a);
                        a = fmaf(r.y, w.y, a);
                        a = fmaf(r.z, w.z, a);
                        a = fmaf(r.w, w.w, a);
                        acc[(i * 3 + dxi) * 9 + q] = a;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // one column at a time: bounds live ds_read results
            }
        }
        __syncthreads();
    }

    const float cf = (float)p.C, inv = 1.f / cf;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (!pv[i]) continue;
        const size_t pix = (size_t)(b * p.h + y0 + ls + i) * p.w + px;
        float *o = p.out + pix * p.rec;
        if (dir == 0) {
            // fwd channels c = (qx+4)*9 + (qy+4), this thread: 27 consecutive ones from 27*third
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const float a = acc[(i * 3 + d) * 9 + q];
                    o[27 * third + d * 9 + q] = POW2 ? a * inv : a / cf;   // output:div(N), CostVolMulti.lua:100
                }
        } else {
            // bwd channel of displacement (+qx,+qy) is the mirrored index 80 - c
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const float a = acc[(i * 3 + d) * 9 + q];
                    o[81 + 80 - (27 * third + d * 9 + q)] = POW2 ? a * inv : a / cf;
                }
            if (third == 0) {
                float2 f = make_float2(0.f, 0.f), fb = make_float2(0.f, 0.f);
                if (p.flow) f = *reinterpret_cast<const float2 *>(p.flow + pix * 2);
                if (p.flow_b) fb = *reinterpret_cast<const float2 *>(p.flow_b + pix * 2);
                o[162] = f.x;
                o[163] = f.y;
                if (p.rec == kCvRecFull || pix == (size_t)p.B * p.h * p.w - 1) {
                    o[164] = fb.x; o[165] = fb.y; o[166] = 0.f; o[167] = 0.f;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// v3: the v1 tiling (8 x 16 pixels, 16 x 24 halo, 8-channel chunks) with THREE threads per
// (pixel, direction): each owns one third of the qx range = 27 consecutive output channels.
// 27 accumulators instead of 81 -> ~64 VGPRs, 12 waves per block and 24 per CU, so the
// latency-bound halo gather (2 items per thread, all 8 loads in flight at once) has three
// times as many loads in flight per CU.  LDS traffic per FMA is the same as v1.
template <bool POW2>
__global__ __launch_bounds__(768) void warp_costvol_v3_kernel(const CorrLaunch p)
{
    __shared__ __attribute__((aligned(16))) float4 nb[2][2][HH * HP];   // [map][k4][pixel] 32 KB
    __shared__ float4 samp_w[2][NHALO];                                  // 12 KB
    __shared__ SampIdx samp_i[2][NHALO];                                 // 6 KB

    const int tid = threadIdx.x;
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    int bid = p.ablate & 8 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx_i * TW, y0 = ty_i * TH;

    const float *ref = p.ref + (size_t)b * p.img_stride;
    const float *nbr[2] = {p.nbr_fut + (size_t)b * p.img_stride, p.nbr_past + (size_t)b * p.img_stride};

    for (int i = tid; i < 2 * NHALO; i += 768) {
        const int map = i / NHALO, hp = i - map * NHALO;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int y = y0 - R + hy, x = x0 - R + hx;
        float4 wgt = make_float4(0.f, 0.f, 0.f, 0.f);
        SampIdx si;
        si.idx = 0; si.flags = 0;
        if (y >= 0 && y < p.h && x >= 0 && x < p.w) {
            float u = 0.f, v = 0.f;
            if (p.flow) {
                const float2 f = *reinterpret_cast<const float2 *>(p.flow + ((size_t)(b * p.h + y) * p.w + x) * 2);
                const float k = map == 0 ? p.k : -p.k;   // nn.MulConstant(20*(f-ref)/2^(l-2)), pwc.lua:404
                u = f.x * k; v = f.y * k;
            }
            int xl, yt;
            float wx, wy;
            top_left(u + (float)x, p.w, xl, wx);
            top_left(v + (float)y, p.h, yt, wy);
            si.idx = yt * p.w + xl;
            si.flags = ((xl + 1 <= p.w - 1) ? 1 : 0) | ((yt + 1 <= p.h - 1) ? 2 : 0);
            wgt = make_float4(wx * wy, (1.f - wx) * wy, wx * (1.f - wy), (1.f - wx) * (1.f - wy));
        }
        samp_w[map][hp] = wgt;
        samp_i[map][hp] = si;
    }

    float acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.f;

    // 128 threads (2 waves) per role; role = direction * 3 + third is wave-uniform
    const int pl = tid & 127, role = tid >> 7;
    const int dir = role / 3, third = role - 3 * dir;
    const int ly = pl >> 4, lx = pl & 15;
    const int py = y0 + ly, px = x0 + lx;
    const bool pvalid = py < p.h && px < p.w;
    const float *refp = ref + (size_t)(pvalid ? (py * p.w + px) : 0) * p.pix_stride;
    // displacement (qx, qy): qx = 3*third - 4 + g (g = 0..2), neighbour at halo (ly+4-qy, lx+4-qx)
    const float4 *myn = &nb[dir][0][(ly + R) * HP + (lx + R) - (3 * third - 4)];

    __syncthreads();
    // Per-thread gather items (fixed over chunks): 2 * 2 * NHALO = 1536 (map, halo pixel, k4)
    // triples = 2 per thread.  The 8 tap loads of chunk c+1 are issued BEFORE the FMAs of chunk
    // c and blended into LDS after them, so the latency of the data-dependent gather (L2 / HBM)
    // hides under the correlation of the previous chunk.
    typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vectors: register arrays of HIP float4 go to scratch
    f32x4 wgt[2];
    int g_off[2], g_dx[2], g_dy[2], g_map[2];
    f32x4 *g_dst[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = tid + it * 768;
        const int k4 = i & 1;
        const int rest = i >> 1;
        const int map = rest / NHALO, hp = rest - map * NHALO;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const float4 w4 = samp_w[map][hp];
        wgt[it] = f32x4{w4.x, w4.y, w4.z, w4.w};
        const SampIdx si = samp_i[map][hp];
        g_map[it] = map;
        g_off[it] = si.idx * p.pix_stride + 4 * k4;
        g_dx[it] = (si.flags & 1) * p.pix_stride;
        g_dy[it] = (si.flags & 2) ? p.w * p.pix_stride : 0;
        g_dst[it] = reinterpret_cast<f32x4 *>(&nb[map][k4][hy * HP + hx]);
    }
    f32x4 t_tl[2], t_tr[2], t_bl[2], t_br[2], r0, r1;
#define B2F_GATHER_ISSUE(c0_)                                                              \
    do {                                                                                   \
        _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                 \
            const float *src = (g_map[it] ? nbr[1] : nbr[0]) + g_off[it] + (c0_);          \
            t_tl[it] = *reinterpret_cast<const f32x4 *>(src);                              \
            t_tr[it] = *reinterpret_cast<const f32x4 *>(src + g_dx[it]);                   \
            t_bl[it] = *reinterpret_cast<const f32x4 *>(src + g_dy[it]);                   \
            t_br[it] = *reinterpret_cast<const f32x4 *>(src + g_dy[it] + g_dx[it]);        \
        }                                                                                  \
        r0 = *reinterpret_cast<const f32x4 *>(refp + (c0_));                               \
        r1 = *reinterpret_cast<const f32x4 *>(refp + (c0_) + 4);                           \
    } while (0)
#define B2F_GATHER_BLEND()                                                                 \
    do {                                                                                   \
        _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                 \
            const f32x4 wg = wgt[it];                                                      \
            *g_dst[it] = wg.x * t_tl[it] + wg.y * t_tr[it] + wg.z * t_bl[it] + wg.w * t_br[it]; \
        }                                                                                  \
    } while (0)

    B2F_GATHER_ISSUE(0);
    B2F_GATHER_BLEND();
    f32x4 rc0 = r0, rc1 = r1;
    __syncthreads();
    for (int c0 = 0; c0 < p.C; c0 += 8) {
        // branch-free: the last iteration re-fetches its own chunk (harmless) so that the loop body is
        // one basic block and the scheduling barriers below hold
        const int cn = min(c0 + 8, p.C - 8);
        B2F_GATHER_ISSUE(cn);
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4) {
            const f32x4 r = k4 ? rc1 : rc0;
            const f32x4 *base = reinterpret_cast<const f32x4 *>(myn) + k4 * (HH * HP);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                f32x4 v[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) v[j] = base[-(j - 4) * HP - g];
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float a = acc[g * 9 + j];
                    a = fmaf(r.x, v[j].x, a);
                    a = fmaf(r.y, v[j].y, a);
                    a = fmaf(r.z, v[j].z, a);
                    a = fmaf(r.w, v[j].w, a);
                    acc[g * 9 + j] = a;
                }
                __builtin_amdgcn_sched_barrier(0);   // one column at a time: bounds live ds_read results
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();          // every wave is done reading this chunk's halo
        B2F_GATHER_BLEND();
        rc0 = r0; rc1 = r1;
        __syncthreads();
    }
#undef B2F_GATHER_ISSUE
#undef B2F_GATHER_BLEND

    if (!pvalid) return;
    if ((p.ablate & 4) && acc[0] != 12345.678f) return;
    const size_t pix = (size_t)(b * p.h + py) * p.w + px;
    float *o = p.out + pix * p.rec;
    const float cf = (float)p.C, inv = 1.f / cf;
    if (dir == 0) {
#pragma unroll
        for (int c = 0; c < 27; ++c) o[27 * third + c] = POW2 ? acc[c] * inv : acc[c] / cf;   // output:div(N), CostVolMulti.lua:100
    } else {
#pragma unroll
        for (int c = 0; c < 27; ++c) o[81 + 80 - (27 * third + c)] = POW2 ? acc[c] * inv : acc[c] / cf;
        if (third == 0) {
            float2 f = make_float2(0.f, 0.f), fb = make_float2(0.f, 0.f);
            if (p.flow) f = *reinterpret_cast<const float2 *>(p.flow + pix * 2);
            if (p.flow_b) fb = *reinterpret_cast<const float2 *>(p.flow_b + pix * 2);
            o[162] = f.x;
            o[163] = f.y;
            if (p.rec == kCvRecFull || pix == (size_t)p.B * p.h * p.w - 1) {
                o[164] = fb.x; o[165] = fb.y; o[166] = 0.f; o[167] = 0.f;
            }
        }
    }
}

hipError_t launch_warp_costvol(const CorrLaunch &p_in, hipStream_t s)
{
    if (p_in.C % 8 != 0 || p_in.pix_stride % 4 != 0) return hipErrorInvalidValue;
    static const int variant = getenv("B2F_CORR_VARIANT") ? atoi(getenv("B2F_CORR_VARIANT")) : 1;
    static const int ablate = getenv("B2F_CORR_ABLATE") ? atoi(getenv("B2F_CORR_ABLATE")) : 0;
    CorrLaunch p = p_in;
    p.ablate = ablate;
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    if (variant == 1) {
        const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
        dim3 grid((unsigned)(tiles_x * tiles_y * p.B));
        static const int nk4 = getenv("B2F_CORR_NK4") ? atoi(getenv("B2F_CORR_NK4")) : 2;
        if (nk4 == 4 && p.C % 16 == 0) {
            if (pow2) hipLaunchKernelGGL((warp_costvol_kernel<true, 4, 24>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((warp_costvol_kernel<false, 4, 24>), grid, dim3(256), 0, s, p);
        } else if (nk4 == 3) {   // 8-channel chunks, unpadded pitch
            if (pow2) hipLaunchKernelGGL((warp_costvol_kernel<true, 2, 24>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((warp_costvol_kernel<false, 2, 24>), grid, dim3(256), 0, s, p);
        } else {
            if (pow2) hipLaunchKernelGGL((warp_costvol_kernel<true, 2, 32>), grid, dim3(256), 0, s, p);
            else hipLaunchKernelGGL((warp_costvol_kernel<false, 2, 32>), grid, dim3(256), 0, s, p);
        }
        return hipGetLastError();
    }
    if (variant == 3) {
        const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
        dim3 grid((unsigned)(tiles_x * tiles_y * p.B));
        if (pow2) hipLaunchKernelGGL(warp_costvol_v3_kernel<true>, grid, dim3(768), 0, s, p);
        else hipLaunchKernelGGL(warp_costvol_v3_kernel<false>, grid, dim3(768), 0, s, p);
        return hipGetLastError();
    }
    const int tiles_x = (p.w + v2::TW2 - 1) / v2::TW2, tiles_y = (p.h + v2::TH2 - 1) / v2::TH2;
    dim3 grid((unsigned)(tiles_x * tiles_y * p.B));
    if (pow2) hipLaunchKernelGGL(warp_costvol_v2_kernel<true>, grid, dim3(v2::NT2), 0, s, p);
    else hipLaunchKernelGGL(warp_costvol_v2_kernel<false>, grid, dim3(v2::NT2), 0, s, p);
    return hipGetLastError();
}

// Generic single-direction cost volume for windows other than the shipped 9x9
// (createModelMulti(nil) uses win 5, pwc.lua:88).  One thread per output element;
// not on the hot path.
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
