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

template <bool POW2>
__global__ __launch_bounds__(256, 3) void warp_costvol_kernel(const CorrLaunch p)
{
    __shared__ __attribute__((aligned(16))) float4 nb[2][2][HH * HP];   // [map][k4][pixel] 32 KB
    __shared__ float4 samp_w[2][NHALO];                                  // 12 KB
    __shared__ SampIdx samp_i[2][NHALO];                                 // 6 KB

    const int tid = threadIdx.x;
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    int bid = blockIdx.x;
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
    const float4 *myn = &nb[dir][0][(ly + R) * HP + (lx + R)];

    __syncthreads();
    for (int c0 = 0; c0 < p.C; c0 += 8) {
        // ---- gather + blend the warped halo chunk into LDS ----
#pragma unroll 1
        for (int i = tid; i < 2 * 2 * NHALO; i += 256) {
            const int k4 = i & 1;
            const int rest = i >> 1;
            const int map = rest / NHALO, hp = rest - map * NHALO;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const float4 wg = samp_w[map][hp];
            const SampIdx si = samp_i[map][hp];
            const float *src = nbr[map] + (size_t)si.idx * p.pix_stride + c0 + 4 * k4;
            const float4 tl = *reinterpret_cast<const float4 *>(src);
            const int dx = si.flags & 1, dy = (si.flags & 2) ? p.w : 0;
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
        // reference pixel chunk (address is clamped to a valid pixel for out-of-image lanes)
        const float4 r01[2] = {*reinterpret_cast<const float4 *>(refp + c0), *reinterpret_cast<const float4 *>(refp + c0 + 4)};
        __syncthreads();
        // ---- correlate ----
#pragma unroll
        for (int k4 = 0; k4 < 2; ++k4) {
            const float4 r = r01[k4];
            const float4 *base = myn + k4 * (HH * HP);
#pragma unroll
            for (int g = 0; g < 9; ++g) {          // one qx column (9 consecutive channels) at a time
                float4 v[9];
#pragma unroll
                for (int j = 0; j < 9; ++j) v[j] = base[-(j - 4) * HP - (g - 4)];
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    float a = acc[g * 9 + j];
                    a = fmaf(r.x, v[j].x, a);
                    a = fmaf(r.y, v[j].y, a);
                    a = fmaf(r.z, v[j].z, a);
                    a = fmaf(r.w, v[j].w, a);
                    acc[g * 9 + j] = a;
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting all 81 reads (spills)
            }
        }
        __syncthreads();
    }

    if (!pvalid) return;
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

hipError_t launch_warp_costvol(const CorrLaunch &p, hipStream_t s)
{
    if (p.C % 8 != 0 || p.pix_stride % 4 != 0) return hipErrorInvalidValue;
    const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
    dim3 grid((unsigned)(tiles_x * tiles_y * p.B));
    const bool pow2 = (p.C & (p.C - 1)) == 0;
    if (pow2) hipLaunchKernelGGL(warp_costvol_kernel<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(warp_costvol_kernel<false>, grid, dim3(256), 0, s, p);
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
