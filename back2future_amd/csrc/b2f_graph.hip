// Generic executor for the graph shapes of createModelMulti (models/pwc.lua:87-508) other than the shipped one:
// any odd search window (pwc_ws), 2..7 levels, pwc_skip >= 0, two_frame, pwc_sum_cvs, residual, occ_input,
// rescale_flow, flownet_factor, pwc_siamese (createModelMulti(nil) itself is win 5 / levels 4, pwc.lua:88).
// pwc_skip = 0 (pwc.lua:120-122,171-173,359,423-429,462-471): a stride-1 level-1 unit on the image, decoders down to full
// resolution, the level's own flow / occlusion maps as outputs.  pwc_siamese = 0 (pwc.lua:125-127,175,182): the
// average-pooled 3-channel image in place of the learned pyramid (8-float pixel records with five zeros = one chunk).
// The shipped graph keeps its fused fast path (b2f_api.hip:forward_impl); here every node of the Lua graph is one or
// a few straightforward kernels -- feature warps materialised (warpingUnit, pwc.lua:68-73), cost volumes for any
// window, JoinTable as channel copies into a zero-padded chunk-planar decoder input -- and the convolutions run on the
// same MFMA / Winograd kernels through the packed-weight table.  Correctness path, not a tuned one: parity with the
// oracle's orc_pwc_forward_ex is what the tests hold it to.
#include "b2f_ctx.h"

#include <cmath>

namespace b2f {
namespace {


// element (b, c, pix) of a chunk-planar tensor with `chunks` 8-channel planes per image
__device__ __forceinline__ size_t cp8_at(int b, int chunks, size_t hw, int c, size_t pix)
{
    return (((size_t)b * chunks + (c >> 3)) * hw + pix) * 8 + (c & 7);
}

// nn.JoinTable(2) piece: C channels of src -> channels c_off .. c_off + C - 1 of a chunk-planar destination.
// kind 0: src chunk-planar with src_chunks planes; 1: packed B x hw x 2 (flows); 2: planar B x C x hw
__global__ void put_channels_kernel(const float *src, int kind, int src_chunks, int C, int B, size_t hw, float *dst, int dst_chunks, int c_off)
{
    const size_t n = (size_t)B * hw * C;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const size_t r = i / C;
    const size_t pix = r % hw;
    const int b = (int)(r / hw);
    float v;
    if (kind == 0) v = src[cp8_at(b, src_chunks, hw, c, pix)];
    else if (kind == 1) v = src[((size_t)b * hw + pix) * C + c];
    else v = src[((size_t)b * C + c) * hw + pix];
    dst[cp8_at(b, dst_chunks, hw, c_off + c, pix)] = v;
}

// nn.CostVolMulti(win, true){ref, frmF} and (win, false){ref, frmB} (CostVolMulti.lua:49-109) on chunk-planar maps:
// channel d = (qx + n) * win + (qy + n); fwd: frmF[y - qy, x - qx], bwd: frmB[y + qy, x + qx]; out of range -> 0; / C.
// dstJ: JoinTable{fwd, bwd} (bwd at channel nd; frmB may be null: two_frame).  dstS (optional): CAddTable{fwd, bwd}.
__global__ void costvol_cp8_kernel(const float *ref, const float *frmF, const float *frmB, int chunks, int C, int B, int h, int w, int win,
                                   float *dstJ, int chunksJ, float *dstS, int chunksS)
{
    const int nd = win * win;
    const size_t hw = (size_t)h * w;
    const size_t n = (size_t)B * hw * nd;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % nd);
    const size_t r = i / nd;
    const size_t pix = r % hw;
    const int b = (int)(r / hw);
    const int y = (int)(pix / w), x = (int)(pix % w);
    const int nn = (win - 1) / 2;
    const int qx = d / win - nn, qy = d % win - nn;
    float vf = 0.f, vb = 0.f;
    const float cf = (float)C;
    {
        const int yy = y - qy, xx = x - qx;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            float a = 0.f;
            for (int k = 0; k < C; ++k) a = fmaf(ref[cp8_at(b, chunks, hw, k, pix)], frmF[cp8_at(b, chunks, hw, k, (size_t)yy * w + xx)], a);
            vf = a;
        }
        vf = vf / cf;
    }
    if (frmB) {
        const int yy = y + qy, xx = x + qx;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            float a = 0.f;
            for (int k = 0; k < C; ++k) a = fmaf(ref[cp8_at(b, chunks, hw, k, pix)], frmB[cp8_at(b, chunks, hw, k, (size_t)yy * w + xx)], a);
            vb = a;
        }
        vb = vb / cf;
    }
    dstJ[cp8_at(b, chunksJ, hw, d, pix)] = vf;
    if (frmB) dstJ[cp8_at(b, chunksJ, hw, nd + d, pix)] = vb;
    if (dstS) dstS[cp8_at(b, chunksS, hw, d, pix)] = vf + vb;
}

// warpingUnit(I, F * k) (pwc.lua:68-73,393-409; sampler: BilinearSamplerBHWD.cu:41-115) on a chunk-planar map,
// flow packed B x hw x 2
__global__ void warp_cp8_kernel(const float *img, int chunks, const float *flow, float k, int B, int h, int w, float *out)
{
    const size_t hw = (size_t)h * w;
    const size_t n = (size_t)B * chunks * hw * 8;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c8 = (int)(i & 7);
    size_t r = i >> 3;
    const size_t pix = r % hw;
    r /= hw;
    const int ch = (int)(r % chunks);
    const int b = (int)(r / chunks);
    const int y = (int)(pix / w), x = (int)(pix % w);
    const float2 g = *reinterpret_cast<const float2 *>(flow + ((size_t)b * hw + pix) * 2);
    int xl, yt;
    float wx, wy;
    bhwd_top_left(g.x * k + (float)x, w, xl, wx);
    bhwd_top_left(g.y * k + (float)y, h, yt, wy);
    const float *src = img + (((size_t)b * chunks + ch) * hw + (size_t)yt * w + xl) * 8 + c8;
    const bool x1 = xl + 1 <= w - 1, y1 = yt + 1 <= h - 1;
    const float tl = src[0];
    const float tr = x1 ? src[8] : 0.f;
    const float bl = y1 ? src[(size_t)w * 8] : 0.f;
    const float br = (x1 && y1) ? src[(size_t)(w + 1) * 8] : 0.f;
    out[i] = wx * wy * tl + (1.f - wx) * wy * tr + wx * (1.f - wy) * bl + (1.f - wx) * (1.f - wy) * br;
}

// residual flow, nn.CAddTable{decoder output, ufs[l+1]} (pwc.lua:341-351): fs is the 2-channel decoder output inside an
// 8-float chunk record, u packed B x hw x 2
__global__ void add_flow_kernel(float *fs, const float *u, size_t npix)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * 2) return;
    fs[(i >> 1) * 8 + (i & 1)] = fs[(i >> 1) * 8 + (i & 1)] + u[i];
}

// out = in * k elementwise (nn.MulConstant(2.0) of rescale_flow, pwc.lua:364-369,382-388)
__global__ void scale_kernel(float *p, size_t n, float k)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * k;
}

// nn.SpatialSoftMax over the 2 logits (first two floats of an 8-float record) + log2(f) x SpatialUpSamplingNearest(2)
// (pwc.lua:308-321) -> planar B x 2 x f h x f w
__global__ void softmax_nearest_kernel(const float *logits, int B, int h, int w, int f, float *out)
{
    const int Hf = f * h, Wf = f * w;
    const size_t n = (size_t)B * Hf * Wf;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int X = (int)(i % Wf);
    const size_t r = i / Wf;
    const int Y = (int)(r % Hf);
    const int b = (int)(r / Hf);
    const float2 z = *reinterpret_cast<const float2 *>(logits + (((size_t)b * h + Y / f) * w + X / f) * 8);
    const float m = fmaxf(z.x, z.y);
    const float e0 = expf(z.x - m), e1 = expf(z.y - m);
    const float sum = e0 + e1;
    const size_t hwf = (size_t)Hf * Wf;
    out[((size_t)b * 2) * hwf + (size_t)Y * Wf + X] = e0 / sum;
    out[((size_t)b * 2 + 1) * hwf + (size_t)Y * Wf + X] = e1 / sum;
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

struct Bump {
    float *base;
    size_t off = 0;
    float *take(size_t n)
    {
        float *p = base ? base + off : nullptr;
        off += (n + 63) & ~(size_t)63;
        return p;
    }
};

}  // namespace

// model:forward for a non-shipped graph.  outs: the output table in order (pwc.lua:459-489), planar device buffers, all
// present.  dry: only size the workspace (returns floats needed in *need).
static int graph_run(b2f_ctx *c, hipStream_t s, bool cap, const void *dev_in, int in_kind, int B, int H, int W, float *const *outs,
                     bool dry, size_t *need)
{
    const GraphOpts &g = c->g;
    const int L = g.levels, LST = g.l_st(), nd = g.nd();
    const bool past = g.past_flow;
    const int unit = in_kind == B2F_IN_UNIT;
    c->cur_batch = c->req_batch > 0 ? c->req_batch : B;   // run_conv's kernel rule reads it (as forward_impl sets it for the shipped graph)
    Bump A{dry ? nullptr : c->arena};
    int hh[8], ww[8];
    for (int l = 1; l <= 7; ++l) { hh[l] = H >> (l - 1); ww[l] = W >> (l - 1); }
#define GK(...) do { if (!dry) { hipLaunchKernelGGL(__VA_ARGS__); HIPCHK(hipGetLastError()); } } while (0)
#define GL(expr) do { if (!dry) HIPCHK(expr); } while (0)

    // packed frames + image pyramid ds[f][k], f in {1, 3} (pwc.lua:148-158): NHWC8, [frame][B][h][w][8]
    float *img = A.take((size_t)3 * B * H * W * kImgC);
    GL(launch_pack_input((const float *)dev_in, unit, B, H, W, img, s));
    float *ds[8] = {nullptr};
    for (int k = 2; k <= L - LST + 1; ++k) {
        ds[k] = A.take((size_t)2 * B * (H >> (k - 1)) * (W >> (k - 1)) * kImgC);
        const int hk = H >> (k - 2), wk = W >> (k - 2);
        for (int f = 0; f < 2; ++f) {
            const float *src = (k == 2) ? img + (size_t)(f == 0 ? 0 : 2) * B * H * W * kImgC : ds[k - 1] + (size_t)f * B * hk * wk * kImgC;
            GL(launch_avgpool2_nhwc(src, B, hk, wk, kImgC, ds[k] + (size_t)f * B * (hk / 2) * (wk / 2) * kImgC, s));
        }
    }
    // feature pyramid, three frames batched (pwc.lua:169-211); FS(l) = floats per pixel of a level's chunk-planar map
    auto FS = [&](int l) { return (g.feat(l) + 7) / 8 * 8; };
    float *cs[8] = {nullptr};
    if (!g.siamese) {   // pwc.lua:175,182: nn.Identity / nn.SpatialAveragePooling(2,2,2,2) of the image
        cs[1] = img;
        for (int l = 2; l <= L; ++l) {
            cs[l] = A.take((size_t)3 * B * hh[l] * ww[l] * kImgC);
            GL(launch_avgpool2_nhwc(cs[l - 1], 3 * B, hh[l - 1], ww[l - 1], kImgC, cs[l], s));
        }
    } else {
        const int l0 = g.feat_first();
        float *tmp = A.take((size_t)3 * B * hh[l0] * ww[l0] * FS(l0));
        for (int l = l0; l <= L; ++l) {
            cs[l] = A.take((size_t)3 * B * hh[l] * ww[l] * FS(l));
            if (dry) continue;
            if (l == 1) {            // convUnit(3, featMaps[1], 1) on the packed frames (pwc.lua:171-173)
                const ConvSeg in1 = cp8_seg(img, 3, (size_t)H * W);
                CHK(run_conv_layer(c, s, cap, find_conv_id(c, KIND_FEAT, 1, 1), &in1, 3 * B, H, W, 1, 1, tmp));
            } else if (l == 2 && l0 == 2) {
                HIPCHK(launch_conv_first((const float *)dev_in, unit, B, H, W, c->wpk_dev + c->first_w_off, c->wpk_dev + c->first_b_off, tmp, s));
            } else {
                const ConvSeg in1 = cp8_seg(cs[l - 1], g.feat(l - 1), (size_t)hh[l - 1] * ww[l - 1]);
                CHK(run_conv_layer(c, s, cap, find_conv_id(c, KIND_FEAT, l, 1), &in1, 3 * B, hh[l - 1], ww[l - 1], 2, 1, tmp));
            }
            const ConvSeg in2 = cp8_seg(tmp, g.feat(l), (size_t)hh[l] * ww[l]);
            CHK(run_conv_layer(c, s, cap, find_conv_id(c, KIND_FEAT, l, 2), &in2, 3 * B, hh[l], ww[l], 1, 1, cs[l]));
        }
    }
    auto frame = [&](int l, int f) { return cs[l] ? cs[l] + (size_t)(f - 1) * B * hh[l] * ww[l] * FS(l) : nullptr; };

    float *ws[4][8] = {{nullptr}};          // warped features, chunk-planar
    float *ufs[9] = {nullptr}, *ubfs[9] = {nullptr};   // packed B x (2h x 2w) x 2, index = the level they come from
    float *uoccs[9] = {nullptr};            // planar B x 2 x 2h x 2w
    const int per = past ? 5 : 4;
    auto dec_buf = [&](size_t px, int i) { return A.take(px * kDec[i]); };

    for (int l = L; l >= LST; --l) {   // pwc.lua:237
        const int h = hh[l], w = ww[l], C = g.feat(l), chunksC = (C + 7) / 8;
        const size_t hw = (size_t)h * w, px = (size_t)B * hw;
        const float *ref = frame(l, 2);
        const float *in_fut = (l == L) ? frame(l, 3) : ws[3][l];
        const float *in_past = g.two_frame ? nullptr : ((l == L) ? frame(l, 1) : ws[1][l]);
        // cost volumes (pwc.lua:246-285)
        const int ndo = g.nd_occ(), chJ = (ndo + 7) / 8, chS = (nd + 7) / 8;
        float *cvJ = A.take(px * chJ * 8), *cvS = g.sum_cvs && !g.two_frame ? A.take(px * chS * 8) : nullptr;
        GL(hipMemsetAsync(cvJ, 0, px * chJ * 8 * sizeof(float), s));
        if (cvS) GL(hipMemsetAsync(cvS, 0, px * chS * 8 * sizeof(float), s));
        GK(costvol_cp8_kernel, dim3(nblk(px * nd)), dim3(256), 0, s, ref, in_fut, in_past, chunksC, C, B, h, w, g.win, cvJ, chJ, cvS, chS);
        const float *cv_flow = cvS ? cvS : cvJ;
        const int ndf = g.nd_flow(), chF = (ndf + 7) / 8;

        float *d[6];
        for (int i = 1; i <= 5; ++i) d[i] = dec_buf(px, i);
        auto decoder = [&](int kind, const float *din, int n, float *out2) -> int {
            if (dry) return 0;
            const ConvSeg in0 = cp8_seg(din, n, hw);
            CHK(run_conv_layer(c, s, cap, find_conv_id(c, kind, l, 1), &in0, B, h, w, 1, 1, d[1]));
            for (int i = 2; i <= 6; ++i) {
                const ConvSeg in = cp8_seg(d[i - 1], kDec[i - 1], hw);
                CHK(run_conv_layer(c, s, cap, find_conv_id(c, kind, l, i), &in, B, h, w, 1, i < 6, i == 6 ? out2 : d[i]));
            }
            return 0;
        };
        auto put = [&](const float *src, int kind, int src_chunks, int Cn, float *dst, int dst_chunks, int c_off) -> int {
            GK(put_channels_kernel, dim3(nblk(px * Cn)), dim3(256), 0, s, src, kind, src_chunks, Cn, B, hw, dst, dst_chunks, c_off);
            return 0;
        };

        // occlusion decoder + SpatialSoftMax + nearest upsampling (pwc.lua:288-321)
        {
            const int n = g.occ_in(l), chn = (n + 7) / 8;
            float *din = A.take(px * chn * 8), *logits = A.take(px * 8);
            GL(hipMemsetAsync(din, 0, px * chn * 8 * sizeof(float), s));
            int off = 0;
            CHK(put(cvJ, 0, chJ, ndo, din, chn, off)); off += ndo;
            CHK(put(ref, 0, chunksC, C, din, chn, off)); off += C;
            if (g.two_frame) { CHK(put(frame(l, 3), 0, chunksC, C, din, chn, off)); off += C; }   // cs[ref+1][l], pwc.lua:292-296
            if (l != L) {
                CHK(put(ufs[l + 1], 1, 0, 2, din, chn, off)); off += 2;
                if (g.occ_input) { CHK(put(uoccs[l + 1], 2, 0, 2, din, chn, off)); off += 2; }
            }
            CHK(decoder(KIND_OCC, din, n, logits));
            if (g.occ_input) {
                uoccs[l] = A.take(px * 4 * 2);
                GK(softmax_nearest_kernel, dim3(nblk(px * 4)), dim3(256), 0, s, logits, B, h, w, 2, uoccs[l]);
            }
            const int f = 1 << g.skip;   // skip_occs = skip x nearest x2; pwc_skip = 0: occs[l] itself (pwc.lua:468-470)
            float *o = dry ? nullptr : outs[(l - LST) * per + (past ? 2 : 1)];
            GK(softmax_nearest_kernel, dim3(nblk(px * f * f)), dim3(256), 0, s, logits, B, h, w, f, o);
        }
        // flow decoders (pwc.lua:325-352)
        float *fs = A.take(px * 8), *bfs = past ? A.take(px * 8) : nullptr;
        for (int pass = 0; pass < (past ? 2 : 1); ++pass) {
            const int kind = pass ? KIND_PAST : KIND_FLOW;
            float *o2 = pass ? bfs : fs;
            const float *u = pass ? ubfs[l + 1] : ufs[l + 1];
            if (l == L) {
                CHK(decoder(kind, cv_flow, ndf, o2));
            } else {
                const int n = g.flow_in(l), chn = (n + 7) / 8;
                float *din = A.take(px * chn * 8);
                GL(hipMemsetAsync(din, 0, px * chn * 8 * sizeof(float), s));
                CHK(put(cv_flow, 0, chF, ndf, din, chn, 0));
                CHK(put(ref, 0, chunksC, C, din, chn, ndf));
                CHK(put(u, 1, 0, 2, din, chn, ndf + C));
                CHK(decoder(kind, din, n, o2));
                if (g.residual) GK(add_flow_kernel, dim3(nblk(px * 2)), dim3(256), 0, s, o2, u, px);
            }
        }
        // upsampling (pwc.lua:359-390): ufs = bilinear x2 [x 2.0]; skip_ufs = l_st - 2 more of the same
        for (int pass = 0; pass < (past ? 2 : 1); ++pass) {
            float *src = pass ? bfs : fs;
            if (g.skip == 0) {   // pwc.lua:462-466: fs[l] / bfs[l] are the outputs; ufs only feeds the next level (:359)
                float *o = dry ? nullptr : outs[(l - LST) * per + pass];
                GL(launch_nhwc_to_planar(src, 8, 2, B, h, w, o, s));
                if (l == LST) continue;
            }
            float *u = A.take(px * 4 * 2);
            GL(launch_upsample_flow2x(src, 8, B, h, w, u, s));
            if (g.rescale_flow) GK(scale_kernel, dim3(nblk(px * 8)), dim3(256), 0, s, u, px * 8, 2.0f);
            (pass ? ubfs : ufs)[l] = u;
            float *cur = u;
            int ch_ = 2 * h, cw_ = 2 * w;
            for (int i = 2; i <= LST - 1; ++i) {
                float *nx = A.take((size_t)B * ch_ * cw_ * 4 * 2);
                GL(launch_upsample_flow2x(cur, 2, B, ch_, cw_, nx, s));
                if (g.rescale_flow) GK(scale_kernel, dim3(nblk((size_t)B * ch_ * cw_ * 8)), dim3(256), 0, s, nx, (size_t)B * ch_ * cw_ * 8, 2.0f);
                cur = nx; ch_ *= 2; cw_ *= 2;
            }
            if (g.skip == 0) continue;
            float *o = dry ? nullptr : outs[(l - LST) * per + pass];
            GL(launch_nhwc_to_planar(cur, 2, 2, B, ch_, cw_, o, s));
        }
        // warps (pwc.lua:393-446)
        const int f_i = g.two_frame ? 2 : 1, l_i = 3;
        for (int f = 1; f <= 3; f += 2) {
            if (l > LST && f >= f_i && f <= l_i) {
                const float k = g.rescale_flow ? (float)((double)g.flownet_factor * (f - 2)) : (float)((double)g.flownet_factor * (f - 2) / std::pow(2.0, l - 2));
                const int Cn = FS(l - 1);
                ws[f][l - 1] = A.take(px * 4 * Cn);
                GK(warp_cp8_kernel, dim3(nblk(px * 4 * Cn)), dim3(256), 0, s, frame(l - 1, f), Cn / 8, ufs[l], k, B, 2 * h, 2 * w, ws[f][l - 1]);
            }
            // image warps iws[f][l] (pwc.lua:410-446)
            const int kimg = l - LST + 1, uh = H >> (l - LST), uw = W >> (l - LST);
            const size_t img_f = (size_t)B * uh * uw * kImgC;
            const float *im = (kimg == 1) ? (img ? img + (size_t)(f == 1 ? 0 : 2) * img_f : nullptr) : (ds[kimg] ? ds[kimg] + (size_t)(f == 1 ? 0 : 1) * img_f : nullptr);
            const float k2 = g.rescale_flow ? (float)((double)g.flownet_factor * (f - 2)) : (float)((double)g.flownet_factor * (f - 2) / std::pow(2.0, l - LST));
            if (!dry) {
                const float *flowp = (past && f < 2) ? outs[(l - LST) * per + 1] : outs[(l - LST) * per];
                float *o = outs[(l - LST) * per + per - 2 + (f == 1 ? 0 : 1)];
                HIPCHK(launch_warp_image_planar(im, flowp, k2, B, uh, uw, o, s));
            }
        }
    }
#undef GK
#undef GL
    if (need) *need = A.off;
    return 0;
}

int graph_forward(b2f_ctx *c, hipStream_t s, bool cap, const void *dev_in, int in_kind, int B, int H, int W, float *const *outs)
{
    size_t need = 0;
    CHK(graph_run(c, s, cap, dev_in, in_kind, B, H, W, outs, true, &need));
    CHK(ensure_arena(c, need));
    return graph_run(c, s, cap, dev_in, in_kind, B, H, W, outs, false, nullptr);
}

}  // namespace b2f
