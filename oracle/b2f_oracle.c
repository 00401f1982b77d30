/*
 * b2f_oracle.c -- CPU ORACLE for the back2future computeFlow hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The shipped path (back2future_amd/csrc, libb2f.so) never links,
 * imports or falls back to anything in oracle/.
 *
 * PARITY UNPINNED: the reference holds no golden vectors, known-answer tests
 * or recorded outputs for this path, Torch7/LuaJIT are not
 * installed, and the pretrained .t7 files are not in the tree.  What pins this
 * restatement instead: (1) it follows the reference text line by line (cited
 * below), (2) tests/test_oracle_vs_torch.py cross-checks every op against
 * PyTorch-CPU as an independent implementation of the same THNN-lineage ops,
 * (3) known-answer tests derived from the reference text (impulse test of
 * CostVolMulti.lua:225-254, zero-flow warp = identity, border clamp).
 *
 * Conventions: all tensors fp32, layouts as in the reference (B x C x H x W,
 * "BDHW"), except the sampler which is BHWD exactly like the module it restates.
 * [3P] marks Torch7 package semantics (nn/cunn/cudnn/image) that are not
 * readable under /root/reference and are restated from the published algorithm.
 *
 * Build: gcc -O3 -march=native -ffp-contract=off -fopenmp -shared -fPIC
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* transforms.lua:33-45  ColorNormalize: for every RGB triple c and colour i:
 * img[3c+i] = (img[3c+i] + (-mean[i])) / std[i]   (add then div, on a clone).
 * back2future.lua:33-36 gives mean/std.  Arithmetic is done in the tensor's
 * own type; image.load yields float tensors, so fp32 here.                   */
ORC_API void orc_color_normalize(float *img, int nch, int H, int W)
{
    static const float mean[3] = {0.485f, 0.456f, 0.406f};
    static const float std_[3] = {0.229f, 0.224f, 0.225f};
    const long hw = (long)H * W;
    for (int c = 0; c < nch; ++c) {
        const float m = -mean[c % 3], s = std_[c % 3];
        float *p = img + (long)c * hw;
        for (long i = 0; i < hw; ++i) p[i] = (p[i] + m) / s;
    }
}

/* back2future.lua:54-67: round each side down to a multiple of 64. */
ORC_API int orc_fine_size(int n) { return (n % 64 == 0) ? n : n - (n % 64); }

/* ------------------------------------------------------------------------- */
/* image.scale(src, W, H) default mode 'bilinear' [3P: torch/image
 * generic/image.c Main_scaleBilinear / scaleLinear_rowcol].  Separable: the
 * width pass writes a (C, Hs, Wd) temporary, then the height pass.  Per line:
 * up-scaling = align-corners lerp with the last sample copied; down-scaling =
 * running fractional box average; equal length = copy.  The intermediates
 * (scale, fractions, accumulator) are C floats in that package.  Called at
 * back2future.lua:71.                                                        */
static void scale_line(const float *src, long sstride, long slen,
                       float *dst, long dstride, long dlen)
{
    if (dlen > slen) {
        const float scale = (float)(slen - 1) / (float)(dlen - 1);
        if (slen == 1) {
            for (long di = 0; di < dlen - 1; ++di) dst[di * dstride] = src[0];
        } else {
            for (long di = 0; di < dlen - 1; ++di) {
                float si_f = di * scale;
                long si_i = (long)si_f;
                si_f -= si_i;
                dst[di * dstride] = (1 - si_f) * src[si_i * sstride] +
                                    si_f * src[(si_i + 1) * sstride];
            }
        }
        dst[(dlen - 1) * dstride] = src[(slen - 1) * sstride];
    } else if (dlen < slen) {
        long si0_i = 0;
        float si0_f = 0;
        const float scale = (float)slen / (float)dlen;
        for (long di = 0; di < dlen; ++di) {
            float si1_f = (di + 1) * scale;
            long si1_i = (long)si1_f;
            si1_f -= si1_i;
            float acc = (1 - si0_f) * src[si0_i * sstride];
            float n = 1 - si0_f;
            for (long si = si0_i + 1; si < si1_i; ++si) {
                acc += src[si * sstride];
                n += 1;
            }
            if (si1_i < slen) {
                acc += si1_f * src[si1_i * sstride];
                n += si1_f;
            }
            dst[di * dstride] = acc / n;
            si0_i = si1_i;
            si0_f = si1_f;
        }
    } else {
        for (long i = 0; i < dlen; ++i) dst[i * dstride] = src[i * sstride];
    }
}

ORC_API void orc_image_scale_bilinear(const float *src, int C, int Hs, int Ws,
                                      float *dst, int Hd, int Wd)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)C * Hs * Wd);
    for (int c = 0; c < C; ++c) {
        for (int y = 0; y < Hs; ++y)
            scale_line(src + ((long)c * Hs + y) * Ws, 1, Ws,
                       tmp + ((long)c * Hs + y) * Wd, 1, Wd);
        for (int x = 0; x < Wd; ++x)
            scale_line(tmp + (long)c * Hs * Wd + x, Wd, Hs,
                       dst + (long)c * Hd * Wd + x, Wd, Hd);
    }
    free(tmp);
}

/* image.scale(src, W, H, 'simple') [3P: Main_scaleSimple]: nearest with a
 * float ratio, ii = (long)(i * (Ws/Wd)), clamped.  back2future.lua:82,89,91.
 * Double in/out because computeFlow converts est to double first (:77,:87).  */
ORC_API void orc_image_scale_simple_f64(const double *src, int C, int Hs, int Ws,
                                        double *dst, int Hd, int Wd)
{
    const float scx = (float)Ws / (float)Wd, scy = (float)Hs / (float)Hd;
    for (int c = 0; c < C; ++c)
        for (int j = 0; j < Hd; ++j)
            for (int i = 0; i < Wd; ++i) {
                long ii = (long)((float)i * scx), jj = (long)((float)j * scy);
                if (ii > Ws - 1) ii = Ws - 1;
                if (jj > Hs - 1) jj = Hs - 1;
                dst[((long)c * Hd + j) * Wd + i] = src[((long)c * Hs + jj) * Ws + ii];
            }
}

ORC_API void orc_image_scale_simple_u8(const uint8_t *src, int C, int Hs, int Ws,
                                       uint8_t *dst, int Hd, int Wd)
{
    const float scx = (float)Ws / (float)Wd, scy = (float)Hs / (float)Hd;
    for (int c = 0; c < C; ++c)
        for (int j = 0; j < Hd; ++j)
            for (int i = 0; i < Wd; ++i) {
                long ii = (long)((float)i * scx), jj = (long)((float)j * scy);
                if (ii > Ws - 1) ii = Ws - 1;
                if (jj > Hs - 1) jj = Hs - 1;
                dst[((long)c * Hd + j) * Wd + i] = src[((long)c * Hs + jj) * Ws + ii];
            }
}

/* ------------------------------------------------------------------------- */
/* nn.SpatialConvolution(Ci,Co,3,3,s,s,1,1) [3P]: cross-correlation, zero pad 1,
 * weight Co x Ci x 3 x 3, Hout = floor((H+2-3)/s)+1.  pwc.lua:60,62,78-83.
 * Optional fused nn.LeakyReLU(0.2,true) (pwc.lua:61,63,78-82).               */
ORC_API void orc_conv3x3(const float *x, int B, int Ci, int H, int W,
                         const float *w, const float *bias, int Co, int stride,
                         int leaky, float *y)
{
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Co; ++co) {
            float *yo = y + ((long)b * Co + co) * Ho * Wo;
            for (long i = 0; i < (long)Ho * Wo; ++i) yo[i] = bias[co];
            for (int ci = 0; ci < Ci; ++ci) {
                const float *xi = x + ((long)b * Ci + ci) * H * W;
                const float *wk = w + ((long)co * Ci + ci) * 9;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const float wv = wk[ky * 3 + kx];
                        /* valid output range: 0 <= s*Y+ky-1 < H */
                        int Y0 = 0, Y1 = Ho, X0 = 0, X1 = Wo;
                        while (Y0 < Ho && stride * Y0 + ky - 1 < 0) ++Y0;
                        while (Y1 > Y0 && stride * (Y1 - 1) + ky - 1 >= H) --Y1;
                        while (X0 < Wo && stride * X0 + kx - 1 < 0) ++X0;
                        while (X1 > X0 && stride * (X1 - 1) + kx - 1 >= W) --X1;
                        for (int Y = Y0; Y < Y1; ++Y) {
                            const float *xr = xi + (long)(stride * Y + ky - 1) * W + (kx - 1);
                            float *yr = yo + (long)Y * Wo;
                            if (stride == 1)
                                for (int X = X0; X < X1; ++X) yr[X] += wv * xr[X];
                            else
                                for (int X = X0; X < X1; ++X) yr[X] += wv * xr[stride * X];
                        }
                    }
            }
            if (leaky)
                for (long i = 0; i < (long)Ho * Wo; ++i)
                    yo[i] = yo[i] > 0 ? yo[i] : 0.2f * yo[i];
        }
}

/* nn.SpatialAveragePooling(2,2,2,2) [3P]: mean of non-overlapping 2x2. pwc.lua:155 */
ORC_API void orc_avgpool2(const float *x, int BC, int H, int W, float *y)
{
    const int Ho = H / 2, Wo = W / 2;
    for (int c = 0; c < BC; ++c)
        for (int Y = 0; Y < Ho; ++Y)
            for (int X = 0; X < Wo; ++X) {
                const float *p = x + ((long)c * H + 2 * Y) * W + 2 * X;
                y[((long)c * Ho + Y) * Wo + X] = (p[0] + p[1] + p[W] + p[W + 1]) / 4.0f;
            }
}

/* nn.SpatialUpSamplingBilinear(2.0) [3P: THNN/THCUNN SpatialUpSamplingBilinear]:
 * output 2h x 2w, align-corners ratios r = (in-1)/(out-1) in float;
 * h1r = r*h2; h1 = (int)h1r; h1p = h1 < in-1; lambda = h1r - h1.  pwc.lua:360-381 */
ORC_API void orc_upsample_bilinear2x(const float *x, int BC, int h, int w, float *y)
{
    const int H2 = 2 * h, W2 = 2 * w;
    const float rh = (H2 > 1) ? (float)(h - 1) / (float)(H2 - 1) : 0.f;
    const float rw = (W2 > 1) ? (float)(w - 1) / (float)(W2 - 1) : 0.f;
    for (int c = 0; c < BC; ++c) {
        const float *xi = x + (long)c * h * w;
        float *yo = y + (long)c * H2 * W2;
        for (int h2 = 0; h2 < H2; ++h2) {
            const float h1r = rh * h2;
            const int h1 = (int)h1r;
            const int h1p = (h1 < h - 1) ? 1 : 0;
            const float h1l = h1r - h1, h0l = 1.f - h1l;
            for (int w2 = 0; w2 < W2; ++w2) {
                const float w1r = rw * w2;
                const int w1 = (int)w1r;
                const int w1p = (w1 < w - 1) ? 1 : 0;
                const float w1l = w1r - w1, w0l = 1.f - w1l;
                const float *p = xi + (long)h1 * w + w1;
                yo[(long)h2 * W2 + w2] =
                    h0l * (w0l * p[0] + w1l * p[w1p]) +
                    h1l * (w0l * p[(long)h1p * w] + w1l * p[(long)h1p * w + w1p]);
            }
        }
    }
}

/* nn.SpatialUpSamplingNearest(2.0) [3P]: out[Y][X] = in[Y/2][X/2]. pwc.lua:312,319 */
ORC_API void orc_upsample_nearest2x(const float *x, int BC, int h, int w, float *y)
{
    for (int c = 0; c < BC; ++c)
        for (int Y = 0; Y < 2 * h; ++Y)
            for (int X = 0; X < 2 * w; ++X)
                y[((long)c * 2 * h + Y) * 2 * w + X] = x[((long)c * h + Y / 2) * w + X / 2];
}

/* nn.SpatialSoftMax [3P]: softmax over the channel dim at every pixel
 * (max-subtracted form).  pwc.lua:308.                                       */
ORC_API void orc_spatial_softmax(const float *x, int B, int C, int h, int w, float *y)
{
    const long hw = (long)h * w;
    for (int b = 0; b < B; ++b)
        for (long i = 0; i < hw; ++i) {
            const float *p = x + (long)b * C * hw + i;
            float *q = y + (long)b * C * hw + i;
            float m = p[0];
            for (int c = 1; c < C; ++c) m = p[c * hw] > m ? p[c * hw] : m;
            float s = 0;
            for (int c = 0; c < C; ++c) { q[c * hw] = expf(p[c * hw] - m); s += q[c * hw]; }
            for (int c = 0; c < C; ++c) q[c * hw] = q[c * hw] / s;
        }
}

/* ------------------------------------------------------------------------- */
/* nn.CostVolMulti(win, fwd):updateOutput  --  models/CostVolMulti.lua:49-109,
 * restated loop for loop: output zeroed (:59); q_x_ outer / q_y_ inner with the
 * running channel index i (:66-67,:92); shifts scaled by (f-1) (:68-69);
 * negated when fwd == false (:71-74); the 1-based slice ranges of :76-87
 * (qx/px, qy/py) turned 0-based; cost = cmul(ref[qy,qx], frame[py,px]) summed
 * over channels and ADDED into output[i] on the q-range (:89-90); final
 * division by N*(frames-1) (:100).  frames[0] is the reference map.          */
ORC_API void orc_costvol(const float *const *frames, int nframes, int B, int N,
                         int h, int w, int win, int fwd, float *out)
{
    const int n = (win - 1) / 2;
    const long hw = (long)h * w;
    memset(out, 0, sizeof(float) * (size_t)B * win * win * hw);
    const float *ref = frames[0];
    for (int f = 1; f < nframes; ++f) {
        const float *frame = frames[f];
        int i = 0;
        for (int q_x_ = -n; q_x_ <= n; ++q_x_)
            for (int q_y_ = -n; q_y_ <= n; ++q_y_) {
                int q_x = q_x_ * f, q_y = q_y_ * f; /* (f-1) in 1-based Lua */
                if (!fwd) { q_x = -q_x; q_y = -q_y; }
                /* 0-based inclusive-exclusive ranges */
                int qx0 = q_x, qx1 = w, px0 = 0;
                if (q_x < 0) { qx0 = 0; qx1 = w + q_x; px0 = -q_x; }
                int qy0 = q_y, qy1 = h, py0 = 0;
                if (q_y < 0) { qy0 = 0; qy1 = h + q_y; py0 = -q_y; }
                if (qx1 > qx0 && qy1 > qy0) {
#pragma omp parallel for schedule(static)
                    for (int b = 0; b < B; ++b) {
                        float *o = out + ((long)b * win * win + i) * hw;
                        for (int k = 0; k < N; ++k) {
                            const float *r = ref + ((long)b * N + k) * hw;
                            const float *g = frame + ((long)b * N + k) * hw;
                            for (int y = qy0; y < qy1; ++y) {
                                const float *rr = r + (long)y * w;
                                const float *gg = g + (long)(y - qy0 + py0) * w + (px0 - qx0);
                                float *oo = o + (long)y * w;
                                for (int x = qx0; x < qx1; ++x) oo[x] += rr[x] * gg[x];
                            }
                        }
                    }
                }
                ++i;
            }
    }
    const float div = (float)(N * (nframes - 1));
    for (long j = 0; j < (long)B * win * win * hw; ++j) out[j] = out[j] / div;
}

/* ------------------------------------------------------------------------- */
/* nn.BilinearSamplerBHWD:updateOutput, CUDA semantics --
 * extras/stnbhwd/BilinearSamplerBHWD.cu:6-20 (getTopLeft: xcoord = x + xOut,
 * clamp to [0, width-1], point = floor, weight = 1-(xcoord-point)), :69-70
 * (grid channel 0 = x, 1 = y), :88-91 (a neighbour outside the image
 * contributes 0), :101-104 (the four-term blend).  Output is sized by the grid
 * (BilinearSamplerBHWD.lua:70).  NOT generic/BilinearSamplerBHWD.c (that is the
 * normalized-grid CPU variant the models were not trained with). */
ORC_API void orc_warp_bhwd(const float *img, const float *grid, int B, int ih, int iw,
                           int C, int gh, int gw, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int yOut = 0; yOut < gh; ++yOut)
            for (int xOut = 0; xOut < gw; ++xOut) {
                const float *g = grid + (((long)b * gh + yOut) * gw + xOut) * 2;
                float xc = g[0] + xOut;
                if (xc < 0) xc = 0;
                if (xc > (iw - 1)) xc = iw - 1;
                const int xl = (int)floorf(xc);
                const float xw = 1 - (xc - xl);
                float yc = g[1] + yOut;
                if (yc < 0) yc = 0;
                if (yc > (ih - 1)) yc = ih - 1;
                const int yt = (int)floorf(yc);
                const float yw = 1 - (yc - yt);
                const int xin0 = xl >= 0 && xl <= iw - 1, xin1 = xl + 1 >= 0 && xl + 1 <= iw - 1;
                const int yin0 = yt >= 0 && yt <= ih - 1, yin1 = yt + 1 >= 0 && yt + 1 <= ih - 1;
                const float *tl = img + (((long)b * ih + yt) * iw + xl) * C;
                float *o = out + (((long)b * gh + yOut) * gw + xOut) * C;
                for (int t = 0; t < C; ++t) {
                    const float vtl = (xin0 && yin0) ? tl[t] : 0.f;
                    const float vtr = (xin1 && yin0) ? tl[C + t] : 0.f;
                    const float vbl = (xin0 && yin1) ? tl[(long)iw * C + t] : 0.f;
                    const float vbr = (xin1 && yin1) ? tl[(long)iw * C + C + t] : 0.f;
                    o[t] = xw * yw * vtl + (1 - xw) * yw * vtr + xw * (1 - yw) * vbl +
                           (1 - xw) * (1 - yw) * vbr;
                }
            }
}

/* warpingUnit(I, F) -- pwc.lua:68-73: Transpose BDHW->BHWD on both inputs,
 * sampler, Transpose back.  F has already been through nn.MulConstant(k)
 * (pwc.lua:404,443), which is folded in here as `k` applied in fp32.         */
static void warping_unit(const float *I, const float *F, float k, int B, int C,
                         int h, int w, float *out)
{
    const long hw = (long)h * w;
    float *ib = (float *)malloc(sizeof(float) * (size_t)B * hw * C);
    float *fb = (float *)malloc(sizeof(float) * (size_t)B * hw * 2);
    float *ob = (float *)malloc(sizeof(float) * (size_t)B * hw * C);
    for (int b = 0; b < B; ++b)
        for (long i = 0; i < hw; ++i) {
            for (int c = 0; c < C; ++c) ib[((long)b * hw + i) * C + c] = I[((long)b * C + c) * hw + i];
            for (int c = 0; c < 2; ++c) fb[((long)b * hw + i) * 2 + c] = F[((long)b * 2 + c) * hw + i] * k;
        }
    orc_warp_bhwd(ib, fb, B, h, w, C, h, w, ob);
    for (int b = 0; b < B; ++b)
        for (long i = 0; i < hw; ++i)
            for (int c = 0; c < C; ++c) out[((long)b * C + c) * hw + i] = ob[((long)b * hw + i) * C + c];
    free(ib); free(fb); free(ob);
}

ORC_API void orc_warping_unit(const float *I, const float *F, float k, int B, int C,
                              int h, int w, float *out)
{
    warping_unit(I, F, k, B, C, h, w, out);
}

/* ------------------------------------------------------------------------- */
/* Backward passes (not on the computeFlow path -- training only).
 *
 * nn.BilinearSamplerBHWD:updateGradInput, CUDA kernel backwardBilinearSampling<onlyGrid>
 * (extras/stnbhwd/BilinearSamplerBHWD.cu:161-307; Lua side BilinearSamplerBHWD.lua:81-107
 * zeroes both gradInputs first).  Per output pixel: the four taps' weights from getTopLeft
 * (:199-202); every channel's gradOutput is scattered into gradInputImages with the tap
 * weights (atomicAdd, :236-259; skipped when grad_img == NULL = the onlyGrid instantiation);
 * the four dot products <input tap, gradOutput> are accumulated by 32 threads striding the
 * channels (:231) and summed by the shared-memory tree of sumReduceShMem (:27-36), restated
 * here in the same association; grid gradient :289-290, stored x first (:296-297).
 * The atomic accumulation order of the reference is not defined; this restatement adds in
 * (yOut, xOut, channel) order.                                                            */
ORC_API void orc_warp_bhwd_backward(const float *img, const float *grid, const float *grad_out,
                                    int B, int ih, int iw, int C, int gh, int gw,
                                    float *grad_img, float *grad_grid)
{
    if (grad_img) memset(grad_img, 0, sizeof(float) * (size_t)B * ih * iw * C);
    for (int b = 0; b < B; ++b)
        for (int yOut = 0; yOut < gh; ++yOut)
            for (int xOut = 0; xOut < gw; ++xOut) {
                const float *g = grid + (((long)b * gh + yOut) * gw + xOut) * 2;
                float xc = g[0] + xOut;
                if (xc < 0) xc = 0;
                if (xc > (iw - 1)) xc = iw - 1;
                const int xl = (int)floorf(xc);
                const float xw = 1 - (xc - xl);
                float yc = g[1] + yOut;
                if (yc < 0) yc = 0;
                if (yc > (ih - 1)) yc = ih - 1;
                const int yt = (int)floorf(yc);
                const float yw = 1 - (yc - yt);
                const int tl_in = xl >= 0 && xl <= iw - 1 && yt >= 0 && yt <= ih - 1;
                const int tr_in = xl + 1 >= 0 && xl + 1 <= iw - 1 && yt >= 0 && yt <= ih - 1;
                const int bl_in = xl >= 0 && xl <= iw - 1 && yt + 1 >= 0 && yt + 1 <= ih - 1;
                const int br_in = xl + 1 >= 0 && xl + 1 <= iw - 1 && yt + 1 >= 0 && yt + 1 <= ih - 1;
                const long tl = (((long)b * ih + yt) * iw + xl) * C;
                const long tr = tl + C, bl = tl + (long)iw * C, br = bl + C;
                const float *go = grad_out + (((long)b * gh + yOut) * gw + xOut) * C;
                float part[4][32];
                memset(part, 0, sizeof part);
                for (int tx = 0; tx < 32; ++tx)
                    for (int t = tx; t < C; t += 32) {
                        const float gv = go[t];
                        if (tl_in) { part[0][tx] += img[tl + t] * gv; if (grad_img) grad_img[tl + t] += xw * yw * gv; }
                        if (tr_in) { part[1][tx] += img[tr + t] * gv; if (grad_img) grad_img[tr + t] += (1 - xw) * yw * gv; }
                        if (bl_in) { part[2][tx] += img[bl + t] * gv; if (grad_img) grad_img[bl + t] += xw * (1 - yw) * gv; }
                        if (br_in) { part[3][tx] += img[br + t] * gv; if (grad_img) grad_img[br + t] += (1 - xw) * (1 - yw) * gv; }
                    }
                float dot[4];
                for (int k = 0; k < 4; ++k) {
                    float *sh = part[k];
                    for (int st = 16; st >= 1; st >>= 1)
                        for (int i = 0; i < st; ++i) sh[i] = sh[i] + sh[i + st];
                    dot[k] = sh[0];
                }
                const float yf = -xw * dot[0] + xw * dot[2] - (1 - xw) * dot[1] + (1 - xw) * dot[3];
                const float xf = -yw * dot[0] + yw * dot[1] - (1 - yw) * dot[2] + (1 - yw) * dot[3];
                float *gg = grad_grid + (((long)b * gh + yOut) * gw + xOut) * 2;
                gg[0] = xf;
                gg[1] = yf;
            }
}

/* nn.CostVolMulti:updateGradInput -- models/CostVolMulti.lua:111-181, loop for loop (two
 * input frames, as pwc.lua always passes): gradInputs zeroed (:124-126); for every
 * displacement i (q_x_ outer, q_y_ inner, :135-136), go = gradOutput[:, i, qy, qx] repeated
 * over the N channels (:160-161), gradInputRef[qy, qx] += go .* frame[py, px] (:163),
 * gradInputFrame[py, px] += go .* ref[qy, qx] (:164); both divided by N (frames - 1) (:175-177). */
ORC_API void orc_costvol_backward(const float *ref, const float *frame, const float *grad_out,
                                  int B, int N, int h, int w, int win, int fwd,
                                  float *grad_ref, float *grad_frame)
{
    const int n = (win - 1) / 2;
    const long hw = (long)h * w;
    memset(grad_ref, 0, sizeof(float) * (size_t)B * N * hw);
    memset(grad_frame, 0, sizeof(float) * (size_t)B * N * hw);
    int i = 0;
    for (int q_x_ = -n; q_x_ <= n; ++q_x_)
        for (int q_y_ = -n; q_y_ <= n; ++q_y_) {
            int q_x = q_x_, q_y = q_y_;
            if (!fwd) { q_x = -q_x; q_y = -q_y; }
            int qx0 = q_x, qx1 = w, px0 = 0;
            if (q_x < 0) { qx0 = 0; qx1 = w + q_x; px0 = -q_x; }
            int qy0 = q_y, qy1 = h, py0 = 0;
            if (q_y < 0) { qy0 = 0; qy1 = h + q_y; py0 = -q_y; }
            if (qx1 > qx0 && qy1 > qy0) {
                for (int b = 0; b < B; ++b) {
                    const float *go = grad_out + ((long)b * win * win + i) * hw;
                    for (int k = 0; k < N; ++k) {
                        const float *r = ref + ((long)b * N + k) * hw, *g = frame + ((long)b * N + k) * hw;
                        float *gr = grad_ref + ((long)b * N + k) * hw, *gf = grad_frame + ((long)b * N + k) * hw;
                        for (int y = qy0; y < qy1; ++y)
                            for (int x = qx0; x < qx1; ++x) {
                                const long q = (long)y * w + x, pp = (long)(y - qy0 + py0) * w + (x - qx0 + px0);
                                gr[q] += go[q] * g[pp];
                                gf[pp] += go[q] * r[q];
                            }
                    }
                }
            }
            ++i;
        }
    const float div = (float)N;
    for (long j = 0; j < (long)B * N * hw; ++j) { grad_ref[j] = grad_ref[j] / div; grad_frame[j] = grad_frame[j] / div; }
}

/* ------------------------------------------------------------------------- */
/* createModelMulti(opt) of models/pwc.lua:87-508, every branch the option table
 * selects (frames = 3, pwc_siamese = 1 and pwc_skip >= 1 are fixed: the only values
 * the reference's CostVolMulti call sites / shipped models use).
 * The shipped models are opts.lua:83-98: levels 7, pwc_ws 9, pwc_skip 2, residual 0,
 * occ_input 0, rescale_flow 0, flownet_factor 20, pwc_sum_cvs false, two_frame 0,
 * past_flow = false ("Ours-Hard") / true ("Ours-Soft-*"); createModelMulti(nil)
 * defaults to win 5 / levels 4 (pwc.lua:88).
 *
 * Canonical flat weight order used by this repo (oracle and product agree on
 * it) = graph-construction order of pwc.lua: feature units
 * l = 2..levels {conv1.w, conv1.b, conv2.w, conv2.b}; then for l = levels..l_st:
 * occ decoder, flow decoder, [past-flow decoder], each 6 x {w, b}; every w in
 * Torch layout Co x Ci x 3 x 3.                                               */
typedef struct orc_opts {
    int win;            /* opt.pwc_ws        pwc.lua:88,108  */
    int levels;         /* opt.levels        :88,110         */
    int skip;           /* opt.pwc_skip      :93,114  (l_st = skip + 1, :136) */
    int two_frame;      /* opt.two_frame     :91,116         */
    int sum_cvs;        /* opt.pwc_sum_cvs   :92,119         */
    int residual;       /* opt.residual      :97,113         */
    int occ_input;      /* opt.occ_input     :98,111         */
    int rescale_flow;   /* opt.rescale_flow  :95,118         */
    int past_flow;      /* opt.past_flow     :101,120        */
    float flownet_factor; /* opt.flownet_factor :94,117      */
    int pruned;         /* not a reference option: 1 = compute only what computeFlow reads (est[1], est[3],
                           back2future.lua:77,87); the other output-table entries are left untouched */
    int siamese;        /* opt.pwc_siamese   :99,115  (0: the average-pooled image instead of the learned pyramid, :125-127,182) */
} orc_opts;

static const int FEAT[8] = {0, 3, 16, 32, 64, 96, 128, 192}; /* featMaps, pwc.lua:29,89 */
static const int DEC[7] = {0, 128, 128, 96, 64, 32, 2};       /* decoder(), pwc.lua:76-85 */

ORC_API void orc_default_opts(orc_opts *o, int past_flow)
{
    o->win = 9; o->levels = 7; o->skip = 2; o->two_frame = 0; o->sum_cvs = 0; o->residual = 0;
    o->occ_input = 0; o->rescale_flow = 0; o->past_flow = past_flow ? 1 : 0; o->flownet_factor = 20.f;
    o->pruned = 0; o->siamese = 1;
}

/* featMaps[l] -- pwc.lua:89,120-127: pwc_skip = 0 gives the level-1 unit featMaps[2] maps; pwc_siamese = 0 makes every level the
 * 3-channel image */
static int featc(const orc_opts *o, int l)
{
    if (!o->siamese) return 3;
    if (l == 1 && o->skip == 0) return FEAT[2];
    return FEAT[l];
}

static int opts_ok(const orc_opts *o)
{
    return o->win >= 1 && (o->win & 1) && o->levels >= 2 && o->levels <= 7 && o->skip >= 0 && o->skip + 1 <= o->levels;
}

static long conv_params(int ci, int co) { return (long)co * ci * 9 + co; }
static long decoder_params(int n)
{
    long s = 0;
    int ci = n;
    for (int i = 1; i <= 6; ++i) { s += conv_params(ci, DEC[i]); ci = DEC[i]; }
    return s;
}
/* pwc.lua:254-285: channels of the cost volume the flow / occlusion decoders see */
static int nd_flow(const orc_opts *o) { return (o->two_frame || o->sum_cvs) ? o->win * o->win : 2 * o->win * o->win; }
static int nd_occ(const orc_opts *o) { return o->two_frame ? o->win * o->win : 2 * o->win * o->win; }
/* pwc.lua:288-305 */
static int occ_in_ch(const orc_opts *o, int l)
{
    int n = nd_occ(o) + featc(o, l);
    if (o->two_frame) n += featc(o, l);
    if (l != o->levels) { n += 2; if (o->occ_input) n += 2; }
    return n;
}
/* pwc.lua:325-337 */
static int flow_in_ch(const orc_opts *o, int l) { return l == o->levels ? nd_flow(o) : nd_flow(o) + featc(o, l) + 2; }

/* the feature units in graph-construction order -- pwc.lua:169-183: [level 1 with pwc_skip = 0], levels 2..levels; none without
 * the siamese net */
static long feat_params(const orc_opts *o)
{
    long s = 0;
    if (!o->siamese) return 0;
    for (int l = (o->skip == 0 ? 1 : 2); l <= o->levels; ++l) {
        const int ci = (l == 1) ? 3 : featc(o, l - 1), co = featc(o, l);
        s += conv_params(ci, co) + conv_params(co, co);
    }
    return s;
}

ORC_API long orc_param_count_ex(const orc_opts *o)
{
    if (!opts_ok(o)) return -1;
    long s = 0;
    s += feat_params(o);
    for (int l = o->levels; l >= o->skip + 1; --l) {
        s += decoder_params(occ_in_ch(o, l)) + decoder_params(flow_in_ch(o, l));
        if (o->past_flow) s += decoder_params(flow_in_ch(o, l));
    }
    return s;
}

ORC_API long orc_param_count(int past_flow)
{
    orc_opts o;
    orc_default_opts(&o, past_flow);
    return orc_param_count_ex(&o);
}

static float *falloc(long n) { return (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1)); }

/* convUnit -- pwc.lua:58-65 */
static const float *conv_unit(const float *x, int B, int ci, int co, int H, int W, int stride,
                              const float *p, float *tmp, float *y)
{
    const float *w1 = p, *b1 = w1 + (long)co * ci * 9, *w2 = b1 + co, *b2 = w2 + (long)co * co * 9;
    orc_conv3x3(x, B, ci, H, W, w1, b1, co, stride, 1, tmp);
    orc_conv3x3(tmp, B, co, H / stride, W / stride, w2, b2, co, 1, 1, y);
    return b2 + co;
}

/* decoder(n) -- pwc.lua:76-85: six 3x3 convs, LeakyReLU(0.2) after the first five.
 * y == NULL: skip the computation (pruned mode), only advance the parameter pointer. */
static const float *run_decoder(const float *x, int B, int n, int h, int w, const float *p, float *y)
{
    if (!y) return p + decoder_params(n);
    const long hw = (long)h * w;
    float *a = falloc((long)B * 128 * hw), *b = falloc((long)B * 128 * hw);
    const float *in = x;
    int ci = n;
    for (int i = 1; i <= 6; ++i) {
        const float *wt = p, *bs = p + (long)DEC[i] * ci * 9;
        float *o = (i == 6) ? y : ((i & 1) ? a : b);
        orc_conv3x3(in, B, ci, h, w, wt, bs, DEC[i], 1, i < 6, o);
        p = bs + DEC[i];
        in = o;
        ci = DEC[i];
    }
    free(a); free(b);
    return p;
}

/* nn.JoinTable(2): channel concat in table order (pwc.lua:267,308,334,337) */
static void join_channels(float *dst, int B, long hw, int nsrc, const float *const *src, const int *ch)
{
    int ct = 0;
    for (int s = 0; s < nsrc; ++s) ct += ch[s];
    for (int b = 0; b < B; ++b) {
        int off = 0;
        for (int s = 0; s < nsrc; ++s) {
            memcpy(dst + ((long)b * ct + off) * hw, src[s] + (long)b * ch[s] * hw, sizeof(float) * (size_t)ch[s] * hw);
            off += ch[s];
        }
    }
}

/* Output table of model:forward, pwc.lua:459-489 with skip > 0: per level l = l_st..levels
 * skip_ufs[l], [skip_ubfs[l]], skip_occs[l], iws[1][l], iws[3][l], all at (H, W) >> (l - l_st).  */
ORC_API int orc_pwc_output_shapes_ex(int H, int W, const orc_opts *o, int *ch, int *oh, int *ow)
{
    int no = 0;
    const int l_st = o->skip + 1;
    for (int l = l_st; l <= o->levels; ++l) {
        const int per = o->past_flow ? 5 : 4;
        for (int j = 0; j < per; ++j) {
            const int is_img = (j >= per - 2);
            ch[no] = is_img ? 3 : 2; oh[no] = H >> (l - l_st); ow[no] = W >> (l - l_st); ++no;
        }
    }
    return no;
}

/* model:forward(imgs).  x: B x 9 x H x W (normalized), H, W multiples of 2^(levels-1).
 * outs: the output table in order (see orc_pwc_output_shapes_ex); caller allocates.  In
 * pruned mode only skip_ufs[l_st], skip_occs[l_st] and -- without past_flow -- iws[1][l_st]
 * (computeFlow's est[1] / est[3], back2future.lua:77,87) are written.
 * Returns the number of output tensors, -1 for unsupported options.                    */
ORC_API int orc_pwc_forward_ex(const float *x, int B, int H, int W, const float *params,
                               const orc_opts *o, float **outs)
{
    if (!opts_ok(o) || (H % (1 << (o->levels - 1))) || (W % (1 << (o->levels - 1)))) return -1;
    const int frames = 3, ref = 2; /* :130-133 */
    const int LEVELS = o->levels, L_ST = o->skip + 1 /* :136 */, WIN = o->win, nd = WIN * WIN;
    const int f_i = o->two_frame ? ref : 1, l_i = o->two_frame ? ref + 1 : frames; /* :160-165 */
    const int pruned = o->pruned;
    float *Is[4] = {0}, *ds[4][8] = {{0}}, *cs[4][8] = {{0}}, *ws[4][8] = {{0}};
    float *fs[9] = {0}, *bfs[9] = {0}, *ufs[9] = {0}, *ubfs[9] = {0}, *uoccs[9] = {0};
    float *skip_ufs[9] = {0}, *skip_ubfs[9] = {0}, *occs[9] = {0}, *skip_occs[9] = {0};
    float *iws[4][9] = {{0}};
    int hh[9], wl[9];
    for (int l = 1; l <= LEVELS; ++l) { hh[l] = H >> (l - 1); wl[l] = W >> (l - 1); }
    const long HW = (long)H * W;

    /* nn.Narrow(2, a, 3) -- pwc.lua:139-145 */
    for (int f = 1; f <= frames; ++f) {
        Is[f] = falloc((long)B * 3 * HW);
        for (int b = 0; b < B; ++b)
            memcpy(Is[f] + (long)b * 3 * HW, x + ((long)b * 9 + 3 * (f - 1)) * HW, sizeof(float) * 3 * HW);
    }
    /* image pyramid for the warped-image outputs -- pwc.lua:148-158 */
    for (int f = 1; f <= frames; ++f) {
        if (f == ref) continue;
        ds[f][1] = Is[f];
        if (pruned) continue;
        for (int k = 2; k <= LEVELS - L_ST + 1; ++k) {
            ds[f][k] = falloc((long)B * 3 * (H >> (k - 1)) * (W >> (k - 1)));
            orc_avgpool2(ds[f][k - 1], B * 3, H >> (k - 2), W >> (k - 2), ds[f][k]);
        }
    }
    /* siamese feature pyramid, shared weights -- pwc.lua:169-211; frames f_i..l_i only */
    const float *p = params;
    {
        for (int f = f_i; f <= l_i; ++f) {
            cs[f][1] = Is[f];                                   /* :204 (nn.Identity with pwc_siamese = 0, :175) */
            const float *q = p;
            if (o->siamese && o->skip == 0) {                   /* :171-173,202-203: convUnit(3, featMaps[1], 1) */
                float *tmp = falloc((long)B * featc(o, 1) * HW);
                cs[f][1] = falloc((long)B * featc(o, 1) * HW);
                q = conv_unit(Is[f], B, 3, featc(o, 1), H, W, 1, q, tmp, cs[f][1]);
                free(tmp);
            }
            for (int l = 2; l <= LEVELS; ++l) {
                cs[f][l] = falloc((long)B * featc(o, l) * hh[l] * wl[l]);
                if (!o->siamese) {                              /* :182: nn.SpatialAveragePooling(2,2,2,2) */
                    orc_avgpool2(cs[f][l - 1], B * 3, hh[l - 1], wl[l - 1], cs[f][l]);
                    continue;
                }
                float *tmp = falloc((long)B * featc(o, l) * hh[l] * wl[l]);
                q = conv_unit(cs[f][l - 1], B, featc(o, l - 1), featc(o, l), hh[l - 1], wl[l - 1], 2, q, tmp, cs[f][l]);
                free(tmp);
            }
        }
        p += feat_params(o);
    }

    for (int l = LEVELS; l >= L_ST; --l) { /* pwc.lua:237 */
        const int h = hh[l], w = wl[l], C = featc(o, l);
        const long hw = (long)h * w;
        /* :238-244: raw features on the coarsest level, warped ones below */
        const float *in_fut = (l == LEVELS) ? cs[3][l] : ws[3][l];
        const float *in_past = (l == LEVELS) ? cs[1][l] : ws[1][l];
        /* cost volumes -- :246-285 */
        float *cv_fwd = falloc((long)B * nd * hw), *cv_bwd = 0, *cv_join = 0, *cv_sum = 0;
        { const float *fr[2] = {cs[ref][l], in_fut}; orc_costvol(fr, 2, B, C, h, w, WIN, 1, cv_fwd); }
        const float *cvs_flow, *cvs_occ;
        if (!o->two_frame) {
            cv_bwd = falloc((long)B * nd * hw);
            { const float *fr[2] = {cs[ref][l], in_past}; orc_costvol(fr, 2, B, C, h, w, WIN, 0, cv_bwd); }
            cv_join = falloc((long)B * 2 * nd * hw); /* JoinTable :267 / :273 */
            { const float *s[2] = {cv_fwd, cv_bwd}; int c[2] = {nd, nd}; join_channels(cv_join, B, hw, 2, s, c); }
            if (!o->sum_cvs) {
                cvs_flow = cv_join; cvs_occ = cv_join;
            } else { /* CAddTable :272 */
                cv_sum = falloc((long)B * nd * hw);
                for (long i = 0; i < (long)B * nd * hw; ++i) cv_sum[i] = cv_fwd[i] + cv_bwd[i];
                cvs_flow = cv_sum; cvs_occ = cv_join;
            }
        } else { /* :279-284 */
            cvs_flow = cv_fwd; cvs_occ = cv_fwd;
        }

        /* occlusion decoder + SpatialSoftMax -- :288-322 */
        {
            const int n = occ_in_ch(o, l);
            /* pruned: an occlusion map is live only at l_st, or everywhere when it feeds the next level (occ_input) */
            const int live = !pruned || l == L_ST || o->occ_input;
            if (live) {
                float *din = falloc((long)B * n * hw), *dout = falloc((long)B * 2 * hw);
                const float *s[5]; int c[5]; int ns = 0;
                s[ns] = cvs_occ; c[ns++] = nd_occ(o);
                s[ns] = cs[ref][l]; c[ns++] = C;
                if (o->two_frame) { s[ns] = cs[ref + 1][l]; c[ns++] = C; }       /* :292-296 */
                if (l != LEVELS) {
                    s[ns] = ufs[l + 1]; c[ns++] = 2;                              /* :299-301 */
                    if (o->occ_input) { s[ns] = uoccs[l + 1]; c[ns++] = 2; }      /* :302-305 */
                }
                join_channels(din, B, hw, ns, s, c);
                p = run_decoder(din, B, n, h, w, p, dout);
                occs[l] = falloc((long)B * 2 * hw);
                orc_spatial_softmax(dout, B, 2, h, w, occs[l]);
                free(din); free(dout);
                /* uoccs = nearest x2 (:311-313); skip_occs = l_st - 2 more nearest x2 (:316-321) */
                uoccs[l] = falloc((long)B * 2 * hw * 4);
                orc_upsample_nearest2x(occs[l], B * 2, h, w, uoccs[l]);
                float *cur = uoccs[l];
                int ch_ = 2 * h, cw_ = 2 * w;
                for (int i = 2; i <= L_ST - 1; ++i) {
                    float *nx = falloc((long)B * 2 * ch_ * cw_ * 4);
                    orc_upsample_nearest2x(cur, B * 2, ch_, cw_, nx);
                    if (cur != uoccs[l]) free(cur);
                    cur = nx; ch_ *= 2; cw_ *= 2;
                }
                skip_occs[l] = cur;
            } else {
                p = run_decoder(0, B, n, h, w, p, 0);
            }
        }
        /* flow decoders -- :325-352 */
        {
            const int n = flow_in_ch(o, l);
            const int past_live = o->past_flow && !pruned;
            fs[l] = falloc((long)B * 2 * hw);
            if (past_live) bfs[l] = falloc((long)B * 2 * hw);
            if (l == LEVELS) {
                p = run_decoder(cvs_flow, B, n, h, w, p, fs[l]);
                if (o->past_flow) p = run_decoder(cvs_flow, B, n, h, w, p, past_live ? bfs[l] : 0);
            } else {
                float *din = falloc((long)B * n * hw);
                const float *s[3] = {cvs_flow, cs[ref][l], ufs[l + 1]};
                int c[3] = {nd_flow(o), C, 2};
                join_channels(din, B, hw, 3, s, c);
                p = run_decoder(din, B, n, h, w, p, fs[l]);
                if (o->residual) /* CAddTable :342 */
                    for (long i = 0; i < (long)B * 2 * hw; ++i) fs[l][i] = fs[l][i] + ufs[l + 1][i];
                if (o->past_flow) {
                    if (past_live) { s[2] = ubfs[l + 1]; join_channels(din, B, hw, 3, s, c); }
                    p = run_decoder(din, B, n, h, w, p, past_live ? bfs[l] : 0);
                    if (past_live && o->residual) /* :344 */
                        for (long i = 0; i < (long)B * 2 * hw; ++i) bfs[l][i] = bfs[l][i] + ubfs[l + 1][i];
                }
                free(din);
            }
        }
        free(cv_fwd); free(cv_bwd); free(cv_join); free(cv_sum);
        /* upsampling -- :359-390 (skip > 0): ufs = bilinear x2 [* 2 with rescale_flow];
         * skip_ufs = l_st - 2 further bilinear x2 [* 2 each]                              */
        for (int pass = 0; pass < 2; ++pass) {
            float **src = pass ? bfs : fs, **u = pass ? ubfs : ufs, **sk = pass ? skip_ubfs : skip_ufs;
            if (!src[l]) continue;
            if (o->skip == 0 && l == L_ST) continue;            /* :359: "if skip > 0 or l > l_st" */
            u[l] = falloc((long)B * 2 * hw * 4);
            orc_upsample_bilinear2x(src[l], B * 2, h, w, u[l]);
            if (o->rescale_flow) for (long i = 0; i < (long)B * 2 * hw * 4; ++i) u[l][i] = u[l][i] * 2.0f; /* :365-369 */
            float *cur = u[l];
            int ch_ = 2 * h, cw_ = 2 * w;
            for (int i = 2; i <= L_ST - 1; ++i) { /* :377-389 */
                float *nx = falloc((long)B * 2 * ch_ * cw_ * 4);
                orc_upsample_bilinear2x(cur, B * 2, ch_, cw_, nx);
                if (o->rescale_flow) for (long j = 0; j < (long)B * 2 * ch_ * cw_ * 4; ++j) nx[j] = nx[j] * 2.0f;
                if (cur != u[l]) free(cur);
                cur = nx; ch_ *= 2; cw_ *= 2;
            }
            sk[l] = cur;
        }
        /* warps -- :393-446 */
        for (int f = 1; f <= frames; ++f) {
            if (f == ref) continue;
            if (l > L_ST && f >= f_i && f <= l_i) { /* :395-408 */
                const float k = o->rescale_flow ? (float)((double)o->flownet_factor * (f - ref))
                                                : (float)((double)o->flownet_factor * (f - ref) / pow(2, l - 2));
                ws[f][l - 1] = falloc((long)B * featc(o, l - 1) * hw * 4);
                warping_unit(cs[f][l - 1], ufs[l], k, B, featc(o, l - 1), 2 * h, 2 * w, ws[f][l - 1]);
            }
            /* :422-446: image warp with skip_u(b)fs */
            if (pruned && !(l == L_ST && f == 1 && !o->past_flow)) continue;
            const float *tmp = (o->past_flow && f < ref) ? skip_ubfs[l] : skip_ufs[l];
            if (o->skip == 0) tmp = (o->past_flow && f < ref) ? bfs[l] : fs[l];   /* :423-429 */
            const float k2 = o->rescale_flow ? (float)((double)o->flownet_factor * (f - ref))
                                             : (float)((double)o->flownet_factor * (f - ref) / pow(2, l - L_ST));
            const int uh = H >> (l - L_ST), uw = W >> (l - L_ST);
            iws[f][l] = falloc((long)B * 3 * uh * uw);
            warping_unit(ds[f][l - L_ST + 1], tmp, k2, B, 3, uh, uw, iws[f][l]);
        }
    }

    /* output table -- pwc.lua:459-489 */
    int no = 0;
    for (int l = L_ST; l <= LEVELS; ++l) {
        const long px = (long)B * (H >> (l - L_ST)) * (W >> (l - L_ST));
        /* :462-471: with pwc_skip = 0 the level's own maps, not the upsampled ones */
        const float *o_f = o->skip == 0 ? fs[l] : skip_ufs[l], *o_b = o->skip == 0 ? bfs[l] : skip_ubfs[l];
        const float *o_o = o->skip == 0 ? occs[l] : skip_occs[l];
        if (o_f && (!pruned || l == L_ST)) memcpy(outs[no], o_f, sizeof(float) * 2 * px);
        ++no;
        if (o->past_flow) { if (o_b) memcpy(outs[no], o_b, sizeof(float) * 2 * px); ++no; }
        if (o_o && (!pruned || l == L_ST)) memcpy(outs[no], o_o, sizeof(float) * 2 * px);
        ++no;
        if (iws[1][l]) memcpy(outs[no], iws[1][l], sizeof(float) * 3 * px);
        ++no;
        if (iws[3][l]) memcpy(outs[no], iws[3][l], sizeof(float) * 3 * px);
        ++no;
    }

    for (int f = 1; f <= frames; ++f) {
        free(Is[f]);
        for (int k = 2; k < 8; ++k) free(ds[f][k]);
        if (cs[f][1] != Is[f]) free(cs[f][1]);
        for (int l = 1; l <= 7; ++l) { if (l > 1) free(cs[f][l]); free(ws[f][l]); }
        for (int l = 0; l < 9; ++l) free(iws[f][l]);
    }
    for (int l = 0; l < 9; ++l) {
        if (skip_ufs[l] != ufs[l]) free(skip_ufs[l]);
        if (skip_ubfs[l] != ubfs[l]) free(skip_ubfs[l]);
        if (skip_occs[l] != uoccs[l]) free(skip_occs[l]);
        free(fs[l]); free(bfs[l]); free(ufs[l]); free(ubfs[l]); free(occs[l]); free(uoccs[l]);
    }
    return no;
}

/* The shipped graph (opts.lua:83-98): 20 (Hard) / 25 (Soft) output tensors. */
ORC_API int orc_pwc_forward(const float *x, int B, int H, int W, const float *params,
                            int past_flow, float **outs)
{
    orc_opts o;
    orc_default_opts(&o, past_flow);
    return orc_pwc_forward_ex(x, B, H, W, params, &o, outs);
}

ORC_API int orc_pwc_output_shapes(int H, int W, int past_flow, int *ch, int *oh, int *ow)
{
    orc_opts o;
    orc_default_opts(&o, past_flow);
    return orc_pwc_output_shapes_ex(H, W, &o, ch, oh, ow);
}

/* ------------------------------------------------------------------------- */
/* computeFlow(im1, im2, im3) -- back2future.lua:47-95.  im*: 3 x H0 x W0 planar
 * floats in [0,1].  Outputs: flow 2 x H0 x W0 double (raw network flow, NOT
 * multiplied by 20: SURVEY s0.4), fwd_occ / bwd_occ 1 x H0 x W0 bytes.  est[3]
 * is read by fixed position (:87), i.e. the occlusion map only for Soft
 * models; for a Hard model est[3] = warped image 1 and its channels 2 / 1 are
 * thresholded, exactly as the reference would.  occ_prob (optional, may be
 * NULL) receives est[3][0..1] at net resolution before thresholding.         */
ORC_API int orc_compute_flow(const float *im1, const float *im2, const float *im3,
                             int H0, int W0, const float *params, int past_flow,
                             double *flow, uint8_t *fwd_occ, uint8_t *bwd_occ,
                             float *flow_net, float *occ_net)
{
    const long hw0 = (long)H0 * W0;
    float *imgs = falloc(9 * hw0); /* torch.cat :48 */
    memcpy(imgs, im1, sizeof(float) * 3 * hw0);
    memcpy(imgs + 3 * hw0, im2, sizeof(float) * 3 * hw0);
    memcpy(imgs + 6 * hw0, im3, sizeof(float) * 3 * hw0);
    orc_color_normalize(imgs, 9, H0, W0); /* :49 */
    const int fw = orc_fine_size(W0), fh = orc_fine_size(H0); /* :54-67 */
    if (fw <= 0 || fh <= 0) { free(imgs); return -1; }
    const long hw = (long)fh * fw;
    float *net_in = falloc(9 * hw);
    orc_image_scale_bilinear(imgs, 9, H0, W0, net_in, fh, fw); /* :71 */
    free(imgs);

    int ch[32], oh[32], ow[32];
    const int no = orc_pwc_output_shapes(fh, fw, past_flow, ch, oh, ow);
    float *outs[32];
    for (int i = 0; i < no; ++i) outs[i] = falloc((long)ch[i] * oh[i] * ow[i]);
    orc_pwc_forward(net_in, 1, fh, fw, params, past_flow, outs); /* :73-74 */
    free(net_in);

    /* :77-84 */
    double *fd = (double *)malloc(sizeof(double) * 2 * hw);
    for (long i = 0; i < 2 * hw; ++i) fd[i] = (double)outs[0][i];
    if (flow_net) memcpy(flow_net, outs[0], sizeof(float) * 2 * hw);
    const double sc_h = (double)H0 / (double)fh, sc_w = (double)W0 / (double)fw;
    orc_image_scale_simple_f64(fd, 2, fh, fw, flow, H0, W0);
    for (long i = 0; i < hw0; ++i) { flow[hw0 + i] *= sc_h; }
    for (long i = 0; i < hw0; ++i) { flow[i] *= sc_w; }
    free(fd);
    /* :87-91 : est[3], channel 2 -> fwd, channel 1 -> bwd, threshold 0.6666 in double */
    const float *occ = outs[2];
    if (occ_net) memcpy(occ_net, occ, sizeof(float) * 2 * hw);
    uint8_t *m = (uint8_t *)malloc((size_t)hw);
    for (long i = 0; i < hw; ++i) m[i] = ((double)occ[hw + i] >= 0.6666) ? 1 : 0;
    orc_image_scale_simple_u8(m, 1, fh, fw, fwd_occ, H0, W0);
    for (long i = 0; i < hw; ++i) m[i] = ((double)occ[i] >= 0.6666) ? 1 : 0;
    orc_image_scale_simple_u8(m, 1, fh, fw, bwd_occ, H0, W0);
    free(m);
    for (int i = 0; i < no; ++i) free(outs[i]);
    return 0;
}
