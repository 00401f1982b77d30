// Small HBM-bound kernels around the two hot kernels: input packing, the stand-alone
// bilinear sampler, flow upsampling, softmax + nearest upsampling, layout changes.
// All NHWC fp32 on the device; planar (BDHW) only at the API boundary.
#include "b2f_internal.h"

namespace b2f {

// ---- input: torch.cat + ColorNormalize + resize(1,9,H,W) (back2future.lua:48-49,73,
// transforms.lua:33-45) and nn.Narrow(2,a,3) (pwc.lua:139-145): planar B x 9 x H x W ->
// three frame-major NHWC images with 8 channels (RGB + 5 zeros = one conv K-chunk). ----
__global__ void pack_input_kernel(const float *in, int normalize, int B, int H, int W, float *img)
{
    const size_t hw = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * hw) return;
    const size_t b = i / hw, p = i - b * hw;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float x = in[(b * 9 + f * 3 + c) * hw + p];
            if (normalize) x = color_normalize(x, c);
            v[c] = x;
        }
        float4 *o = reinterpret_cast<float4 *>(img + (((size_t)f * B + b) * hw + p) * kImgC);
        o[0] = make_float4(v[0], v[1], v[2], 0.f);
        o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

hipError_t launch_pack_input(const float *in, int normalize, int B, int H, int W, float *img, hipStream_t s)
{
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(pack_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, normalize, B, H, W, img);
    return hipGetLastError();
}


// ---- nn.BilinearSamplerBHWD forward, CUDA semantics (BilinearSamplerBHWD.cu:41-115);
// one thread per (output pixel, channel): consecutive lanes = consecutive channels. ----
__global__ void warp_nhwc_kernel(const float *img, long img_stride, int pix_stride, int C, int ih, int iw,
                                 const float *grid, float k, int B, int gh, int gw, float *out,
                                 int out_pix_stride)
{
    const size_t total = (size_t)B * gh * gw * C;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    size_t pix = i / C;
    const int x = (int)(pix % gw);
    const size_t row = pix / gw;
    const int y = (int)(row % gh);
    const int b = (int)(row / gh);
    const float2 g = *reinterpret_cast<const float2 *>(grid + pix * 2);
    int xl, yt;
    float wx, wy;
    bhwd_top_left(g.x * k + (float)x, iw, xl, wx);
    bhwd_top_left(g.y * k + (float)y, ih, yt, wy);
    const float *src = img + (size_t)b * img_stride + ((size_t)yt * iw + xl) * pix_stride + c;
    const bool x1 = xl + 1 <= iw - 1, y1 = yt + 1 <= ih - 1;
    const float tl = src[0];
    const float tr = x1 ? src[pix_stride] : 0.f;
    const float bl = y1 ? src[(size_t)iw * pix_stride] : 0.f;
    const float br = (x1 && y1) ? src[(size_t)(iw + 1) * pix_stride] : 0.f;
    out[pix * out_pix_stride + c] = wx * wy * tl + (1.f - wx) * wy * tr + wx * (1.f - wy) * bl + (1.f - wx) * (1.f - wy) * br;
}

hipError_t launch_warp_nhwc(const float *img, long img_stride, int pix_stride, int C, int ih, int iw,
                            const float *grid, float k, int B, int gh, int gw, float *out,
                            int out_pix_stride, hipStream_t s)
{
    const size_t n = (size_t)B * gh * gw * C;
    hipLaunchKernelGGL(warp_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img, img_stride,
                       pix_stride, C, ih, iw, grid, k, B, gh, gw, out, out_pix_stride);
    return hipGetLastError();
}

// ---- nn.SpatialUpSamplingBilinear(2.0) on a 2-channel flow field (pwc.lua:360-381):
// align-corners ratios, h1 = (int)(r*h2), lambda = r*h2 - h1, h1p = h1 < h-1. ----
__device__ __forceinline__ float2 bilerp2(const float *in, int ps, int h, int w, float rh, float rw, int y2, int x2)
{
    const float h1r = rh * (float)y2;
    const int h1 = (int)h1r;
    const int h1p = (h1 < h - 1) ? 1 : 0;
    const float h1l = h1r - (float)h1, h0l = 1.f - h1l;
    const float w1r = rw * (float)x2;
    const int w1 = (int)w1r;
    const int w1p = (w1 < w - 1) ? 1 : 0;
    const float w1l = w1r - (float)w1, w0l = 1.f - w1l;
    const float *p = in + ((size_t)h1 * w + w1) * ps;
    const float2 a = *reinterpret_cast<const float2 *>(p), b = *reinterpret_cast<const float2 *>(p + (size_t)w1p * ps);
    const float2 c = *reinterpret_cast<const float2 *>(p + (size_t)h1p * w * ps);
    const float2 d = *reinterpret_cast<const float2 *>(p + ((size_t)h1p * w + w1p) * ps);
    float2 o;
    o.x = h0l * (w0l * a.x + w1l * b.x) + h1l * (w0l * c.x + w1l * d.x);
    o.y = h0l * (w0l * a.y + w1l * b.y) + h1l * (w0l * c.y + w1l * d.y);
    return o;
}

// One thread = four consecutive output pixels of a row (grid: x groups, output row, image -- no integer divisions); 16-byte
// stores where the row length allows them.
__global__ __launch_bounds__(256) void upsample_flow2x_kernel(const float *in, int ps, int B, int h, int w, float *out, int planar)
{
    const int H2 = 2 * h, W2 = 2 * w;
    const int x0 = 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int y2 = (int)blockIdx.y, b = (int)blockIdx.z;
    if (x0 >= W2) return;
    const float rh = (H2 > 1) ? (float)(h - 1) / (float)(H2 - 1) : 0.f;
    const float rw = (W2 > 1) ? (float)(w - 1) / (float)(W2 - 1) : 0.f;
    const float *src = in + (size_t)b * h * w * ps;
    float2 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = bilerp2(src, ps, h, w, rh, rw, y2, min(x0 + k, W2 - 1));
    const size_t hw2 = (size_t)H2 * W2, row = (size_t)y2 * W2 + x0;
    if (planar) {
        float *o0 = out + ((size_t)b * 2) * hw2 + row, *o1 = o0 + hw2;
        if ((W2 & 3) == 0) {
            *reinterpret_cast<float4 *>(o0) = make_float4(o[0].x, o[1].x, o[2].x, o[3].x);
            *reinterpret_cast<float4 *>(o1) = make_float4(o[0].y, o[1].y, o[2].y, o[3].y);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x0 + k < W2) { o0[k] = o[k].x; o1[k] = o[k].y; }
        }
    } else {
        float2 *op = reinterpret_cast<float2 *>(out) + (size_t)b * hw2 + row;
        if ((W2 & 3) == 0) {
            *reinterpret_cast<float4 *>(op) = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
            *reinterpret_cast<float4 *>(op + 2) = make_float4(o[2].x, o[2].y, o[3].x, o[3].y);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x0 + k < W2) op[k] = o[k];
        }
    }
}

static hipError_t launch_upsample(const float *in, int in_pix_stride, int B, int h, int w, float *out, int planar, hipStream_t s)
{
    if (B <= 0 || h <= 0 || w <= 0) return hipSuccess;
    if (2 * h > 65535 || B > 65535) return hipErrorInvalidValue;
    const int groups = (2 * w + 3) / 4;
    hipLaunchKernelGGL(upsample_flow2x_kernel, dim3((unsigned)((groups + 255) / 256), (unsigned)(2 * h), (unsigned)B), dim3(256), 0, s, in,
                       in_pix_stride, B, h, w, out, planar);
    return hipGetLastError();
}
hipError_t launch_upsample_flow2x(const float *in, int in_pix_stride, int B, int h, int w, float *out, hipStream_t s)
{
    return launch_upsample(in, in_pix_stride, B, h, w, out, 0, s);
}
hipError_t launch_upsample_flow2x_planar(const float *in, int in_pix_stride, int B, int h, int w, float *out, hipStream_t s)
{
    return launch_upsample(in, in_pix_stride, B, h, w, out, 1, s);
}

// ---- nn.SpatialSoftMax over the 2 decoder logits (pwc.lua:308) + two
// nn.SpatialUpSamplingNearest(2) (pwc.lua:311-321) -> planar B x 2 x 4h x 4w.  One thread = one source pixel's four output
// columns of one output row (the softmax once, two 16-byte stores). ----
__global__ __launch_bounds__(256) void softmax_nearest4_kernel(const float *logits, int ps, int B, int h, int w, float *out)
{
    const int H4 = 4 * h, W4 = 4 * w;
    const int x = (int)(blockIdx.x * blockDim.x + threadIdx.x);       // source column
    const int Y = (int)blockIdx.y, b = (int)blockIdx.z;
    if (x >= w) return;
    const float2 z = *reinterpret_cast<const float2 *>(logits + (((size_t)b * h + (Y >> 2)) * w + x) * ps);
    const float m = fmaxf(z.x, z.y);
    const float e0 = expf(z.x - m), e1 = expf(z.y - m);
    const float sum = e0 + e1;
    const float p0 = e0 / sum, p1 = e1 / sum;
    const size_t hw4 = (size_t)H4 * W4;
    float *o0 = out + ((size_t)b * 2) * hw4 + (size_t)Y * W4 + 4 * x;
    *reinterpret_cast<float4 *>(o0) = make_float4(p0, p0, p0, p0);
    *reinterpret_cast<float4 *>(o0 + hw4) = make_float4(p1, p1, p1, p1);
}

hipError_t launch_softmax_nearest4_planar(const float *logits, int in_pix_stride, int B, int h, int w, float *out, hipStream_t s)
{
    if (B <= 0 || h <= 0 || w <= 0) return hipSuccess;
    if (4 * h > 65535 || B > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(softmax_nearest4_kernel, dim3((unsigned)((w + 255) / 256), (unsigned)(4 * h), (unsigned)B), dim3(256), 0, s, logits,
                       in_pix_stride, B, h, w, out);
    return hipGetLastError();
}

// ---- iws[1][3] for Hard models (pwc.lua:422-446): warp the full-resolution frame-1 image
// by k * skip_ufs[3]; image is the packed NHWC8 copy, flow and output are planar. ----
__global__ void warp_image_planar_kernel(const float *img8, const float *flow, float k, int B, int H, int W, float *out)
{
    const size_t hw = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * hw) return;
    const size_t b = i / hw, p = i - b * hw;
    const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
    const float u = flow[(b * 2) * hw + p] * k, v = flow[(b * 2 + 1) * hw + p] * k;
    int xl, yt;
    float wx, wy;
    bhwd_top_left(u + (float)x, W, xl, wx);
    bhwd_top_left(v + (float)y, H, yt, wy);
    const float *src = img8 + (b * hw + (size_t)yt * W + xl) * kImgC;
    const bool x1 = xl + 1 <= W - 1, y1 = yt + 1 <= H - 1;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 tl = *reinterpret_cast<const float4 *>(src);
    const float4 tr = x1 ? *reinterpret_cast<const float4 *>(src + kImgC) : z;
    const float4 bl = y1 ? *reinterpret_cast<const float4 *>(src + (size_t)W * kImgC) : z;
    const float4 br = (x1 && y1) ? *reinterpret_cast<const float4 *>(src + (size_t)(W + 1) * kImgC) : z;
    const float w00 = wx * wy, w01 = (1.f - wx) * wy, w10 = wx * (1.f - wy), w11 = (1.f - wx) * (1.f - wy);
    out[(b * 3) * hw + p] = w00 * tl.x + w01 * tr.x + w10 * bl.x + w11 * br.x;
    out[(b * 3 + 1) * hw + p] = w00 * tl.y + w01 * tr.y + w10 * bl.y + w11 * br.y;
    out[(b * 3 + 2) * hw + p] = w00 * tl.z + w01 * tr.z + w10 * bl.z + w11 * br.z;
}

hipError_t launch_warp_image_planar(const float *img8, const float *flow_planar, float k, int B, int H, int W,
                                    float *out, hipStream_t s)
{
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(warp_image_planar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, img8, flow_planar, k, B, H, W, out);
    return hipGetLastError();
}


// ---- first layer of the siamese pyramid straight from the API tensor: torch.cat + ColorNormalize
// (back2future.lua:48-49, transforms.lua:33-45) + nn.Narrow (pwc.lua:139-145) + convUnit's first
// nn.SpatialConvolution(3,16,3,3,2,2,1,1) + LeakyReLU(0.2) (pwc.lua:58-61) in ONE pass over the
// planar B x 9 x H x W input: no packed NHWC copy of the frames is ever written (that copy was
// 96 B/pixel of HBM writes).  HBM-bound: 36 B read + 48 B written per input pixel.  Block = 256
// threads = 8 x 32 output pixels of one (frame, batch) image; the 17 x 65 x 3 input patch is
// normalized once into LDS; weights [tap 27][cout 16] come in as scalar (SGPR) operands.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void conv_first_kernel(const float *in, int normalize, int B, int H, int W,
                                                         const float *wt /*27 x 16*/, const float *bias /*16*/,
                                                         float *out /*[3][B][2][H/2*W/2][8]*/)
{
    constexpr int TH = 8, TW = 32, PH = 2 * TH + 1, PW = 2 * TW + 1;
    __shared__ float patch[3][PH][PW + 1];
    const int Ho = H >> 1, Wo = W >> 1;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;          // f * B + b
    const int f = img / B, b = img - f * B;
    const int ox0 = tx_i * TW, oy0 = ty_i * TH;
    const int ix0 = 2 * ox0 - 1, iy0 = 2 * oy0 - 1;
    const size_t hw = (size_t)H * W;
    const float *src = in + ((size_t)b * 9 + (size_t)f * 3) * hw;
    // staging without integer divisions: wave w takes patch rows (c, py) = w, w + 4, ... (51 rows = 13 per wave, the
    // row bookkeeping is scalar), lane = column 0..63, column 64 by lane 0; all loads of a wave are issued
    // before the first one is used
    {
        const int lane = threadIdx.x & 63;
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        float v[13], v64[13];
        bool ok[13], ok64[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int r = wv + 4 * i;
            const int c = r >= 2 * PH ? 2 : (r >= PH ? 1 : 0), py = r - c * PH;
            const int gy = iy0 + py;
            const bool row_ok = r < 3 * PH && gy >= 0 && gy < H;
            const float *rowp = src + (size_t)(row_ok ? c : 0) * hw + (size_t)(row_ok ? gy : 0) * W;
            const int gx = ix0 + lane, gx64 = ix0 + 64;
            ok[i] = row_ok && gx >= 0 && gx < W;
            ok64[i] = row_ok && lane == 0 && gx64 < W;
#ifdef B2F_CF_ABLATE
            if (B2F_CF_ABLATE & 2) { v[i] = (float)gx; v64[i] = (float)gy; continue; }
#endif
            v[i] = rowp[ok[i] ? gx : 0];
            v64[i] = rowp[ok64[i] ? gx64 : 0];
        }
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            const int r = wv + 4 * i;
            const int c = r >= 2 * PH ? 2 : (r >= PH ? 1 : 0), py = r - c * PH;
            if (r < 3 * PH) {
                float a = v[i], b = v64[i];
                if (normalize) {
                    a = color_normalize(a, c);
                    b = color_normalize(b, c);
                }
                patch[c][py][lane] = ok[i] ? a : 0.f;    // zero padding of the NORMALIZED image
                if (lane == 0) patch[c][py][64] = ok64[i] ? b : 0.f;
            }
        }
    }
    __syncthreads();
    const int ty = threadIdx.x >> 5, tx = threadIdx.x & 31;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = bias[o];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float v = patch[c][2 * ty + ky][2 * tx + kx];
                const float *w = wt + ((c * 3 + ky) * 3 + kx) * 16;
#pragma unroll
                for (int o = 0; o < 16; ++o) acc[o] = fmaf(v, w[o], acc[o]);
            }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= Ho || ox >= Wo) return;
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = acc[o] > 0.f ? acc[o] : 0.2f * acc[o];
#ifdef B2F_CF_ABLATE
    if ((B2F_CF_ABLATE & 1) && acc[0] != 12345.678f) return;
#endif
    const size_t hwo = (size_t)Ho * Wo;
    float *op = out + (size_t)img * hwo * 16 + ((size_t)oy * Wo + ox) * 8;
    *reinterpret_cast<float4 *>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4 *>(op + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    *reinterpret_cast<float4 *>(op + hwo * 8) = make_float4(acc[8], acc[9], acc[10], acc[11]);
    *reinterpret_cast<float4 *>(op + hwo * 8 + 4) = make_float4(acc[12], acc[13], acc[14], acc[15]);
}

hipError_t launch_conv_first(const float *in, int normalize, int B, int H, int W, const float *wt, const float *bias,
                             float *out, hipStream_t s)
{
    const int Ho = H / 2, Wo = W / 2;
    const int tiles = ((Wo + 31) / 32) * ((Ho + 7) / 8);
    hipLaunchKernelGGL(conv_first_kernel, dim3((unsigned)(tiles * 3 * B)), dim3(256), 0, s, in, normalize, B, H, W, wt, bias, out);
    return hipGetLastError();
}

// nn.SpatialConvolution(Ci, 2, 3,3,1,1,1,1) -- the last layer of every decoder (pwc.lua:82, 2 outputs: flow
// (u, v) resp. the two occlusion logits) -- as a plain VALU kernel: with 2 outputs an MFMA tile would be 94 %
// padding.  HBM-bound (Ci * 4 B read per pixel).  Block = 256 threads = 16 x 16 pixels; the 18 x 18 patch of up
// to 32 input channels sits in LDS as [chunk][k4][pixel] float4 (conflict-free b128 reads); weights
// [chunk][tap][8 ci][2 co] are staged in LDS with the patch and read back as broadcasts (rounds 1-4 read them as scalar operands:
// 36 dependent s_load round trips through a cold scalar cache were ~10 of a launch's ~16 us whatever its size); one thread = one
// pixel, 2 x 9 x Ci fmas.
// Chunk-planar in, chunk-planar out (channels 0, 1 of output chunk 0).
// CPB = input chunks staged per pass: 2 -> 21 KB of LDS, seven blocks per CU: launches that fill the chip; 4 -> 42 KB, the 32-channel
// input of the decoders' last layer in ONE pass (one memory round trip instead of two): launches of a few hundred blocks at most,
// where the round trips are all the run time (a single triplet: 6 launches x ~3 us).  Same operations in the same order.
template <int CPB>
__global__ __launch_bounds__(256) void conv_narrow2_kernel(const float *in, long img_stride, long chunk_stride, int pix_stride,
                                                           int nchunks, int H, int W, const float *wt, const float *bias,
                                                           float *out, long out_img_stride, int out_pix_stride, int leaky)
{
    constexpr int T = 16, P = T + 2, NP = P * P;     // 324 patch pixels
    __shared__ float4 patch[CPB][2][NP + 4];
    __shared__ float4 wsh[CPB * 36];                 // [chunk][tap][ci pair][ci0.co0 ci0.co1 ci1.co0 ci1.co1]
    const int tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
    int bid = blockIdx.x;
    const int tx_i = bid % tiles_x;
    bid /= tiles_x;
    const int ty_i = bid % tiles_y;
    const int img = bid / tiles_y;
    const int ox0 = tx_i * T, oy0 = ty_i * T;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const float *src = in + (size_t)img * img_stride;
    float a0 = bias[0], a1 = bias[1];
    for (int c0 = 0; c0 < nchunks; c0 += CPB) {
        const int nc = min(CPB, nchunks - c0);
        if (c0) __syncthreads();
        // all loads (clamped addresses, no branch) in flight before the first LDS write
        constexpr int NIT = (CPB * 2 * NP + 255) / 256;
        float4 v[NIT];
        bool ok[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + k * 256;
            const int c = i / (2 * NP), r = i - c * (2 * NP);
            const int pix = r >> 1, k4 = r & 1;
            const int py = pix / P, px = pix - py * P;
            const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
            ok[k] = i < nc * 2 * NP && gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[k] = *reinterpret_cast<const float4 *>(src + (size_t)(ok[k] ? c0 + c : 0) * chunk_stride +
                                                     (ok[k] ? (size_t)gy * W + gx : 0) * pix_stride + 4 * k4);
        }
        float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int)threadIdx.x < nc * 36) wv = *reinterpret_cast<const float4 *>(wt + (size_t)c0 * (9 * 8 * 2) + 4 * threadIdx.x);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + k * 256;
            const int c = i / (2 * NP), r = i - c * (2 * NP);
            if (i < nc * 2 * NP) patch[c][r & 1][r >> 1] = ok[k] ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if ((int)threadIdx.x < CPB * 36) wsh[threadIdx.x] = wv;
        __syncthreads();
        for (int c = 0; c < nc; ++c) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int pp = (ty + ky) * P + tx + kx;
                    const float4 lo = patch[c][0][pp], hi = patch[c][1][pp];
                    const float4 *wk = wsh + (c * 9 + ky * 3 + kx) * 4;
                    const float4 w0 = wk[0], w1 = wk[1], w2 = wk[2], w3 = wk[3];
                    a0 = fmaf(lo.x, w0.x, a0);  a1 = fmaf(lo.x, w0.y, a1);
                    a0 = fmaf(lo.y, w0.z, a0);  a1 = fmaf(lo.y, w0.w, a1);
                    a0 = fmaf(lo.z, w1.x, a0);  a1 = fmaf(lo.z, w1.y, a1);
                    a0 = fmaf(lo.w, w1.z, a0);  a1 = fmaf(lo.w, w1.w, a1);
                    a0 = fmaf(hi.x, w2.x, a0);  a1 = fmaf(hi.x, w2.y, a1);
                    a0 = fmaf(hi.y, w2.z, a0);  a1 = fmaf(hi.y, w2.w, a1);
                    a0 = fmaf(hi.z, w3.x, a0);  a1 = fmaf(hi.z, w3.y, a1);
                    a0 = fmaf(hi.w, w3.z, a0);  a1 = fmaf(hi.w, w3.w, a1);
                }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= H || ox >= W) return;
    if (leaky) { a0 = a0 > 0.f ? a0 : 0.2f * a0; a1 = a1 > 0.f ? a1 : 0.2f * a1; }
    float *op = out + (size_t)img * out_img_stride + ((size_t)oy * W + ox) * out_pix_stride;
    *reinterpret_cast<float2 *>(op) = make_float2(a0, a1);
}

hipError_t launch_conv_narrow2(const ConvLaunch &p, hipStream_t s)
{
    if (p.stride != 1 || p.cout != 2 || p.nseg != 1 || (p.seg[0].pix_stride & 3) || (p.out_pix_stride & 1)) return hipErrorInvalidValue;
    const int tiles = ((p.W + 15) / 16) * ((p.H + 15) / 16);
    if ((long)tiles * p.nimg <= 3L * device_cu_count() && p.seg[0].nchunks > 2)   // at most three blocks per CU anyway
        hipLaunchKernelGGL(conv_narrow2_kernel<4>, dim3((unsigned)(tiles * p.nimg)), dim3(256), 0, s, p.seg[0].ptr, p.seg[0].img_stride,
                           p.seg[0].chunk_stride, p.seg[0].pix_stride, p.seg[0].nchunks, p.H, p.W, p.wpk, p.bias, p.out,
                           p.out_img_stride, p.out_pix_stride, p.leaky);
    else
        hipLaunchKernelGGL(conv_narrow2_kernel<2>, dim3((unsigned)(tiles * p.nimg)), dim3(256), 0, s, p.seg[0].ptr, p.seg[0].img_stride,
                           p.seg[0].chunk_stride, p.seg[0].pix_stride, p.seg[0].nchunks, p.H, p.W, p.wpk, p.bias, p.out,
                           p.out_img_stride, p.out_pix_stride, p.leaky);
    return hipGetLastError();
}

size_t narrow2_wpk_floats(int cin_chunks) { return (size_t)cin_chunks * 9 * 8 * 2; }

// [chunk][tap][8 ci][2 co]; channels missing from cin_map (padding) get zero weights
void narrow2_pack_weights(const float *w, const float *b, int Ci, const int *cin_map, int cin_chunks, float *wpk, float *bpk)
{
    for (int c = 0; c < cin_chunks; ++c)
        for (int t = 0; t < 9; ++t)
            for (int j = 0; j < 8; ++j)
                for (int o = 0; o < 2; ++o) {
                    const int k = c * kCK + j;
                    const int ci = cin_map ? cin_map[k] : (k < Ci ? k : -1);
                    wpk[((c * 9 + t) * 8 + j) * 2 + o] = ci >= 0 ? w[((size_t)o * Ci + ci) * 9 + t] : 0.f;
                }
    bpk[0] = b[0];
    bpk[1] = b[1];
}

// ---- iws[1][3] for Hard models straight from the planar API tensor: frame `frame` of in
// (B x 9 x H x W), normalized on the fly, warped by k * planar flow -> planar B x 3 x H x W ----
__global__ void warp_input_planar_kernel(const float *in, int normalize, int frame, const float *flow, float k, int B,
                                         int H, int W, float *out)
{
    const size_t hw = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * hw) return;
    const size_t b = i / hw, p = i - b * hw;
    const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
    const float u = flow[(b * 2) * hw + p] * k, v = flow[(b * 2 + 1) * hw + p] * k;
    int xl, yt;
    float wx, wy;
    bhwd_top_left(u + (float)x, W, xl, wx);
    bhwd_top_left(v + (float)y, H, yt, wy);
    const int dx = (xl + 1 <= W - 1) ? 1 : 0, dy = (yt + 1 <= H - 1) ? W : 0;   // weight is 0 when folded
    const float w00 = wx * wy, w01 = (1.f - wx) * wy, w10 = wx * (1.f - wy), w11 = (1.f - wx) * (1.f - wy);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *src = in + (b * 9 + (size_t)frame * 3 + c) * hw + (size_t)yt * W + xl;
        float tl = src[0], tr = src[dx], bl = src[dy], br = src[dy + dx];
        if (normalize) {
            tl = color_normalize(tl, c); tr = color_normalize(tr, c);
            bl = color_normalize(bl, c); br = color_normalize(br, c);
        }
        out[(b * 3 + c) * hw + p] = w00 * tl + w01 * tr + w10 * bl + w11 * br;
    }
}

hipError_t launch_warp_input_planar(const float *in, int normalize, int frame, const float *flow_planar, float k, int B,
                                    int H, int W, float *out, hipStream_t s)
{
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(warp_input_planar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, normalize, frame,
                       flow_planar, k, B, H, W, out);
    return hipGetLastError();
}

// ---- nn.SpatialAveragePooling(2,2,2,2) (pwc.lua:155) on NHWC, float4 over channels ----
__global__ void avgpool2_kernel(const float *in, int nimg, int H, int W, int C, float *out)
{
    const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
    const size_t n = (size_t)nimg * Ho * Wo * C4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c4 = (int)(i % C4);
    size_t pix = i / C4;
    const int X = (int)(pix % Wo);
    const size_t r = pix / Wo;
    const int Y = (int)(r % Ho);
    const size_t b = r / Ho;
    const float4 *s = reinterpret_cast<const float4 *>(in) + ((b * H + 2 * Y) * W + 2 * X) * C4 + c4;
    const float4 a = s[0], bb = s[C4], c = s[(size_t)W * C4], d = s[(size_t)(W + 1) * C4];
    float4 o;
    o.x = (a.x + bb.x + c.x + d.x) / 4.0f;
    o.y = (a.y + bb.y + c.y + d.y) / 4.0f;
    o.z = (a.z + bb.z + c.z + d.z) / 4.0f;
    o.w = (a.w + bb.w + c.w + d.w) / 4.0f;
    reinterpret_cast<float4 *>(out)[i] = o;
}

hipError_t launch_avgpool2_nhwc(const float *in, int nimg, int H, int W, int C, float *out, hipStream_t s)
{
    const size_t n = (size_t)nimg * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(avgpool2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, nimg, H, W, C, out);
    return hipGetLastError();
}

// ---- layout changes at the API boundary (nn.Transpose({2,3},{3,4}) and back, pwc.lua:69-71) ----
__global__ void nhwc_to_planar_kernel(const float *in, int pix_stride, int C, int B, int h, int w, float *out)
{
    const size_t hw = (size_t)h * w;
    const size_t n = (size_t)B * C * hw;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t p = i % hw;
    const size_t bc = i / hw;
    const int c = (int)(bc % C);
    const size_t b = bc / C;
    out[i] = in[(b * hw + p) * pix_stride + c];
}

hipError_t launch_nhwc_to_planar(const float *in, int pix_stride, int C, int B, int h, int w, float *out, hipStream_t s)
{
    const size_t n = (size_t)B * C * h * w;
    hipLaunchKernelGGL(nhwc_to_planar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, pix_stride, C, B, h, w, out);
    return hipGetLastError();
}

__global__ void cp8_to_planar_kernel(const float *in, int C, int B, int h, int w, float *out)
{
    const size_t hw = (size_t)h * w;
    const size_t n = (size_t)B * C * hw;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t p = i % hw;
    const size_t bc = i / hw;
    const int c = (int)(bc % C);
    const size_t b = bc / C;
    const int nch = (C + 7) / 8;
    out[i] = in[((b * nch + (c >> 3)) * hw + p) * 8 + (c & 7)];
}

hipError_t launch_cp8_to_planar(const float *in, int C, int B, int h, int w, float *out, hipStream_t s)
{
    const size_t n = (size_t)B * C * h * w;
    hipLaunchKernelGGL(cp8_to_planar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, C, B, h, w, out);
    return hipGetLastError();
}

__global__ void planar_to_nhwc_kernel(const float *in, int C, int B, int h, int w, float *out, int pix_stride)
{
    const size_t hw = (size_t)h * w;
    const size_t n = (size_t)B * hw * pix_stride;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % pix_stride);
    const size_t bp = i / pix_stride;
    const size_t p = bp % hw, b = bp / hw;
    out[i] = c < C ? in[(b * C + c) * hw + p] : 0.f;
}

hipError_t launch_planar_to_nhwc(const float *in, int C, int B, int h, int w, float *out, int pix_stride, hipStream_t s)
{
    const size_t n = (size_t)B * h * w * pix_stride;
    hipLaunchKernelGGL(planar_to_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, C, B, h, w, out, pix_stride);
    return hipGetLastError();
}

__global__ void fill_kernel(float *p, size_t n, float v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

hipError_t launch_fill(float *p, size_t n, float v, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

}  // namespace b2f
