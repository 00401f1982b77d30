// Device side of the computeFlow boundary (/root/reference/back2future.lua:48-93): what the reference does on the
// host around model:forward -- ColorNormalize, image.scale(..., W, H) 'bilinear' down to multiples of 64, and
// after the forward pass image.scale(..., 'simple') back to the input size and the 0.6666 thresholds -- runs here
// on the uploaded planes, so the host only moves bytes (and widens the flow to f64, b2f_pipeline.hip).  The arithmetic
// is the CPU routines' (oracle/b2f_oracle.c) operation for operation: every output element is produced by one
// thread with the same sequence of IEEE fp32 operations (the file is built with -ffp-contract=off and correctly
// rounded division), so the results are bit-identical to the CPU ones.
#include "b2f_internal.h"

namespace b2f {

// element `di` of image.scale's separable line resampler [3P torch/image generic/image.c scaleLinear_rowcol]:
// up = align-corners lerp with the last sample copied, down = fractional box average, float accumulators.
// `fetch(i)` returns source sample i of the line.
template <class F>
__device__ __forceinline__ float scale_line_elem(F fetch, long slen, long dlen, long di)
{
    if (dlen > slen) {
        if (di == dlen - 1) return fetch(slen - 1);
        if (slen == 1) return fetch(0);
        const float scale = (float)(slen - 1) / (float)(dlen - 1);
        float f = di * scale;
        const long i0 = (long)f;
        f -= i0;
        return (1 - f) * fetch(i0) + f * fetch(i0 + 1);
    }
    if (dlen == slen) return fetch(di);
    const float scale = (float)slen / (float)dlen;
    // the running (a_i, a_f) of the sequential loop is the previous element's (e_i, e_f) = split(di * scale)
    float a_f = di * scale;
    const long a_i = (long)a_f;
    a_f -= a_i;
    float e_f = (di + 1) * scale;
    const long e_i = (long)e_f;
    e_f -= e_i;
    float acc = (1 - a_f) * fetch(a_i), wsum = 1 - a_f;
    for (long si = a_i + 1; si < e_i; ++si) { acc += fetch(si); wsum += 1; }
    if (e_i < slen) { acc += e_f * fetch(e_i); wsum += e_f; }
    return acc / wsum;
}

// rows: [planes][Hs][Ws] -> [planes][Hs][Wd]; normalize = ColorNormalize on the fly (plane % 3 = colour)
__global__ void scale_rows_kernel(const float *src, int normalize, long planes, int Hs, int Ws, int Wd, float *dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)planes * Hs * Wd) return;
    const long dx = (long)(i % Wd);
    const size_t row = i / Wd;                 // plane * Hs + y
    const int ch = (int)((row / Hs) % 3);
    const float *s = src + row * Ws;
    dst[i] = scale_line_elem([&](long k) { const float v = s[k]; return normalize ? color_normalize(v, ch) : v; }, Ws, Wd, dx);
}

// columns: [planes][Hs][Wd] -> [planes][Hd][Wd]
__global__ void scale_cols_kernel(const float *src, long planes, int Hs, int Hd, int Wd, float *dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)planes * Hd * Wd) return;
    const int x = (int)(i % Wd);
    const size_t t = i / Wd;
    const long dy = (long)(t % Hd);
    const size_t plane = t / Hd;
    const float *s = src + plane * Hs * Wd + x;
    dst[i] = scale_line_elem([&](long k) { return s[k * Wd]; }, Hs, Hd, dy);
}

hipError_t launch_image_scale(const float *src, int normalize, long planes, int Hs, int Ws, float *tmp, float *dst,
                              int Hd, int Wd, hipStream_t s)
{
    const size_t n1 = (size_t)planes * Hs * Wd, n2 = (size_t)planes * Hd * Wd;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, src, normalize, planes, Hs, Ws, Wd, tmp);
    hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, tmp, planes, Hs, Hd, Wd, dst);
    return hipGetLastError();
}

// back2future.lua:77-93: image.scale(..., 'simple') of est[1] and est[3] to H0 x W0 and the thresholds
// fwd_occ = ge(occ_est[2], 0.6666), bwd_occ = ge(occ_est[1], 0.6666).  The flow leaves the device as the network's
// fp32 values (its :double() copy times sc_w / sc_h is formed by the host threads that hand it to the caller:
// half the bytes on the link); flow32 == nullptr when H0 x W0 is the network size (the flow is downloaded as is).
// flow_net [B][2][fh][fw], est3 [B][est3_ch][fh][fw] -> flow32 [B][2][H0][W0] f32, fwd/bwd [B][H0][W0] u8
__global__ void postprocess_kernel(const float *flow_net, const float *est3, int est3_ch, int B, int fh, int fw, int H0,
                                   int W0, float *flow32, unsigned char *fwd_occ, unsigned char *bwd_occ)
{
    const size_t hw0 = (size_t)H0 * W0, hw = (size_t)fh * fw;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * hw0) return;
    const size_t b = t / hw0, d = t - b * hw0;
    const int j = (int)(d / W0), i = (int)(d - (size_t)j * W0);
    // image.scale 'simple' [3P]: src index = (long)(dst * (float)src_len / dst_len), clamped
    const float scx = (float)fw / (float)W0, scy = (float)fh / (float)H0;
    long jj = (long)((float)j * scy);
    if (jj > fh - 1) jj = fh - 1;
    long ii = (long)((float)i * scx);
    if (ii > fw - 1) ii = fw - 1;
    const size_t s = (size_t)jj * fw + ii;
    const float *e3 = est3 + b * est3_ch * hw;
    if (flow32) {
        const float *fn = flow_net + b * 2 * hw;
        flow32[b * 2 * hw0 + d] = fn[s];
        flow32[b * 2 * hw0 + hw0 + d] = fn[hw + s];
    }
    fwd_occ[t] = ((double)e3[hw + s] >= 0.6666) ? 1 : 0;
    bwd_occ[t] = ((double)e3[s] >= 0.6666) ? 1 : 0;
}

hipError_t launch_postprocess(const float *flow_net, const float *est3, int est3_ch, int B, int fh, int fw, int H0, int W0,
                              float *flow32, unsigned char *fwd_occ, unsigned char *bwd_occ, hipStream_t s)
{
    const size_t n = (size_t)B * H0 * W0;
    hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, flow_net, est3, est3_ch, B, fh,
                       fw, H0, W0, flow32, fwd_occ, bwd_occ);
    return hipGetLastError();
}

// 8-bit transport of input planes whose values are all k / 255 (b2f_ctx.h:pack_u8_piece): the same correctly
// rounded division rebuilds the caller's floats bit for bit.  n multiple of 4 not required.
__global__ void unpack_u8_kernel(const unsigned char *in, size_t n, float *out)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n && ((uintptr_t)(in + i) & 3) == 0 && ((uintptr_t)(out + i) & 15) == 0) {
        const uchar4 k = *reinterpret_cast<const uchar4 *>(in + i);
        *reinterpret_cast<float4 *>(out + i) = make_float4(__fdiv_rn((float)k.x, 255.0f), __fdiv_rn((float)k.y, 255.0f),
                                                           __fdiv_rn((float)k.z, 255.0f), __fdiv_rn((float)k.w, 255.0f));
    } else {
        for (size_t q = i; q < n && q < i + 4; ++q) out[q] = __fdiv_rn((float)in[q], 255.0f);
    }
}

hipError_t launch_unpack_u8(const unsigned char *in, size_t n, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(unpack_u8_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, in, n, out);
    return hipGetLastError();
}

}  // namespace b2f
