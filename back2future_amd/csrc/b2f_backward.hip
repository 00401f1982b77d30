// Backward kernels of the two custom modules (training-side, not on the computeFlow path):
//   nn.BilinearSamplerBHWD:updateGradInput   extras/stnbhwd/BilinearSamplerBHWD.cu:161-307 (backwardBilinearSampling<onlyGrid>),
//                                            Lua side BilinearSamplerBHWD.lua:81-107
//   nn.CostVolMulti:updateGradInput          models/CostVolMulti.lua:111-181
// Layouts of the reference modules (BHWD for the sampler, BDHW for the cost volume), host pointers at the C ABI.
#include "b2f_ctx.h"

namespace b2f {
namespace {


// One half-wave (32 lanes) per output pixel, lanes stride the channels by 32 exactly as the reference's 32 x 16 blocks do
// (BilinearSamplerBHWD.cu:231), the four dot products are reduced with a butterfly whose lane-0 association equals
// sumReduceShMem's tree (:27-36) -- so the grid gradient is bit-identical to the oracle's restatement.  Image gradients
// are scattered with hardware fp32 atomics (order undefined, as in the reference).
template <bool ONLY_GRID>
__global__ __launch_bounds__(256) void warp_bhwd_backward_kernel(const float *img, const float *grid, const float *gout, int B, int ih, int iw,
                                                                int C, int gh, int gw, float *gimg, float *ggrid)
{
    const int lane = threadIdx.x & 31;
    const size_t pix = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const size_t npix = (size_t)B * gh * gw;
    if (pix >= npix) return;
    const int xOut = (int)(pix % gw);
    const size_t r = pix / gw;
    const int yOut = (int)(r % gh);
    const int b = (int)(r / gh);
    const float2 g = *reinterpret_cast<const float2 *>(grid + pix * 2);
    int xl, yt;
    float xw, yw;
    bhwd_top_left(g.x + (float)xOut, iw, xl, xw);
    bhwd_top_left(g.y + (float)yOut, ih, yt, yw);
    const bool x1 = xl + 1 <= iw - 1, y1 = yt + 1 <= ih - 1;   // the top-left tap is always inside (coordinates are clamped)
    const size_t tl = (((size_t)b * ih + yt) * iw + xl) * C;
    const size_t tr = tl + C, bl = tl + (size_t)iw * C, br = bl + C;
    const float *go = gout + pix * C;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    for (int t = lane; t < C; t += 32) {
        const float gv = go[t];
        d0 += img[tl + t] * gv;
        if (!ONLY_GRID) unsafeAtomicAdd(gimg + tl + t, xw * yw * gv);
        if (x1) { d1 += img[tr + t] * gv; if (!ONLY_GRID) unsafeAtomicAdd(gimg + tr + t, (1.f - xw) * yw * gv); }
        if (y1) { d2 += img[bl + t] * gv; if (!ONLY_GRID) unsafeAtomicAdd(gimg + bl + t, xw * (1.f - yw) * gv); }
        if (x1 && y1) { d3 += img[br + t] * gv; if (!ONLY_GRID) unsafeAtomicAdd(gimg + br + t, (1.f - xw) * (1.f - yw) * gv); }
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
        d0 = d0 + __shfl_xor(d0, o, 32);
        d1 = d1 + __shfl_xor(d1, o, 32);
        d2 = d2 + __shfl_xor(d2, o, 32);
        d3 = d3 + __shfl_xor(d3, o, 32);
    }
    if (lane == 0) {
        const float yf = -xw * d0 + xw * d2 - (1.f - xw) * d1 + (1.f - xw) * d3;   // BilinearSamplerBHWD.cu:289
        const float xf = -yw * d0 + yw * d1 - (1.f - yw) * d2 + (1.f - yw) * d3;   // :290
        *reinterpret_cast<float2 *>(ggrid + pix * 2) = make_float2(xf, yf);
    }
}

// CostVolMulti:updateGradInput as two gathers (no atomics): every (b, k, y, x) walks the displacements in the order of
// the Lua loops (q_x_ outer, q_y_ inner) and adds go * value with separate multiply and add roundings, i.e. the same
// sequence of fp32 operations per element as the reference's cmul + add -- bit-identical to the oracle.
__global__ void costvol_backward_kernel(const float *ref, const float *frm, const float *gout, int B, int N, int h, int w, int win, int fwd,
                                        float *gref, float *gfrm)
{
    const size_t hw = (size_t)h * w, n = (size_t)B * N * hw;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t pix = i % hw;
    const int bk = (int)(i / hw);
    const int b = bk / N;
    const int y = (int)(pix / w), x = (int)(pix % w);
    const int nn = (win - 1) / 2;
    const float *go = gout + (size_t)b * win * win * hw;
    const float *rp = ref + (size_t)bk * hw, *fp = frm + (size_t)bk * hw;
    float ar = 0.f, af = 0.f;
    int d = 0;
    for (int qx_ = -nn; qx_ <= nn; ++qx_)
        for (int qy_ = -nn; qy_ <= nn; ++qy_, ++d) {
            const int qx = fwd ? qx_ : -qx_, qy = fwd ? qy_ : -qy_;
            // gradInputRef[y, x] += go[d, y, x] * frame[y - qy, x - qx]                       CostVolMulti.lua:163
            const int yy = y - qy, xx = x - qx;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) ar = ar + go[(size_t)d * hw + pix] * fp[(size_t)yy * w + xx];
            // gradInputFrame[y, x] += go[d, y + qy, x + qx] * ref[y + qy, x + qx]              :164
            const int y2 = y + qy, x2 = x + qx;
            if (y2 >= 0 && y2 < h && x2 >= 0 && x2 < w) af = af + go[(size_t)d * hw + (size_t)y2 * w + x2] * rp[(size_t)y2 * w + x2];
        }
    const float div = (float)N;   // N * (frames - 1), :175-177
    gref[i] = ar / div;
    gfrm[i] = af / div;
}

struct DevB {
    float *p = nullptr;
    ~DevB() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float))); return 0; }
};

}  // namespace
}  // namespace b2f

using namespace b2f;

extern "C" {

int b2f_op_warp_bhwd_backward(b2f_ctx *c, const float *img, const float *grid, const float *grad_out, int B, int ih, int iw, int C, int gh,
                              int gw, float *grad_img, float *grad_grid) try
{
    if (!c || !img || !grid || !grad_out || !grad_grid) return api_fail("b2f_op_warp_bhwd_backward: null argument");
    if (B <= 0 || ih <= 0 || iw <= 0 || C <= 0 || gh <= 0 || gw <= 0) return api_fail("b2f_op_warp_bhwd_backward: bad shape");
    HIPCHK(hipSetDevice(c->device));
    DevB di, dg, dgo, dgi, dgg;
    const size_t ni = (size_t)B * ih * iw * C, ng = (size_t)B * gh * gw * 2, no = (size_t)B * gh * gw * C;
    CHK(di.alloc(ni)); CHK(dg.alloc(ng)); CHK(dgo.alloc(no)); CHK(dgg.alloc(ng));
    HIPCHK(hipMemcpy(di.p, img, ni * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dg.p, grid, ng * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dgo.p, grad_out, no * sizeof(float), hipMemcpyHostToDevice));
    const unsigned blocks = (unsigned)(((size_t)B * gh * gw + 7) / 8);
    if (grad_img) {
        CHK(dgi.alloc(ni));
        HIPCHK(hipMemsetAsync(dgi.p, 0, ni * sizeof(float), c->stream));   // BilinearSamplerBHWD.lua:98-99
        hipLaunchKernelGGL((warp_bhwd_backward_kernel<false>), dim3(blocks), dim3(256), 0, c->stream, di.p, dg.p, dgo.p, B, ih, iw, C, gh, gw, dgi.p, dgg.p);
    } else {
        hipLaunchKernelGGL((warp_bhwd_backward_kernel<true>), dim3(blocks), dim3(256), 0, c->stream, di.p, dg.p, dgo.p, B, ih, iw, C, gh, gw, (float *)nullptr, dgg.p);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (grad_img) HIPCHK(hipMemcpy(grad_img, dgi.p, ni * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(grad_grid, dgg.p, ng * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
B2F_CATCH("b2f_op_warp_bhwd_backward")

int b2f_op_costvol_backward(b2f_ctx *c, const float *ref, const float *frm, const float *grad_out, int B, int C, int h, int w, int win, int fwd,
                            float *grad_ref, float *grad_frm) try
{
    if (!c || !ref || !frm || !grad_out || !grad_ref || !grad_frm) return api_fail("b2f_op_costvol_backward: null argument");
    if (win < 1 || win % 2 == 0) return api_fail("b2f_op_costvol_backward: win must be odd");
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0) return api_fail("b2f_op_costvol_backward: bad shape");
    HIPCHK(hipSetDevice(c->device));
    const size_t hw = (size_t)h * w, nin = (size_t)B * C * hw, ngo = (size_t)B * win * win * hw;
    DevB dr, df, dgo, dgr, dgf;
    CHK(dr.alloc(nin)); CHK(df.alloc(nin)); CHK(dgo.alloc(ngo)); CHK(dgr.alloc(nin)); CHK(dgf.alloc(nin));
    HIPCHK(hipMemcpy(dr.p, ref, nin * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(df.p, frm, nin * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dgo.p, grad_out, ngo * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(costvol_backward_kernel, dim3((unsigned)((nin + 255) / 256)), dim3(256), 0, c->stream, dr.p, df.p, dgo.p, B, C, h, w, win, fwd, dgr.p, dgf.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(grad_ref, dgr.p, nin * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(grad_frm, dgf.p, nin * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
B2F_CATCH("b2f_op_costvol_backward")

}  // extern "C"
