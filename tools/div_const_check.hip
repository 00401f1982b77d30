// Check (GPU box): (x + (-mean)) / std without the division -- a' = a / 16, q = a' y, r = fma(-q, std, a'), q' = fma(r, y, q),
// result 16 q' (a itself where a is inf / NaN) with y = RN(1 / std): Markstein's correction step on a scaled dividend, so that q
// cannot overflow before the quotient does -- against the IEEE division __fdiv_rn, for EVERY float32 bit pattern of x and the
// three channels of ColorNormalize (transforms.lua:33-45; mean / std of back2future.lua:33-36).  Prints the number of differing
// results per channel and the range of |a| where they occur.   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/div_const_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void check(int c, float mean, float stdv, float y, unsigned long long *bad, unsigned *lo, unsigned *hi)
{
    const unsigned long long n = 1ull << 32;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float a = x + (-mean);
        const float ref = __fdiv_rn(a, stdv);
        const float as = a * 0.0625f;
        const float q = as * y;
        const float r = __builtin_fmaf(-q, stdv, as);
        float f = __builtin_fmaf(r, y, q) * 16.f;
        if (!(__builtin_fabsf(a) < __builtin_inff())) f = a;
        if (__float_as_uint(ref) != __float_as_uint(f) && !(ref != ref && f != f)) {
            atomicAdd(bad, 1ull);
            const unsigned m = __float_as_uint(a) & 0x7fffffffu;
            atomicMin(lo, m);
            atomicMax(hi, m);
        }
    }
}
int main()
{
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    unsigned long long *bad; unsigned *lo, *hi;
    hipMalloc(&bad, 8); hipMalloc(&lo, 4); hipMalloc(&hi, 4);
    for (int c = 0; c < 3; ++c) {
        const float y = (float)(1.0 / (double)stdv[c]);
        unsigned long long z = 0; unsigned l = 0xffffffffu, h = 0;
        hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice); hipMemcpy(lo, &l, 4, hipMemcpyHostToDevice); hipMemcpy(hi, &h, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, c, mean[c], stdv[c], y, bad, lo, hi);
        hipDeviceSynchronize();
        hipMemcpy(&z, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&l, lo, 4, hipMemcpyDeviceToHost); hipMemcpy(&h, hi, 4, hipMemcpyDeviceToHost);
        float fl, fh; memcpy(&fl, &l, 4); memcpy(&fh, &h, 4);
        printf("channel %d (mean %.3f std %.3f, y = %.9g): %llu of 2^32 inputs differ", c, mean[c], stdv[c], y, z);
        if (z) printf("; |x - mean| of the differing ones in [%g, %g]", fl, fh);
        printf("\n");
    }
    return 0;
}
