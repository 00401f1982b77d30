// Probe (GPU box): issue cost of v_fma_f32 and v_pk_fma_f32 (independent accumulators) with 1, 2 and 3 waves per SIMD.
// hipcc --offload-arch=gfx950 -O2 tools/fma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void probe(long long *out, float *sink, int reps, float x)
{
    f2 a[16];
    for (int i = 0; i < 16; ++i) a[i] = (f2){(float)i, (float)threadIdx.x};
    f2 b = {x, x + 1.f}, c = {x + 2.f, x + 3.f};
    __syncthreads();
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            else {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(b.y), "v"(c.y));
            }
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    if (s == 12345.f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main()
{
    long long *d; float *sink;
    hipMalloc(&d, 256 * 8); hipMalloc(&sink, 1024 * 4);
    for (int waves : {4, 8, 12})
        for (int pk = 0; pk < 2; ++pk) {
            const int reps = 2000;
            for (int k = 0; k < 2; ++k) {
                if (pk) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(64 * waves), 0, 0, d, sink, reps, 1.0f);
                else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(64 * waves), 0, 0, d, sink, reps, 1.0f);
            }
            hipDeviceSynchronize();
            long long h[256];
            hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < 256; ++i) s += (double)h[i];
            const double per32 = s / 256 / reps;       // cycles for 32 FMAs per lane per wave
            printf("%d waves per SIMD, %s: %.1f cycles per 32 FMAs per lane of a wave -> %.2f cycles per lane-FMA-pair per SIMD\n", waves / 4,
                   pk ? "16 v_pk_fma_f32" : "32 v_fma_f32   ", per32, per32 / 16 / (waves / 4));
        }
    return 0;
}
