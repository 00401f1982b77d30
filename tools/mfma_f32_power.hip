// Microbenchmark: sustained rate and shader clock of the fp32 multiply paths of gfx950 with the whole chip busy (256 blocks, one or two
// waves per SIMD, random operands): v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32, v_mfma_f32_4x4x1_16B_f32, plain v_fma_f32 and
// v_pk_fma_f32.  The question: is the dominant kernel's 2.1 GHz (against 2.44 GHz of lighter kernels) the price of the fp32 MFMA itself,
// and is any instruction cheaper per MAC?    hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_power.hip -o /tmp/mfp && /tmp/mfp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void k(const float *in, float *out, long long *stamps, int iters, int waves)
{
    const int wave = threadIdx.x >> 6;
    if (wave >= waves) return;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) & 16383]; b[i] = in[(threadIdx.x * 16 + 8 + i) & 16383]; }
    f32x16 c32[4];
    f32x4 c16[8];
    float cf[32];
    f32x2 cp[16];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c32[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) c16[i][r] = 0.f;
    for (int i = 0; i < 32; ++i) cf[i] = 0.f;
    for (int i = 0; i < 16; ++i) cp[i] = f32x2{0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int s = 0; s < 32; ++s) c32[s & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 7], b[(s + 3) & 7], c32[s & 3], 0, 0, 0);
        } else if (KIND == 1) {
#pragma unroll
            for (int s = 0; s < 64; ++s) c16[s & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 7], b[(s + 3) & 7], c16[s & 7], 0, 0, 0);
        } else if (KIND == 2) {
#pragma unroll
            for (int s = 0; s < 64; ++s) c16[s & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[s & 7], b[(s + 3) & 7], c16[s & 7], 0, 0, 0);
        } else if (KIND == 3) {
#pragma unroll
            for (int s = 0; s < 256; ++s) cf[s & 31] = __builtin_fmaf(a[s & 7], b[(s >> 3) & 7], cf[s & 31]);
        } else {
#pragma unroll
            for (int s = 0; s < 128; ++s) cp[s & 15] = __builtin_elementwise_fma(f32x2{a[s & 7], a[(s + 1) & 7]}, f32x2{b[(s >> 3) & 7], b[(s >> 2) & 7]}, cp[s & 15]);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += c32[i][r];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) sum += c16[i][r];
    for (int i = 0; i < 32; ++i) sum += cf[i];
    for (int i = 0; i < 16; ++i) sum += cp[i][0] + cp[i][1];
    if (sum == 123.456f) out[1] = sum;
    if (blockIdx.x == 8 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

int main()
{
    float *in, *out; long long *st;
    hipMalloc(&in, 16384 * 4); hipMalloc(&out, 64); hipMalloc(&st, 16);
    std::vector<float> h(16384);
    unsigned x = 12345;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) % 20001 - 10000) * 1e-4f; }
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int waves = 4; waves <= 8; waves += 4) {
        auto run = [&](auto kern, const char *name, double macs_per_iter, int iters) {
            long long hs[2];
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            kern<<<256, 512>>>(in, out, st, iters / 10, waves);
            hipEventRecord(e0);
            kern<<<256, 512>>>(in, out, st, iters, waves);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(hs, st, 16, hipMemcpyDeviceToHost);
            const double macs = macs_per_iter * iters * 256.0 * waves;
            printf("%-28s %d waves per CU: clock %.0f MHz, wall %.2f ms, %.1f TFLOP/s (%.1f MAC per cycle and CU)\n", name, waves, hs[0] / (hs[1] / 100.0), ms,
                   2.0 * macs / (ms * 1e-3) * 1e-12, macs / 256.0 / (double)hs[0]);
        };
        run(k<0>, "v_mfma_f32_32x32x2_f32", 32.0 * 2048, 20000);
        run(k<1>, "v_mfma_f32_16x16x4_f32", 64.0 * 1024, 20000);
        run(k<2>, "v_mfma_f32_4x4x1_16B_f32", 64.0 * 256, 20000);
        run(k<3>, "v_fma_f32", 256.0 * 64, 20000);
        run(k<4>, "v_pk_fma_f32", 128.0 * 128, 20000);
    }
    return 0;
}
