// Probe for v_mfma_f32_4x4x1_16b_f32 on gfx950: operand / result lane layout and issue rate (alone, and with one
// ds_read_b128 per 4 MFMAs as the banded-correlation kernel of b2f_corr.hip issues them).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o tools/bin/mfma4x4_probe && tools/bin/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(float *out)
{
    const int l = threadIdx.x;
    f4 z = {0.f, 0.f, 0.f, 0.f};
    // D = A x B with B = 1: every result is the A value that fed it
    f4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1.0f, z, 0, 0, 0);
    f4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), z, 0, 0, 0);
    for (int v = 0; v < 4; ++v) {
        out[(0 * 4 + v) * 64 + l] = da[v];
        out[(1 * 4 + v) * 64 + l] = db[v];
    }
}

template <int LDS_READS>
__global__ __launch_bounds__(256) void rate_kernel(float *out, long long *cyc, int iters)
{
    __shared__ f4 buf[2048];
    const int l = threadIdx.x;
    for (int i = l; i < 2048; i += 256) buf[i] = f4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    f4 acc[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    f4 b = {1.f + l, 2.f, 3.f, 4.f};
    f4 a[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] = f4{(float)i, (float)l, 1.f, 2.f};
    const f4 *base = buf + (l & 63);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (LDS_READS) {
#pragma unroll
            for (int i = 0; i < 25; ++i) a[i] = base[i * 64 + (it & 7)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < 25; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[i][k], b[k], acc[i], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 25; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + l] = s;
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *d;
    hipMalloc(&d, 1 << 20);
    long long *c;
    hipMalloc(&c, 8192);
    layout_kernel<<<1, 64>>>(d);
    std::vector<float> h(512);
    hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
        printf("%s-source lane (+1) of D[vgpr v][lane]:\n", w ? "B" : "A");
        for (int v = 0; v < 4; ++v) {
            printf(" v%d:", v);
            for (int l = 0; l < 16; ++l) printf(" %2.0f", h[(w * 4 + v) * 64 + l]);
            printf(" ... lane 60..63:");
            for (int l = 60; l < 64; ++l) printf(" %2.0f", h[(w * 4 + v) * 64 + l]);
            printf("\n");
        }
    }
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks : {256, 512, 768}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode) rate_kernel<1><<<blocks, 256>>>(d, c, iters);
                else rate_kernel<0><<<blocks, 256>>>(d, c, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long cy;
            hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            const double mf = (double)blocks * 4 * iters * 100;   // wave-level MFMAs
            printf("%s blocks %4d (x4 waves): %.3f ms, %.2f cycles per MFMA per wave (block 0), %.1f TFLOP/s\n",
                   mode ? "mfma + 1 ds_read_b128 per 4" : "mfma only", blocks, ms, (double)cy / (iters * 100.0), mf * 512 / ms / 1e9);
        }
    return 0;
}
