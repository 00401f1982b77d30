// Microbenchmark: what ONE CU's vector-memory path sustains for stores and loads that stay in L2 (every block rewrites / rereads its own 128 KB
// region), as bytes per shader clock and CU -- the yardstick for the output stage of the persistent conv kernels (csrc/b2f_wino6.hip writes 147 KB
// per work item and CU).    hipcc --offload-arch=gfx950 -O3 tools/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: 16-byte stores, 1: 8-byte stores, 2: 4-byte stores, 3: 16-byte loads; WAVES x 64 threads per block, REGION bytes per block, ITERS sweeps
template <int MODE>
__global__ void k(float *buf, int region_floats, int iters, long long *cycles, float *sink)
{
    float *base = buf + (size_t)blockIdx.x * region_floats;
    const int per = MODE == 1 ? 2 : MODE == 2 ? 1 : 4;                       // floats per lane and access
    const int step = blockDim.x * per;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int o = threadIdx.x * per; o < region_floats; o += step) {
            if (MODE == 0) *reinterpret_cast<f32x4 *>(base + o) = f32x4{(float)it, 1.f, 2.f, (float)o};
            else if (MODE == 1) *reinterpret_cast<f32x2 *>(base + o) = f32x2{(float)it, (float)o};
            else if (MODE == 2) base[o] = (float)it;
            else acc += *reinterpret_cast<const volatile f32x4 *>(base + o);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

int main()
{
    const int nblk = 256, region = 128 * 1024 / 4, iters = 40;
    float *buf, *sink; long long *cyc;
    hipMalloc(&buf, (size_t)nblk * region * 4); hipMalloc(&sink, 64); hipMalloc(&cyc, nblk * sizeof(long long));
    hipMemset(buf, 0, (size_t)nblk * region * 4);
    const char *names[4] = {"16-byte stores", "8-byte stores", "4-byte stores", "16-byte loads"};
    for (int waves : {1, 2, 4, 8, 16}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int blocks : {1, 256}) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                auto launch = [&]() {
                    if (mode == 0) k<0><<<blocks, waves * 64>>>(buf, region, iters, cyc, sink);
                    else if (mode == 1) k<1><<<blocks, waves * 64>>>(buf, region, iters, cyc, sink);
                    else if (mode == 2) k<2><<<blocks, waves * 64>>>(buf, region, iters, cyc, sink);
                    else k<3><<<blocks, waves * 64>>>(buf, region, iters, cyc, sink);
                };
                launch();
                hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long h[256]; hipMemcpy(h, cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
                double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
                // readcyclecounter = s_memtime: shader clock ticks on gfx9
                printf("%-15s %2d waves, %3d block(s): %6.1f bytes per cycle and CU (block mean %9.0f cycles; kernel %.3f ms = %.2f TB/s)\n", names[mode], waves, blocks,
                       (double)region * 4 * iters / mean, mean, ms, (double)blocks * region * 4 * iters / ms * 1e-9);
            }
        }
    }
    return 0;
}
