// Microbenchmark: what the memory system sustains for streaming reads, streaming writes and mixes of them (float4 per lane, whole lines,
// 2 GB working set >> the 256 MB last-level cache), as the yardstick for the write-heavy kernels of the path (cost-volume records: 62 %
// of the algorithmic bytes are stores; conv_first: 57 %).    hipcc --offload-arch=gfx950 -O3 tools/hbm_rw.hip -o /tmp/hbm_rw && /tmp/hbm_rw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// every thread: NR reads and NW writes of 16 bytes per iteration, consecutive lanes consecutive addresses
template <int NR, int NW>
__global__ __launch_bounds__(256) void k(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, size_t n4, float *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (NR > NW ? NR : NW) * stride <= n4; i += (NR > NW ? NR : NW) * stride) {
#pragma unroll
        for (int r = 0; r < NR; ++r) acc += in[i + r * stride];
#pragma unroll
        for (int w = 0; w < NW; ++w) out[i + w * stride] = f32x4{(float)i, acc[0], (float)w, 1.f};
    }
    if (NR && acc[0] == 123.456f) sink[0] = acc[1];
}

int main()
{
    const size_t bytes = (size_t)2 << 30, n4 = bytes / 16;
    f32x4 *a, *b; float *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 64);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    auto run = [&](auto kern, const char *name, double rd, double wr) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        kern<<<256 * 16, 256>>>(a, b, n4, sink);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) kern<<<256 * 16, 256>>>(a, b, n4, sink);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-34s %.3f ms  read %.2f TB/s + write %.2f TB/s = %.2f TB/s\n", name, ms, rd * bytes / ms * 1e-9, wr * bytes / ms * 1e-9, (rd + wr) * bytes / ms * 1e-9);
    };
    run(k<4, 0>, "read only", 1.0, 0.0);
    run(k<0, 4>, "write only", 0.0, 1.0);
    run(k<2, 2>, "copy (1 read : 1 write)", 1.0, 1.0);
    run(k<1, 2>, "1 read : 2 writes (records' mix)", 0.5, 1.0);   // covers half of `in`
    run(k<3, 1>, "3 reads : 1 write", 1.0, 1.0 / 3.0);
    return 0;
}
