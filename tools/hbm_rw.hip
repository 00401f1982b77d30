// Microbenchmark: what the memory system sustains for streaming reads, streaming writes and mixes of them (float4 per lane, whole lines,
// 2 GB working set >> the 256 MB last-level cache), as the yardstick for the write-heavy kernels of the path (cost-volume records: 62 %
// of the algorithmic bytes are stores; conv_first: 57 %).    hipcc --offload-arch=gfx950 -O3 tools/hbm_rw.hip -o /tmp/hbm_rw && /tmp/hbm_rw
// Round 6: the first version (grid-stride, 2 - 4 accesses per iteration, plain stores) copied at 4.8 TB/s where MI355X_MICROARCH.md quotes 6.29
// for a float4 copy -- the sweep below looks for the difference: accesses in flight per lane (2 .. 16), grid size, nontemporal loads / stores,
// a contiguous 2-MB-aligned region per block instead of the grid-stride interleave, and hipMemcpyDtoD as a third opinion.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// every thread: NR reads and NW writes of 16 bytes per iteration, consecutive lanes consecutive addresses.
// NT: 1 nontemporal stores, 2 nontemporal loads, 3 both.  REGION: each block streams its own contiguous region (iterations step by the block size)
template <int NR, int NW, int NT, bool REGION>
__global__ __launch_bounds__(256) void k(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, size_t n4, float *sink)
{
    constexpr int NA = NR > NW ? NR : NW;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t i, end, stride;
    if (REGION) {
        const size_t per = n4 / gridDim.x / (NA * 256) * (NA * 256);      // float4 per block, whole iterations
        i = (size_t)blockIdx.x * per + threadIdx.x;
        end = (size_t)blockIdx.x * per + per;
        stride = 256;
    } else {
        stride = (size_t)gridDim.x * blockDim.x;
        i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        end = n4;
    }
    for (; i + (NA - 1) * stride < end; i += NA * stride) {
        f32x4 v[NR ? NR : 1];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = (NT & 2) ? __builtin_nontemporal_load(in + i + r * stride) : in[i + r * stride];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc += v[r];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const f32x4 o = NR ? v[w % (NR ? NR : 1)] : f32x4{(float)i, acc[0], (float)w, 1.f};
            if (NT & 1) __builtin_nontemporal_store(o, out + i + w * stride);
            else out[i + w * stride] = o;
        }
    }
    if (NR && acc[0] == 123.456f) sink[0] = acc[1];
}

int main()
{
    const size_t bytes = (size_t)2 << 30, n4 = bytes / 16;
    f32x4 *a, *b; float *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 64);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int blocks, const char *name, double rd, double wr) {
        kern<<<blocks, 256>>>(a, b, n4, sink);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) kern<<<blocks, 256>>>(a, b, n4, sink);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-58s grid %5d  %.3f ms  read %.2f + write %.2f = %.2f TB/s\n", name, blocks, ms, rd * bytes / ms * 1e-9, wr * bytes / ms * 1e-9, (rd + wr) * bytes / ms * 1e-9);
    };
    printf("-- the round-5 rows (grid-stride, plain accesses)\n");
    run(k<4, 0, 0, false>, 4096, "read only, 4 in flight", 1.0, 0.0);
    run(k<0, 4, 0, false>, 4096, "write only, 4 per iteration", 0.0, 1.0);
    run(k<2, 2, 0, false>, 4096, "copy, 2 in flight", 1.0, 1.0);
    run(k<1, 2, 0, false>, 4096, "1 read : 2 writes (records' mix)", 0.5, 1.0);   // covers half of `in`
    printf("-- copy: accesses in flight per lane, grid size\n");
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        run(k<4, 4, 0, false>, g, "copy, 4 in flight", 1.0, 1.0);
        run(k<8, 8, 0, false>, g, "copy, 8 in flight", 1.0, 1.0);
        run(k<16, 16, 0, false>, g, "copy, 16 in flight", 1.0, 1.0);
    }
    printf("-- copy: cache policy, region per block\n");
    run(k<8, 8, 1, false>, 4096, "copy, 8 in flight, nontemporal stores", 1.0, 1.0);
    run(k<8, 8, 2, false>, 4096, "copy, 8 in flight, nontemporal loads", 1.0, 1.0);
    run(k<8, 8, 3, false>, 4096, "copy, 8 in flight, nontemporal loads + stores", 1.0, 1.0);
    run(k<8, 8, 0, true>, 1024, "copy, 8 in flight, contiguous region per block", 1.0, 1.0);
    run(k<8, 8, 3, true>, 1024, "copy, 8 in flight, region per block, nontemporal", 1.0, 1.0);
    run(k<8, 8, 3, true>, 2048, "copy, 8 in flight, region per block, nontemporal", 1.0, 1.0);
    run(k<16, 16, 3, true>, 1024, "copy, 16 in flight, region per block, nontemporal", 1.0, 1.0);
    printf("-- write only / read only with the best forms\n");
    run(k<0, 8, 1, false>, 4096, "write only, 8 per iteration, nontemporal", 0.0, 1.0);
    run(k<0, 8, 1, true>, 1024, "write only, region per block, nontemporal", 0.0, 1.0);
    run(k<8, 0, 2, false>, 4096, "read only, 8 in flight, nontemporal", 1.0, 0.0);
    run(k<16, 0, 0, false>, 4096, "read only, 16 in flight", 1.0, 0.0);
    run(k<4, 8, 1, false>, 4096, "1 read : 2 writes, nontemporal stores", 0.5, 1.0);
    {
        hipMemcpy(b, a, bytes, hipMemcpyDeviceToDevice);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-58s             %.3f ms  read %.2f + write %.2f = %.2f TB/s\n", "hipMemcpyDtoD", ms, bytes / ms * 1e-9, bytes / ms * 1e-9, 2.0 * bytes / ms * 1e-9);
    }
    return 0;
}
