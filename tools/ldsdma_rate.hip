// Probe (GPU box): issue rate of LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction) from 3 waves of a 512-thread block,
// alone and next to 5 waves that read LDS and issue FMAs back to back (the role-specialised cost-volume kernel's situation).
// hipcc --offload-arch=gfx950 -O2 tools/ldsdma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(const float *src, long long *out, float *sink, int mode, int reps, int stride_kb)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<size_t>(smem));
    float4 *L = reinterpret_cast<float4 *>(smem);
    for (int i = tid; i < 4096; i += 512) L[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    __syncthreads();
    float acc = 0.f;
    if (wave >= 5) {
        const int gw = wave - 5;
        const size_t bi = (size_t)src + (size_t)(blockIdx.x & 7) * 65536 * 4;   // 8 regions of 288 KB: L2 hits
        const unsigned long long base = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(bi >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)bi);
        long long t0 = clock64();
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const int pc = gw + 3 * j;
                const int voff = (pc * stride_kb * 256 + ((r & 7) * 36 * stride_kb * 256) + lane * 4) * 4;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(__builtin_amdgcn_readfirstlane((int)(lds0 + 65536u + 1024u * pc))) : "memory", "m0");
            }
            if (mode & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long t1 = clock64();
        if (lane == 0) out[blockIdx.x * 4 + gw] = t1 - t0;
    } else if (mode & 1) {
        // LDS reads + FMAs until the DMA waves are done is not observable cheaply: run a fixed amount that outlasts them
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < reps * 6; ++r) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = L[(lane + 64 * q + r) & 4095];
                a.x = fmaf(v.x, v.y, a.x); a.y = fmaf(v.y, v.z, a.y); a.z = fmaf(v.z, v.w, a.z); a.w = fmaf(v.w, v.x, a.w);
                a.x = fmaf(v.x, v.z, a.x); a.y = fmaf(v.y, v.w, a.y); a.z = fmaf(v.z, v.x, a.z); a.w = fmaf(v.w, v.y, a.w);
                a.x = fmaf(v.x, v.w, a.x); a.y = fmaf(v.y, v.x, a.y); a.z = fmaf(v.z, v.y, a.z); a.w = fmaf(v.w, v.z, a.w);
                a.x = fmaf(v.x, v.x, a.x); a.y = fmaf(v.y, v.y, a.y); a.z = fmaf(v.z, v.z, a.z); a.w = fmaf(v.w, v.w, a.w);
            }
        }
        acc = a.x + a.y + a.z + a.w;
    }
    __syncthreads();
    if (acc == 12345.f) sink[tid] = acc;
}
int main()
{
    float *src, *sink; long long *dout;
    const size_t floats = (size_t)256 * 65536 + (1 << 22);
    hipMalloc(&src, floats * 4); hipMemset(src, 0, floats * 4);
    hipMalloc(&sink, 512 * 4); hipMalloc(&dout, 256 * 4 * 8);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    const char *nm[4] = {"DMA waves alone, free running", "next to 5 LDS + FMA waves, free running", "alone, vmcnt(0) after every 12", "next to 5 LDS + FMA waves, vmcnt(0) after every 12"};
    for (int stride_kb : {1, 4})
        for (int mode = 0; mode < 4; ++mode) {
            const int reps = 200;
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 150000, 0, src, dout, sink, mode, reps, stride_kb);
            hipDeviceSynchronize();
            long long h[1024];
            hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < 256; ++i) s += (double)(h[4 * i] + h[4 * i + 1] + h[4 * i + 2]) / 3.0;
            printf("pieces %d KB apart, %s: %.0f cycles per round of 12 DMAs per wave (%.1f per DMA of the CU's 36)\n", stride_kb, nm[mode], s / 256 / reps, s / 256 / reps / 36.0);
        }
    return 0;
}
