// Probe (GPU box): does global_load_lds_dwordx4 reach LDS byte addresses beyond 64 KB through M0?  hipcc --offload-arch=gfx950 -O2 tools/ldsdma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void probe(const float4 *src, float4 *out, unsigned base)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    float4 *l = reinterpret_cast<float4 *>(smem);
    for (int i = lane; i < 160 * 1024 / 16; i += 64) l[i] = make_float4(-1.f, -1.f, -1.f, -1.f);
    __syncthreads();
    const float4 *g = src + lane;
    const unsigned lds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>(smem)) + base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)" ::"v"(g), "s"(lds) : "memory", "m0");
    __syncthreads();
    out[lane] = l[base / 16 + lane];
}
int main()
{
    float4 *src, *out;
    hipMalloc(&src, 64 * 16); hipMalloc(&out, 64 * 16);
    float4 h[64];
    for (int i = 0; i < 64; ++i) h[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (unsigned base : {0u, 32768u, 65536u - 1024u, 65536u, 100000u & ~15u, 160u * 1024u - 1024u}) {
        hipMemset(out, 0, 64 * 16);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, src, out, base);
        float4 r[64];
        hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost);
        int ok = 0;
        for (int i = 0; i < 64; ++i) ok += r[i].x == h[i].x && r[i].w == h[i].w;
        printf("LDS base %6u: %d / 64 lanes landed (lane 5 reads %g)\n", base, ok, r[5].x);
    }
    return 0;
}
