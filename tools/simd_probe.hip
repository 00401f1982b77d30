// Probe (GPU box): which SIMD the waves of a workgroup land on, for workgroups of W waves with two workgroups resident per CU
// (LDS-limited).  hipcc --offload-arch=gfx950 -O2 tools/simd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void probe(unsigned *out, int spin)
{
    extern __shared__ char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (spin < 0) smem[threadIdx.x] = 1;
}
int main()
{
    unsigned *d;
    hipMalloc(&d, 512 * 16 * 2 * 4);
    for (int W : {5, 6, 8, 12}) {
        const int lds = W == 12 ? 150000 : 70000;
        const int nb = W == 12 ? 256 : 512;
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipMemset(d, 0xff, 512 * 16 * 2 * 4);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(W * 64), lds, 0, d, 2000000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(512 * 16 * 2);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        // per CU (xcc, se, sh, cu): waves per SIMD
        std::map<unsigned, std::vector<int>> cu;
        std::map<unsigned, std::vector<int>> blocks_of;
        for (int b = 0; b < nb; ++b)
            for (int w = 0; w < W; ++w) {
                const unsigned hw = h[(b * 16 + w) * 2], xcc = h[(b * 16 + w) * 2 + 1] & 15;
                const unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                const unsigned key = xcc << 16 | se << 8 | sh << 4 | cuid;
                if (!cu.count(key)) cu[key] = std::vector<int>(4, 0);
                cu[key][simd]++;
                if (w == 0) blocks_of[key].push_back(b);
            }
        std::map<std::vector<int>, int> hist;
        for (auto &kv : cu) { hist[kv.second]++; }
        printf("W = %d waves per workgroup, %d workgroups: %zu CUs used; waves per SIMD (s0 s1 s2 s3) -> number of CUs\n", W, nb, cu.size());
        for (auto &kv : hist) printf("   %d %d %d %d : %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
        printf("   first block's waves -> simd:");
        for (int w = 0; w < W; ++w) printf(" %u", (h[(0 * 16 + w) * 2] >> 4) & 3);
        printf("   block 1:");
        for (int w = 0; w < W; ++w) printf(" %u", (h[(1 * 16 + w) * 2] >> 4) & 3);
        int shown = 0;
        for (auto &kv : blocks_of) if (shown++ < 3) { printf("\n   CU %05x hosts blocks:", kv.first); for (int b : kv.second) { printf(" %d(", b); for (int w = 0; w < W; ++w) printf("%u", (h[(b * 16 + w) * 2] >> 4) & 3); printf(")"); } }
        printf("\n");
    }
    return 0;
}
