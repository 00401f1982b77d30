// Probe (GPU box): cost of a 64 x 16-byte vector load on the CU's texture-address path, by address pattern, with 12 waves of a CU
// issuing back to back (what the cost-volume kernels' tap gather does).  hipcc --offload-arch=gfx950 -O2 tools/ta_probe.hip
//   pattern 0: 1 KB contiguous per instruction, 128-B aligned
//   pattern 1: runs of 24 pixels x 32 B (768 B) at a row pitch of 480 pixels, lane pair = the two halves of a pixel (the halo gather)
//   pattern 2: pattern 1 shifted by one pixel (32 B): runs start in the middle of a 64-B / 128-B block
//   pattern 3: every pixel (lane pair) on its own row: 32-byte pieces, one per 128-B line
//   pattern 4: pattern 1 with the row start jittered by 0..3 pixels per row (a smooth flow field)
// Prints cycles per load instruction and CU (all waves of the CU issuing), for a working set that stays in L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(768) void probe(const float *base, const int *offs, int nload, long long *out, float *sink)
{
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    const int o0 = offs[tid];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < nload; i += 8) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o0, ((i + k) & 63) * 30720 * 4, 0));
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    const long long t1 = clock64();
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (acc.x == 12345.f) sink[tid] = acc.y;
}
int main()
{
    const int W = 480, NT = 768;
    const size_t floats = (size_t)64 * 30720 * 4 + (size_t)40 * W * 8 + 1024;      // 64 "chunks" 30720 floats apart
    float *buf, *sink; int *doffs; long long *dout;
    hipMalloc(&buf, floats * 4); hipMemset(buf, 0, floats * 4);
    hipMalloc(&sink, NT * 4); hipMalloc(&doffs, NT * 4); hipMalloc(&dout, 256 * 8);
    const char *names[5] = {"1 KB contiguous", "768-B halo rows", "halo rows, +32 B", "32-B pieces, one per line", "halo rows, jittered starts"};
    for (int pat = 0; pat < 5; ++pat) {
        std::vector<int> offs(NT);
        for (int t = 0; t < NT; ++t) {
            const int px = t >> 1, half = t & 1;
            int o;
            if (pat == 0) o = t * 16;
            else if (pat == 3) o = (px * 32 + half * 4) * 4 * 4;                       // 128 B per pixel
            else {
                const int hy = px / 24, hx = px % 24;
                int x = hx + (pat == 2 ? 1 : 0) + (pat == 4 ? (hy * 7) % 4 : 0);
                o = ((hy * W + x) * 8 + half * 4) * 4;
            }
            offs[t] = o;
        }
        hipMemcpy(doffs, offs.data(), NT * 4, hipMemcpyHostToDevice);
        const int nload = 512;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(NT), 0, 0, buf, doffs, nload, dout, sink);
        hipDeviceSynchronize();
        long long h[256];
        hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < 256; ++i) s += (double)h[i];
        printf("pattern %d (%s): %.1f cycles per 64-lane load and CU (12 waves issuing; %.1f per wave instruction)\n", pat, names[pat],
               s / 256 / (nload * 12.0), s / 256 / nload);
    }
    return 0;
}
