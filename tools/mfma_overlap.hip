// Microbenchmark (gfx950): does fp32 MFMA work overlap with VALU / LDS / global-load work
//   (a) of ANOTHER wave on the same SIMD,  (b) of the SAME wave, interleaved between MFMAs?
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o gpurun_out/mfma_overlap ; run on the GPU box.
// One block per CU (LDS-limited), 256 threads = 1 wave/SIMD or 512 threads = 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(acc0, acc1, acc2, acc3, a, b)                                                     \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);                          \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);                          \
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);                          \
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);                          \
    }

// role: 0 = 16 MFMAs / iter, 1 = 64 VALU fma / iter, 2 = 16 ds_read_b128 / iter, 3 = 8 global_load_dwordx4 / iter,
//       4 = 64 VALU v_add (non-FMA), 5 = 32 v_pk_mul_f32
template <int ROLE>
__device__ __forceinline__ void work(int iters, float *sink, const f32x4 *g, f32x4 *lds, int tid)
{
    if (ROLE == 0) {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        float a = tid * 1e-9f, b = 1.0f;
        for (int i = 0; i < iters; ++i) { MFMA16(a0, a1, a2, a3, a, b); }
        sink[tid] = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (ROLE == 1 || ROLE == 4) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = tid + k;
        const float m = 1.0000001f, c = 1e-9f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (ROLE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(c));
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
                }
        }
        float s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
        sink[tid] = s;
    } else if (ROLE == 2) {
        // VALU-free: 16 ds_read_b128 into four rotating destinations, at most 8 in flight
        f32x4 t0, t1, t2, t3;
        const int addr = (tid & 255) * 16;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4 offset:8192\n ds_read_b128 %3, %4 offset:12288\n s_waitcnt lgkmcnt(4)"
                             : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(addr) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sink[tid] = t0[0] + t1[1] + t2[0] + t3[0];
    } else if (ROLE == 3) {
        // VALU-free: 8 global_load_dwordx4 (L2-resident 64 KB window), at most 8 in flight
        f32x4 t0, t1, t2, t3;
        const f32x4 *ptr = g + (tid & 63) + (tid >> 6) * 512;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:1024\n global_load_dwordx4 %2, %4, off offset:2048\n global_load_dwordx4 %3, %4, off offset:3072\n s_waitcnt vmcnt(4)"
                             : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(ptr) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sink[tid] = t0[0] + t1[1] + t2[0] + t3[0];
    } else if (ROLE == 6) {
        // VALU-free: 8 ds_write_b128
        const f32x4 t = {1.f, 2.f, 3.f, (float)tid};
        const int addr = (tid & 255) * 16;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                asm volatile("ds_write_b128 %1, %0\n ds_write_b128 %1, %0 offset:4096\n ds_write_b128 %1, %0 offset:8192\n ds_write_b128 %1, %0 offset:12288\n s_waitcnt lgkmcnt(4)"
                             :: "v"(t), "v"(addr) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sink[tid] = 0;
    } else if (ROLE == 7) {
        // 64 SALU instructions
        int x = iters;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("s_add_i32 %0, %0, 1" : "+s"(x));
        }
        sink[tid] = x;
    } else if (ROLE == 5) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = f32x2{(float)tid, (float)k};
        const f32x2 m = {1.0000001f, 0.999999f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(m));
        }
        float s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k][0] + v[k][1];
        sink[tid] = s;
    }
}

// both roles in the SAME wave, phases back to back (SEQ) or interleaved 1 MFMA : 4 VALU (ILV)
template <int ILV>
__device__ __forceinline__ void work_same_wave(int iters, float *sink, int tid)
{
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float a = tid * 1e-9f, b = 1.0f;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = tid + k;
    const float m = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < iters; ++i) {
        if (ILV) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                f32x16 &acc = (q & 3) == 0 ? a0 : (q & 3) == 1 ? a1 : (q & 3) == 2 ? a2 : a3;
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(q * 4 + k) & 15]) : "v"(m), "v"(c));
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                f32x16 &acc = (q & 3) == 0 ? a0 : (q & 3) == 1 ? a1 : (q & 3) == 2 ? a2 : a3;
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int q = 0; q < 64; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 15]) : "v"(m), "v"(c));
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
    sink[tid] = s + a0[0] + a1[1] + a2[2] + a3[3];
}

// same wave: 16 MFMAs with memory instructions in their shadows.  KIND 0: 8 global loads (waited for one
// iteration later), 1: 16 ds_read_b128, 2: 8 ds_write_b128
template <int KIND>
__device__ __forceinline__ void work_same_wave_mem(int iters, float *sink, const f32x4 *g, int tid)
{
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float a = tid * 1e-9f, b = 1.0f;
    f32x4 t0 = {}, t1 = {}, t2 = {}, t3 = {};
    const f32x4 *ptr = g + (tid & 63) + (tid >> 6) * 512;
    const int addr = (tid & 255) * 16;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            f32x16 &acc = (q & 3) == 0 ? a0 : (q & 3) == 1 ? a1 : (q & 3) == 2 ? a2 : a3;
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            f32x4 &t = (q & 3) == 0 ? t0 : (q & 3) == 1 ? t1 : (q & 3) == 2 ? t2 : t3;
            if (KIND == 0 && (q & 1)) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(ptr) : "memory");
            if (KIND == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(addr) : "memory");
            if (KIND == 2 && (q & 1)) asm volatile("ds_write_b128 %1, %0" :: "v"(t0), "v"(addr) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    sink[tid] = t0[0] + t1[1] + t2[0] + t3[0] + a0[0] + a1[1] + a2[2] + a3[3];
}

// mode < 10: every wave plays role `mode`;  mode 10 + r: waves 0-3 MFMA, waves 4-7 role r (needs 512 threads)
// mode 20/21: same-wave sequential / interleaved MFMA + VALU
__global__ __launch_bounds__(512) void k(int mode, int iters, float *sink, const f32x4 *g)
{
    extern __shared__ f32x4 lds[];
    const int tid = threadIdx.x;
    float *sk = sink + (size_t)blockIdx.x * 512;
    if (tid < 256) lds[tid] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    const int wave = tid >> 6;
    switch (mode) {
    case 0: work<0>(iters, sk, g, lds, tid); break;
    case 1: work<1>(iters, sk, g, lds, tid); break;
    case 2: work<2>(iters, sk, g, lds, tid); break;
    case 3: work<3>(iters, sk, g, lds, tid); break;
    case 4: work<4>(iters, sk, g, lds, tid); break;
    case 5: work<5>(iters, sk, g, lds, tid); break;
    case 6: work<6>(iters, sk, g, lds, tid); break;
    case 7: work<7>(iters, sk, g, lds, tid); break;
    case 11: if (wave < 4) work<0>(iters, sk, g, lds, tid); else work<1>(iters, sk, g, lds, tid); break;
    case 12: if (wave < 4) work<0>(iters, sk, g, lds, tid); else work<2>(iters, sk, g, lds, tid); break;
    case 13: if (wave < 4) work<0>(iters, sk, g, lds, tid); else work<3>(iters, sk, g, lds, tid); break;
    case 14: if (wave < 4) work<0>(iters, sk, g, lds, tid); else work<4>(iters, sk, g, lds, tid); break;
    case 15: if (wave < 4) work<0>(iters, sk, g, lds, tid); else work<5>(iters, sk, g, lds, tid); break;
    case 16: if (wave < 4) work<0>(iters, sk, g, lds, tid); else work<6>(iters, sk, g, lds, tid); break;
    case 17: if (wave < 4) work<0>(iters, sk, g, lds, tid); else work<7>(iters, sk, g, lds, tid); break;
    case 20: work_same_wave<0>(iters, sk, tid); break;
    case 21: work_same_wave<1>(iters, sk, tid); break;
    case 22: work_same_wave_mem<0>(iters, sk, g, tid); break;
    case 23: work_same_wave_mem<1>(iters, sk, g, tid); break;
    case 24: work_same_wave_mem<2>(iters, sk, g, tid); break;
    }
}

int main()
{
    const int iters = 20000, blocks = 256;
    float *sink;
    f32x4 *g;
    hipMalloc(&sink, (size_t)blocks * 512 * 4);
    hipMalloc(&g, 1024 * 64 * 16);
    hipMemset(g, 0, 1024 * 64 * 16);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct { int mode, threads; const char *what; } runs[] = {
        {0, 256, "MFMA only, 1 wave/SIMD (16 mfma/iter)"},
        {0, 512, "MFMA only, 2 waves/SIMD"},
        {1, 256, "VALU fma only, 1 wave/SIMD (64 fma/iter)"},
        {1, 512, "VALU fma only, 2 waves/SIMD"},
        {4, 256, "VALU add only, 1 wave/SIMD (64 add/iter)"},
        {5, 256, "VALU pk_mul only, 1 wave/SIMD (32 pk_mul/iter)"},
        {2, 256, "ds_read_b128 only, 1 wave/SIMD (16/iter)"},
        {6, 256, "ds_write_b128 only, 1 wave/SIMD (8/iter)"},
        {7, 256, "SALU only (64/iter)"},
        {3, 256, "global_load_dwordx4 only, 1 wave/SIMD (8/iter)"},
        {11, 512, "wave A MFMA + wave B VALU fma (other wave, same SIMD)"},
        {14, 512, "wave A MFMA + wave B VALU add"},
        {15, 512, "wave A MFMA + wave B VALU pk_mul"},
        {12, 512, "wave A MFMA + wave B ds_read_b128"},
        {13, 512, "wave A MFMA + wave B global loads"},
        {16, 512, "wave A MFMA + wave B ds_write_b128"},
        {17, 512, "wave A MFMA + wave B SALU"},
        {20, 256, "same wave: 16 MFMA then 64 VALU fma"},
        {21, 256, "same wave: (1 MFMA, 4 VALU fma) x 16"},
        {21, 512, "2 waves/SIMD, each (1 MFMA, 4 VALU fma) x 16"},
        {20, 512, "2 waves/SIMD, each 16 MFMA then 64 VALU"},
        {22, 256, "same wave: 16 MFMA + 8 global loads interleaved"},
        {22, 512, "2 waves/SIMD, each 16 MFMA + 8 global loads"},
        {23, 256, "same wave: 16 MFMA + 16 ds_read_b128 interleaved"},
        {24, 256, "same wave: 16 MFMA + 8 ds_write_b128 interleaved"},
    };
    for (auto &r : runs) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(r.threads), 100 * 1024, 0, r.mode, iters, sink, g);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) printf("mode %2d thr %3d  %8.3f ms  %7.1f ns/iter   %s\n", r.mode, r.threads, ms, ms * 1e6 / iters, r.what);
        }
    }
    return 0;
}
