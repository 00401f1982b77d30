// Probe (MI355X): issue rate of v_mfma_f32_32x32x16_bf16 as a function of how the accumulators are chained.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_chain tools/mfma_bf16_chain.hip && ./mfma_bf16_chain
// mode 0: 9 independent accumulators round-robin (distance 9)      mode 1: one accumulator, every MFMA depends on the previous
// mode 2: chains of 3 on one accumulator, then the next accumulator (what b2f_wino4s.hip's step does)
// mode 3: two accumulators alternating (distance 2)                 mode 4: three accumulators round-robin (distance 3)
// mode 5: chains of 3 with one v_mov between the MFMAs              mode 6: chains of 3, 12 independent VALU ops after each chain
// each with 1 and 2 waves per SIMD (256 / 512 threads per block, one block per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, long long *cyc)
{
    f32x16 acc[9];
    for (int i = 0; i < 9; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a = {threadIdx.x * 3u + 0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3c003c00u, 0x3c003c00u, threadIdx.x, 0x3c003c00u};
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x * 0.001f + i;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk[6];
    unsigned u[6];
    for (int i = 0; i < 6; ++i) { pk[i] = f32x2{v[i], v[i + 6]}; u[i] = threadIdx.x + i; }
    const bf16x8 A = __builtin_bit_cast(bf16x8, a), B = __builtin_bit_cast(bf16x8, b);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define M(i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[i], 0, 0, 0)
        if (MODE == 0) { _Pragma("unroll") for (int r = 0; r < 3; ++r) { _Pragma("unroll") for (int i = 0; i < 9; ++i) M(i); } }
        if (MODE == 1) { _Pragma("unroll") for (int i = 0; i < 27; ++i) M(0); }
        if (MODE == 2) { _Pragma("unroll") for (int i = 0; i < 9; ++i) { M(i); M(i); M(i); } }
        if (MODE == 3) { _Pragma("unroll") for (int i = 0; i < 13; ++i) { M(0); M(1); } M(0); }
        if (MODE == 4) { _Pragma("unroll") for (int i = 0; i < 9; ++i) { M(0); M(1); M(2); } }
        if (MODE == 5) { _Pragma("unroll") for (int i = 0; i < 9; ++i) { M(i); asm volatile("v_mov_b32 %0, %0" : "+v"(v[0])); M(i); asm volatile("v_mov_b32 %0, %0" : "+v"(v[1])); M(i); } }
        if (MODE == 6) { _Pragma("unroll") for (int i = 0; i < 9; ++i) { M(i); M(i); M(i); __builtin_amdgcn_sched_barrier(0);
                                                                       _Pragma("unroll") for (int q = 0; q < 12; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[q])); __builtin_amdgcn_sched_barrier(0); } }
#define FILL(MODE_, ASM_) if (MODE == MODE_) { _Pragma("unroll") for (int i = 0; i < 9; ++i) { M(i); M(i); M(i); __builtin_amdgcn_sched_barrier(0); \
            _Pragma("unroll") for (int q = 0; q < 12; q += 2) { ASM_; ASM_; } __builtin_amdgcn_sched_barrier(0); } }
        // 12 fillers of one kind after each chain of 3 (all independent of the MFMAs)
        FILL(7, asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(pk[q / 2])));
        FILL(8, asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[q / 2]) : "v"(v[q]), "v"(v[q + 1])));
        FILL(9, asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[q / 2])));
        FILL(10, asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(pk[q / 2])));
        FILL(11, asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[q])));
        FILL(12, asm volatile("v_mov_b32 %0, %1" : "=v"(u[q / 2]) : "v"(v[q])));
        FILL(13, asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[q / 2])));
        // 24 fillers (the split's mix: 6 cvt_pk, 8 shifts / ands, 4 pk_add, 6 mov)
        if (MODE == 14) { _Pragma("unroll") for (int i = 0; i < 9; ++i) { M(i); M(i); M(i); __builtin_amdgcn_sched_barrier(0);
            _Pragma("unroll") for (int q = 0; q < 6; ++q) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[q]) : "v"(v[q]), "v"(v[q + 1]));
            _Pragma("unroll") for (int q = 0; q < 4; ++q) { asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[q])); asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[q + 1])); }
            _Pragma("unroll") for (int q = 0; q < 4; ++q) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(pk[q]));
            _Pragma("unroll") for (int q = 0; q < 6; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(u[q]) : "v"(v[q]));
            __builtin_amdgcn_sched_barrier(0); } }
        // the same 24 with the 4 pk_add as 8 v_add
        if (MODE == 15) { _Pragma("unroll") for (int i = 0; i < 9; ++i) { M(i); M(i); M(i); __builtin_amdgcn_sched_barrier(0);
            _Pragma("unroll") for (int q = 0; q < 6; ++q) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[q]) : "v"(v[q]), "v"(v[q + 1]));
            _Pragma("unroll") for (int q = 0; q < 4; ++q) { asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[q])); asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[q + 1])); }
            _Pragma("unroll") for (int q = 0; q < 8; ++q) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[q]));
            _Pragma("unroll") for (int q = 0; q < 6; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(u[q]) : "v"(v[q]));
            __builtin_amdgcn_sched_barrier(0); } }
        // fp32 MFMA next to bf16 MFMA: mode 16 = 27 fp32 MFMAs (v_mfma_f32_32x32x2_f32) only; mode 17 = per accumulator 4 fp32 MFMAs,
        // then on the next accumulator 3 bf16 MFMAs (9 x (4 + 3)... counted as 27 "units"); mode 18 = waves 0-3 fp32 only, waves 4-7 bf16 only
        if (MODE == 16) { _Pragma("unroll") for (int i = 0; i < 27; ++i) acc[i % 9] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i % 12], v[(i + 1) % 12], acc[i % 9], 0, 0, 0); }
        if (MODE == 17) { _Pragma("unroll") for (int i = 0; i < 9; ++i) {
            if (i & 1) { M(i); M(i); M(i); }
            else { _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], v[j + 1], acc[i], 0, 0, 0); } } }
        if (MODE == 18) { if (threadIdx.x < 256) { _Pragma("unroll") for (int i = 0; i < 27; ++i) acc[i % 9] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i % 12], v[(i + 1) % 12], acc[i % 9], 0, 0, 0); }
                          else { _Pragma("unroll") for (int i = 0; i < 27; ++i) M(i % 9); } }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 9; ++i) s += acc[i][threadIdx.x & 15];
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int i = 0; i < 6; ++i) s += pk[i][0] + pk[i][1] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
static void run(const char *what)
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 8);
    for (int thr = 256; thr <= 512; thr += 256) {
        const int iters = 2000;
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(thr), 0, 0, out, 10, cyc);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(thr), 0, 0, out, iters, cyc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double flop = 2.0 * 32 * 32 * 16 * 27.0 * iters * (thr / 64) * 256;
        printf("mode %d  %d waves/SIMD  %6.1f shader cycles per MFMA per wave  %7.3f ms  %7.1f TFLOP/s  (clock %.0f MHz)   %s\n", MODE, thr / 256, (double)h / (27.0 * iters), ms,
               flop / (ms * 1e-3) / 1e12, (double)h / (ms * 1e3), what);
    }
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0>("9 independent accumulators");
    run<1>("one accumulator, all dependent");
    run<2>("chains of 3 on one accumulator");
    run<3>("two accumulators alternating");
    run<4>("three accumulators round-robin");
    run<5>("chains of 3, one v_mov between the MFMAs");
    run<6>("chains of 3 back to back, then 12 v_fma");
    run<7>("chains of 3, then 12 v_pk_fma_f32");
    run<8>("chains of 3, then 12 v_cvt_pk_bf16_f32");
    run<9>("chains of 3, then 12 v_and_b32");
    run<10>("chains of 3, then 12 v_pk_add_f32");
    run<11>("chains of 3, then 12 v_add_f32");
    run<12>("chains of 3, then 12 v_mov_b32");
    run<13>("chains of 3, then 12 v_lshlrev_b32");
    run<14>("chains of 3, then the split's 24 (6 cvt_pk, 8 and/shift, 4 pk_add, 6 mov)");
    run<15>("chains of 3, then the split's 28 with v_add instead of pk_add");
    run<16>("27 fp32 MFMAs (32x32x2) per iteration, no bf16 (TFLOP/s column is meaningless: 1/8 of the MACs)");
    run<17>("5 x 4 fp32 MFMAs + 4 x 3 bf16 MFMAs per iteration, alternating accumulators");
    run<18>("512 threads only: waves 0-3 27 fp32 MFMAs, waves 4-7 27 bf16 MFMAs (do the two kinds overlap on a SIMD?)");
    return 0;
}
