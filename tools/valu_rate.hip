// Microbenchmark: sustained VALU issue rate of the instruction kinds the fused
// flagger leans on, at 1, 2 and 4 wavefronts per SIMD. Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int N = 16;       // independent chains
constexpr int ITER = 2000;  // loop trips; body has N*8 instrs of the kind under test

template <int KIND>
__global__ void k(float *out, float seed, int iters)
{
    float a[N], b[N];
    for (int i = 0; i < N; i++) { a[i] = seed + i + threadIdx.x; b[i] = seed * 0.5f + i; }
    double d[N / 2];
    for (int i = 0; i < N / 2; i++) d[i] = a[i];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 2) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % N]));
                if (KIND == 3) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (KIND == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i / 2]) : "v"(d[(i / 2 + 1) % (N / 2)]));
                if (KIND == 6) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 7) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(b[i]) : "s20", "s21");
                if (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i / 2]) : "v"(d[(i / 2 + 1) % (N / 2)]));
                if (KIND == 11) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[i / 2]) : "v"(a[i]));
                if (KIND == 12) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % N]));
                if (KIND == 13) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 14) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 15) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
                if (KIND == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 17) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
                if (KIND == 18) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 19) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 20) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i / 2]) : "v"(d[(i / 2 + 1) % (N / 2)]));
                if (KIND == 21) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 22) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 23) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 24) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % N]));
                if (KIND == 25) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 26) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 27) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 28) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a[i]));
                if (KIND == 29) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
                if (KIND == 30) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1" : "+v"(a[i]) : "v"(b[i]));
                if (KIND == 31) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[i]) : "v"(d[i / 2]));
                if (KIND == 32) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i / 2]) : "v"(d[(i / 2 + 1) % (N / 2)]));
                if (KIND == 33) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i / 2]), "v"(d[(i / 2 + 1) % (N / 2)]) : "vcc");
                if (KIND == 34) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i / 2]) : "v"(d[(i / 2 + 1) % (N / 2)]));
                if (KIND == 35) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
                if (KIND == 36) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b[i]) : "vcc");
                if (KIND == 37) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) % N]) : "vcc");
                if (KIND == 38) asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            }
        }
    }
    float acc = 0;
    for (int i = 0; i < N; i++) acc += a[i];
    for (int i = 0; i < N / 2; i++) acc += (float)d[i];
    if (acc == 1.2345f) out[0] = acc;
}

template <int KIND>
void run(const char *name, float *out, int per_wave_instrs_per_body)
{
    for (int threads : {256, 512}) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, 1.f, 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, 1.f, ITER);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double instrs_per_simd = (double)ITER * 8 * N * per_wave_instrs_per_body * (threads / 256);
        printf("%-22s waves/SIMD=%d : %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", name, threads / 256, ms,
               ms * 1e6 / instrs_per_simd, ms * 1e6 / instrs_per_simd * 2.4);
    }
}

int main()
{
    float *out; CHECK(hipMalloc(&out, 4));
    run<0>("v_fma_f32", out, 1);
    run<1>("v_max_f32", out, 1);
    run<2>("v_med3_f32", out, 1);
    run<9>("v_add_u32", out, 1);
    run<12>("v_med3_u32", out, 1);
    run<13>("v_max_u32", out, 1);
    run<14>("v_min_i32", out, 1);
    run<15>("v_cmp_u32+cndmask", out, 2);
    run<16>("v_and_b32", out, 1);
    run<17>("v_lshrrev_b32", out, 1);
    run<18>("v_mov_b32", out, 1);
    run<19>("v_add3_u32", out, 1);
    run<20>("v_fma_f64", out, 1);
    run<21>("v_add_f32", out, 1);
    run<22>("v_mul_f32", out, 1);
    run<23>("v_sqrt_f32", out, 1);
    run<24>("v_min3_u32", out, 1);
    run<25>("v_max_i16", out, 1);
    run<26>("v_pk_max_u16", out, 1);
    run<27>("v_sub_u32", out, 1);
    run<28>("v_bfe_u32", out, 1);
    run<29>("v_readlane", out, 1);
    run<30>("v_mov_dpp", out, 1);
    run<31>("v_cvt_f32_f64", out, 1);
    run<32>("v_max_f64", out, 1);
    run<33>("v_cmp_lt_f64", out, 1);
    run<34>("v_pk_mul_f32", out, 1);
    run<35>("v_cmp_lt_u32", out, 1);
    run<36>("v_cmp_lt_f32", out, 1);
    run<37>("v_cndmask(indep)", out, 1);
    run<38>("v_subb_co_u32", out, 1);
    return 0;
}
