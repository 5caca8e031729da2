// Microbenchmark: how fast can workgroups read [C][B] complex64 as strips of S
// baselines x all channels (row segments of 8*S bytes)? Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// each lane reads 16 B (2 baselines); LPR lanes per row segment
template <int LPR, int DEPTH>
__global__ __launch_bounds__(512) void strip_read(const float4 *vis, float *out, int C, int B, int remap)
{
    extern __shared__ float dyn[];
    if (C < 0) dyn[threadIdx.x] = 1.f;
    int id = blockIdx.x;
    const int n = gridDim.x;
    if (remap && n % 64 == 0) { int xcd = id & 7, i = id >> 3; id = ((i >> 3) * 8 + xcd) * 8 + (i & 7); }
    const int q = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
    constexpr int RSTEP = 512 / LPR;
    const size_t stride4 = (size_t)B / 2;  // float4 per row
    const float4 *base = vis + (size_t)id * LPR + q;
    float acc = 0.f;
    for (int rb = r0; rb < C; rb += RSTEP * DEPTH) {
        float4 raw[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) {
            int row = rb + u * RSTEP;
            raw[u] = row < C ? base[(size_t)row * stride4] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < DEPTH; u++) acc += raw[u].x + raw[u].y + raw[u].z + raw[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int LPR, int DEPTH>
float run(const float4 *vis, float *out, int C, int B, int remap, int iters, int lds = 0)
{
    if (lds) CHECK(hipFuncSetAttribute((const void *)strip_read<LPR, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int S = LPR * 2;
    dim3 grid(B / S);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((strip_read<LPR, DEPTH>), grid, dim3(512), lds, 0, vis, out, C, B, remap);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; i++)
        hipLaunchKernelGGL((strip_read<LPR, DEPTH>), grid, dim3(512), lds, 0, vis, out, C, B, remap);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main()
{
    const int C = 4096, B = 32768;
    const size_t bytes = (size_t)C * B * 8;
    float4 *vis; float *out;
    CHECK(hipMalloc(&vis, bytes)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(vis, 1, bytes));
    struct { const char *name; float ms; } res[16]; int n = 0;
#define RUN(L, D, R) res[n].name = "S=" #L "x2 depth=" #D " remap=" #R; res[n++].ms = run<L, D>(vis, out, C, B, R, 10);
    RUN(4, 8, 1) RUN(8, 16, 1)
#define RUNL(L, D, R) res[n].name = "1blk/CU S=" #L "x2 depth=" #D " remap=" #R; res[n++].ms = run<L, D>(vis, out, C, B, R, 10, 100 * 1024);
    RUNL(4, 4, 1) RUNL(4, 8, 1) RUNL(4, 16, 1) RUNL(4, 32, 1) RUNL(4, 8, 0) RUNL(8, 8, 1) RUNL(8, 16, 1)
    for (int i = 0; i < n; i++) printf("%-28s %.3f ms  %.2f TB/s\n", res[i].name, res[i].ms, bytes / res[i].ms / 1e9);
    return 0;
}
