// Microbenchmark 2: strip loads of [C][B] complex64 under the fused kernel's real
// occupancy (LDS-limited blocks per CU). Diagnostic only.
//   usage: strided_read2 (runs a fixed table of experiments)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args { const float4 *vis; float *out; int C; size_t stride4; int n_strips; int group; int rot; };

template <int LPR, int DEPTH, int T>
__global__ __launch_bounds__(T) void strip_read(Args a)
{
    extern __shared__ float dyn[];
    if (a.C < 0) dyn[threadIdx.x] = 1.f;
    int id = blockIdx.x;
    const int G = a.group;
    if (G > 0 && a.n_strips % (8 * G) == 0) { int xcd = id & 7, i = id >> 3; id = ((i / G) * 8 + xcd) * G + (i % G); }
    const int q = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
    constexpr int RSTEP = T / LPR;
    constexpr int BATCH = RSTEP * DEPTH;
    const float4 *base = a.vis + (size_t)id * LPR + q;
    float acc = 0.f;
    const int nb = (a.C + BATCH - 1) / BATCH;
    const int start = a.rot ? (int)((unsigned)blockIdx.x * 2654435761u >> 8) % nb : 0;
    for (int k = 0; k < nb; k++) {
        int b = k + start; if (b >= nb) b -= nb;
        const int rb = b * BATCH + r0;
        float4 raw[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; u++) {
            int row = rb + u * RSTEP;
            raw[u] = row < a.C ? base[(size_t)row * a.stride4] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < DEPTH; u++) acc += raw[u].x + raw[u].y + raw[u].z + raw[u].w;
    }
    if (acc == 123.456f) a.out[0] = acc;
}

template <int LPR, int DEPTH, int T>
void run(const char *name, const float4 *vis, float *out, int C, int B, int pad, int group, int rot, int lds)
{
    CHECK(hipFuncSetAttribute((const void *)strip_read<LPR, DEPTH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Args a{vis, out, C, (size_t)(B + pad) / 2, B / (2 * LPR), group, rot};
    dim3 grid(a.n_strips);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((strip_read<LPR, DEPTH, T>), grid, dim3(T), lds, 0, a);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((strip_read<LPR, DEPTH, T>), grid, dim3(T), lds, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-44s S=%2d T=%d depth=%2d lds=%3dK pad=%3d group=%2d rot=%d : %.3f ms %.2f TB/s\n", name, 2 * LPR, T, DEPTH, lds / 1024, pad, group, rot, ms, (double)C * B * 8 / ms / 1e9);
    fflush(stdout);
}

int main()
{
    const int C = 4096, B = 32768, PADMAX = 256;
    const size_t bytes = (size_t)C * (B + PADMAX) * 8;
    float4 *vis; float *out;
    CHECK(hipMalloc(&vis, bytes)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(vis, 1, bytes));
    const int K76 = 76 * 1024, K136 = 136 * 1024;
    run<2, 8, 256>("actual (strip 4, 2 blk/CU)", vis, out, C, B, 0, 8, 0, K76);
    run<2, 8, 256>("actual + padded stride", vis, out, C, B, 32, 8, 0, K76);
    run<2, 8, 256>("actual + padded stride 256", vis, out, C, B, 256, 8, 0, K76);
    run<2, 8, 256>("actual + rotated start", vis, out, C, B, 0, 8, 1, K76);
    run<2, 8, 256>("actual + group 16", vis, out, C, B, 0, 16, 0, K76);
    run<2, 8, 256>("actual + group 32", vis, out, C, B, 0, 32, 0, K76);
    run<2, 8, 256>("actual + group 64", vis, out, C, B, 0, 64, 0, K76);
    run<2, 8, 256>("actual + no remap", vis, out, C, B, 0, 0, 0, K76);
    run<2, 16, 256>("strip 4 depth 16", vis, out, C, B, 0, 8, 0, K76);
    run<2, 4, 256>("strip 4 depth 4", vis, out, C, B, 0, 8, 0, K76);
    run<2, 8, 256>("strip 4, no LDS limit", vis, out, C, B, 0, 8, 0, 0);
    run<4, 8, 512>("strip 8, 1 blk/CU", vis, out, C, B, 0, 8, 0, K136);
    run<4, 8, 512>("strip 8, 1 blk/CU padded", vis, out, C, B, 32, 8, 0, K136);
    run<4, 8, 512>("strip 8, 1 blk/CU rot", vis, out, C, B, 0, 8, 1, K136);
    run<4, 8, 512>("strip 8, no LDS limit", vis, out, C, B, 0, 8, 0, 0);
    run<8, 8, 512>("strip 16, no LDS limit", vis, out, C, B, 0, 8, 0, 0);
    run<32, 8, 512>("strip 64, no LDS limit", vis, out, C, B, 0, 8, 0, 0);
    run<32, 8, 512>("strip 64, 1 blk/CU", vis, out, C, B, 0, 8, 0, K136);
    return 0;
}
