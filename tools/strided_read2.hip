// Microbenchmark 2: strip loads of [C][B] complex64 under the fused kernel's real
// occupancy (LDS-limited blocks per CU). Diagnostic only.
//   usage: strided_read2 (runs a fixed table of experiments)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args { const float4 *vis; float *out; int C; size_t stride4; int n_strips; int group; int rot; int pf; };

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
    // optional: stream-prefetch (contiguous 1 KiB per wave-instruction) this workgroup's share of the
    // slab that workgroups blockIdx + pf * 512 .. will read: 16 KiB per row x all rows, 65536
    // pieces of 1 KiB over 512 workgroups x (T/64) waves
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slab = blockIdx.x / 512 + a.pf;
    const bool do_pf = a.pf > 0 && (slab + 1) * 512 <= a.n_strips && LPR == 2;
    const int pieces_per_wave = 65536 / (512 * (T / 64));
    int piece = ((blockIdx.x % 512) * (T / 64) + wave) * pieces_per_wave;
    for (int k = 0; k < nb; k++) {
        int b = k + start; if (b >= nb) b -= nb;
        const int rb = b * BATCH + r0;
        float4 raw[DEPTH];
        if (do_pf) {
            for (int u = 0; u < pieces_per_wave / nb; u++, piece++) {
                const int row = piece >> 4, col = piece & 15;
                const float4 *ptr = a.vis + (size_t)row * a.stride4 + (size_t)slab * 1024 + col * 64 + lane;
                // fire and forget into a per-wave 1 KiB dump area of LDS (LDS-DMA: no
                // destination registers that the compiler could reuse while in flight)
                __builtin_amdgcn_global_load_lds(ptr, (__attribute__((address_space(3))) void *)(dyn + wave * 256), 16, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < DEPTH; u++) {
            int row = rb + u * RSTEP;
            raw[u] = row < a.C ? base[(size_t)row * a.stride4] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < DEPTH; u++) acc += raw[u].x + raw[u].y + raw[u].z + raw[u].w;
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (acc == 123.456f) a.out[0] = acc + dyn[threadIdx.x];
}

template <int LPR, int DEPTH, int T>
void run(const char *name, const float4 *vis, float *out, int C, int B, int pad, int group, int rot, int lds, int pf = 0)
{
    CHECK(hipFuncSetAttribute((const void *)strip_read<LPR, DEPTH, T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Args a{vis, out, C, (size_t)(B + pad) / 2, B / (2 * LPR), group, rot, pf};
    dim3 grid(a.n_strips);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((strip_read<LPR, DEPTH, T>), grid, dim3(T), lds + 4096, 0, a);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((strip_read<LPR, DEPTH, T>), grid, dim3(T), lds + 4096, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-44s S=%2d T=%d depth=%2d lds=%3dK pad=%3d group=%2d rot=%d : %.3f ms %.2f TB/s\n", name, 2 * LPR, T, DEPTH, lds / 1024, pad, group, rot, ms, (double)C * B * 8 / ms / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int C = 4096, B = argc > 1 ? atoi(argv[1]) : 32768, PADMAX = 256;
    printf("B = %d baselines: %.0f MiB\n", B, (double)C * B * 8 / 1048576);
    const size_t bytes = (size_t)C * (B + PADMAX) * 8;
    float4 *vis; float *out;
    CHECK(hipMalloc(&vis, bytes)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(vis, 1, bytes));
    const int K76 = 76 * 1024, K136 = 136 * 1024;
    run<2, 8, 256>("actual (strip 4, 2 blk/CU)", vis, out, C, B, 0, 8, 0, K76);
    run<2, 4, 256>("depth 4", vis, out, C, B, 0, 8, 0, K76);
    run<4, 4, 256>("strip 8, 256 threads, 2 blk/CU, depth 4", vis, out, C, B, 0, 8, 0, K76);
    run<4, 8, 256>("strip 8, 256 threads, 2 blk/CU, depth 8", vis, out, C, B, 0, 8, 0, K76);
    run<4, 2, 256>("strip 8, 256 threads, 2 blk/CU, depth 2", vis, out, C, B, 0, 8, 0, K76);
    run<8, 4, 256>("strip 16, 256 threads, 2 blk/CU, depth 4", vis, out, C, B, 0, 8, 0, K76);
    run<8, 8, 256>("strip 16, 256 threads, 2 blk/CU, depth 8", vis, out, C, B, 0, 8, 0, K76);
    run<2, 4, 256>("depth 4 + prefetch 1 round ahead", vis, out, C, B, 0, 8, 0, K76, 1);
    run<2, 4, 256>("depth 4 + prefetch 2 rounds ahead", vis, out, C, B, 0, 8, 0, K76, 2);
    run<2, 4, 256>("depth 4 + prefetch 3 rounds ahead", vis, out, C, B, 0, 8, 0, K76, 3);
    run<2, 8, 256>("depth 8 + prefetch 2 rounds ahead", vis, out, C, B, 0, 8, 0, K76, 2);
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
