// Microbenchmark 3: the load side of a PERSISTENT fused flagger -- one 512-thread workgroup
// per CU walks over strips of 8 baselines (64-byte row segments); the visibilities are
// brought in by LDS-DMA (global_load_lds_dwordx4, per-lane source addresses) into a ring of
// NSLOT step slots (a step = the 64 rows {64 l + j}, l = 0..63, that the 64 lanes of a
// wavefront need at position j of their runs), each wavefront picks its baseline's sample out
// of the slot (ds_read_b64) and dummy arithmetic stands in for the amplitude (per step) and
// for the median / MAD / threshold phases (per strip). Diagnostic only.
//   usage: ring_probe [B]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args {
    const float2 *vis;
    double *sum;
    int C;
    size_t stride;  // float2 elements per row
    int n_strips;
    int group;      // XCD-aware strip order: runs of `group` strips per XCD
    int amp_n;      // dummy VALU wave-instructions per step (amplitude stand-in)
    int tail_n;     // dummy VALU wave-instructions per strip after the steps (median, MAD, threshold)
    int linear;     // 1: a step's rows are 64 CONSECUTIVE rows (order test), 0: rows 64 apart
    int stagger;    // odd CUs sleep this many x 8128 cycles once
};

__device__ __forceinline__ int strip_of(int id, int n_strips, int G)
{
    if (G <= 0) return id;
    const int full = (n_strips / (8 * G)) * (8 * G);
    if (id >= full) return id;
    const int xcd = id & 7, i = id >> 3;
    return ((i / G) * 8 + xcd) * G + (i % G);
}

template <int NSLOT, int G>
__global__ __launch_bounds__(512, 2) void ring_probe(Args a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int STEPS = 64;
    constexpr int NCHUNK = STEPS / G;
    constexpr int PER_CHUNK = G / 2;                // DMA pieces this wavefront issues per chunk
    // pieces this wavefront has issued after a chunk's own when it waits for that chunk:
    // everything up to the end of the ring for a strip's first chunk, one chunk less later
    // (the slots of chunk c - 1 are refilled only behind the barrier of chunk c)
    constexpr int AHEAD0 = (NSLOT / G - 1) * PER_CHUNK;
    constexpr int AHEAD = (NSLOT / G - 2) * PER_CHUNK;
    static_assert(NSLOT >= 2 * G, "ring of at least two chunks");
    (void)AHEAD;
    static_assert(NSLOT % G == 0 && STEPS % G == 0 && G % 2 == 0, "geometry");
    // DMA role: this wavefront fetches piece q (rows l = 16 q .. 16 q + 15) of the steps of
    // parity `par`; lane i -> row 16 q + i / 4, 16-byte chunk (i % 4) ^ swizzle
    const int q = wave & 3, par = wave >> 2;
    const int rl = lane >> 2;
    const int chunk = (lane & 3) ^ ((rl >> 2) & 3);
    const size_t row_bytes = a.stride * 8;
    const int l_of_lane = 16 * q + rl;
    const size_t lane_off = (a.linear ? (size_t)l_of_lane : (size_t)l_of_lane * 64) * row_bytes + chunk * 16;
    const size_t step_bytes = a.linear ? row_bytes * 64 : row_bytes;
    // reader role: baseline = wave (pair p, half h)
    const int p = wave >> 1, h = wave & 1;
    const int l16 = lane & 15;
    const int rd_off = (lane >> 4) * 1024 + l16 * 64 + ((p ^ ((l16 >> 2) & 3)) * 16) + 8 * h;

    if (a.stagger > 0 && (blockIdx.x & 8)) {
        for (int i = 0; i < a.stagger; i++) __builtin_amdgcn_s_sleep(127);
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_void *)lds;
    const int n_iter = (a.n_strips - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    float acc = 0.f;
    float d0 = 1.f, d1 = 2.f, d2 = 3.f, d3 = 4.f, e0 = 2.5f + lane, e1 = 1.5f;  // dummy chains (four independent)
    asm volatile("" : "+v"(e0), "+v"(e1));
    auto issue = [&](const char *strip_base, int j) {  // this wavefront's piece of step j
        if (a.linear == 2) return;  // arithmetic only
        const char *src = strip_base + lane_off + (size_t)j * step_bytes;
        // (inline assembly: the compiler orders every later LDS read behind an LDS-DMA it
        // knows about with s_waitcnt vmcnt(0), which would serialise the ring)
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (j % NSLOT) * 4096 + q * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst));
    };
    auto base_of = [&](int it) -> const char * {
        const int strip = strip_of(it * (int)gridDim.x + (int)blockIdx.x, a.n_strips, a.group);
        return (const char *)a.vis + (size_t)strip * 64;
    };
    if (n_iter <= 0) return;
    constexpr int K = NSLOT / G;  // chunks in the ring
    const char *cur = base_of(0);
    // prologue: the first NSLOT steps of the first strip
    for (int j = par; j < NSLOT; j += 2) issue(cur, j);
    // Schedule per chunk c (register double buffer, ONE barrier per chunk):
    //   wait: own LDS reads of chunk c done, own DMA pieces of chunk c + 1 landed
    //   barrier
    //   refill the slots of chunk c (steps NSLOT further on); read chunk c + 1 into registers
    //   arithmetic on chunk c
    float2 v[2][G];
    __builtin_amdgcn_s_waitcnt(0x0070 | (AHEAD0 & 15) | ((AHEAD0 >> 4) << 14));
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; g++) v[0][g] = *(const float2 *)(lds + g * 4096 + rd_off);
    for (int it = 0; it < n_iter; it++) {
        const bool more = it + 1 < n_iter;
        const char *nxt = more ? base_of(it + 1) : cur;
#pragma unroll
        for (int c = 0; c < NCHUNK; c++) {
            // pieces this wavefront issued after those of chunk c + 1: chunks c + 2 .. c + K - 1
            constexpr int AH = (K - 2) * PER_CHUNK;
            if (more || (c + K) * G <= STEPS)
                __builtin_amdgcn_s_waitcnt(0x0070 | (AH & 15) | ((AH >> 4) << 14));
            else
                __builtin_amdgcn_s_waitcnt(0x0070);
            __syncthreads();
#pragma unroll
            for (int g = par; g < G; g += 2) {
                const int j = c * G + g + NSLOT;
                if (j < STEPS)
                    issue(cur, j);
                else if (more)
                    issue(nxt, j - STEPS);
            }
            if (c + 1 < NCHUNK || more) {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int j = (c + 1) * G + g;  // (of the next strip when j >= STEPS: same slot formula)
                    v[(c + 1) & 1][g] = *(const float2 *)(lds + (j % NSLOT) * 4096 + rd_off);
                }
            }
#pragma unroll
            for (int g = 0; g < G; g++) {
                const float2 x = v[c & 1][g];
                acc += x.x;
                d0 += x.y;
#pragma unroll
                for (int k = 0; k < 48; k += 4) {
                    if (k >= a.amp_n) break;  // (wave-uniform; amp_n <= 48)
                    d0 = __builtin_amdgcn_fmed3f(d0, e0, x.y);
                    d1 = __builtin_amdgcn_fmed3f(d1, e1, x.y);
                    d2 = __builtin_amdgcn_fmed3f(d2, e0, x.x);
                    d3 = __builtin_amdgcn_fmed3f(d3, e1, x.x);
                }
            }
        }
        for (int k = 0; k < a.tail_n; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                d0 = __builtin_amdgcn_fmed3f(d0, e0, e1);
                d1 = __builtin_amdgcn_fmed3f(d1, e1, e0);
                d2 = __builtin_amdgcn_fmed3f(d2, e0, e1);
                d3 = __builtin_amdgcn_fmed3f(d3, e1, e0);
            }
            asm volatile("" : "+v"(e0), "+v"(e1));
        }
        cur = nxt;
    }
    double s = (double)acc;
    if (d0 + d1 + d2 + d3 == 123.456f) s += 1.0;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) atomicAdd(a.sum, s);
}

__global__ void fill(float2 *vis, int C, size_t stride, int B)
{
    const size_t n = (size_t)C * B;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / B), col = (int)(i % B);
        vis[(size_t)row * stride + col] = make_float2((float)((row * 7 + col * 13) % 251), (float)(col & 3));
    }
}

template <int NSLOT, int G>
void run(const char *name, float2 *vis, double *sum, int C, int B, int pad, int group, int amp_n, int tail_n,
         int linear = 0, int stagger = 0, int grid = 256)
{
    const size_t lds_bytes = (size_t)NSLOT * 4096;
    CHECK(hipFuncSetAttribute((const void *)ring_probe<NSLOT, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const size_t stride = (size_t)B + pad;
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, vis, C, stride, B);
    Args a{vis, sum, C, stride, B / 8, group, amp_n, tail_n, linear, stagger};
    CHECK(hipMemset(sum, 0, 8));
    hipLaunchKernelGGL((ring_probe<NSLOT, G>), dim3(grid), dim3(512), lds_bytes, 0, a);
    CHECK(hipDeviceSynchronize());
    double got;
    CHECK(hipMemcpy(&got, sum, 8, hipMemcpyDeviceToHost));
    double want = 0;
    for (int row = 0; row < C; row++)
        for (int col = 0; col < B; col++) want += (row * 7 + col * 13) % 251;
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((ring_probe<NSLOT, G>), dim3(grid), dim3(512), lds_bytes, 0, a);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((ring_probe<NSLOT, G>), dim3(grid), dim3(512), lds_bytes, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-40s ring=%2d G=%2d pad=%4d grp=%2d amp=%3d tail=%5d lin=%d stg=%d grid=%d : %.3f ms %.2f TB/s %s\n", name, NSLOT, G, pad,
           group, amp_n, tail_n, linear, stagger, grid, ms, (double)C * B * 8 / ms / 1e9, got == want ? "sum ok" : "SUM WRONG");
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int C = 4096, B = argc > 1 ? atoi(argv[1]) : 32768, PADMAX = 2048;
    printf("B = %d baselines: %.0f MiB\n", B, (double)C * B * 8 / 1048576);
    const size_t bytes = (size_t)C * (B + PADMAX) * 8;
    float2 *vis; double *sum;
    CHECK(hipMalloc(&vis, bytes)); CHECK(hipMalloc(&sum, 8));
    // 58 VALU instructions per sample: 25 in the amplitude (per step), 33 x 64 afterwards
    run<32, 8>("loads only", vis, sum, C, B, 32, 8, 0, 0);
    run<32, 4>("loads only G 4", vis, sum, C, B, 32, 8, 0, 0);
    run<32, 8>("VALU only: amp 24 + tail 2112", vis, sum, C, B, 0, 8, 24, 2112, 2);
    run<32, 8>("VALU only: amp 44 + tail 832", vis, sum, C, B, 0, 8, 44, 832, 2);
    run<32, 8>("VALU only: amp 32 + tail 768", vis, sum, C, B, 0, 8, 32, 768, 2);
    run<32, 8>("amp 24 + tail 2112", vis, sum, C, B, 32, 8, 24, 2112);
    run<32, 8>("amp 44 + tail 832", vis, sum, C, B, 32, 8, 44, 832);
    run<32, 4>("amp 44 + tail 832 G 4", vis, sum, C, B, 32, 8, 44, 832);
    run<32, 16>("amp 44 + tail 832 G 16", vis, sum, C, B, 32, 8, 44, 832);
    run<16, 4>("amp 44 + tail 832 ring 16 G 4", vis, sum, C, B, 32, 8, 44, 832);
    run<32, 8>("amp 32 + tail 768", vis, sum, C, B, 32, 8, 32, 768);
    run<32, 4>("amp 32 + tail 768 G 4", vis, sum, C, B, 32, 8, 32, 768);
    run<32, 8>("amp 44 + tail 832 stagger 1", vis, sum, C, B, 32, 8, 44, 832, 0, 1);
    run<32, 8>("amp 44 + tail 832 pad 8", vis, sum, C, B, 8, 8, 44, 832);
    run<32, 8>("amp 44 + tail 832 pad 0", vis, sum, C, B, 0, 8, 44, 832);
    return 0;
}
