// Microbenchmark for a two-kernel design: how fast can [C][B] complex64 be turned into
// amplitude-transposed [B][C] float32 (8 B read + 4 B written per sample)? Diagnostic only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ float amp(float re, float im)
{
    // same operation count as the product's |z| (IEEE divide, fma, restricted sqrt)
    const unsigned ur = __float_as_uint(re) & 0x7fffffffu, ui = __float_as_uint(im) & 0x7fffffffu;
    const unsigned umx = max(ur, ui), umn = min(ur, ui);
    const float mx = __uint_as_float(umx), mn = __uint_as_float(umn);
    const float r = __fdiv_rn(mn, __uint_as_float(max(umx, 1u)));
    const float t = __fmaf_rn(r, r, 1.0f);
    const float q = __builtin_amdgcn_rsqf(t);
    const float g = t * q, h = 0.5f * q;
    const float s = __fmaf_rn(h, __fmaf_rn(-g, g, t), g);
    return mx * s;
}

// tile: TC channels x 64 baselines per 256-thread workgroup
template <int TC>
__global__ __launch_bounds__(256) void amp_t(const float2 *__restrict__ vis, float *__restrict__ out, int C, int B)
{
    __shared__ float tile[64][TC + 1];
    const int b0 = blockIdx.x * 64, c0 = blockIdx.y * TC;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = w; r < TC; r += 4) {
        const float2 z = vis[(size_t)(c0 + r) * B + b0 + lane];  // 512 B per wavefront
        tile[lane][r] = amp(z.x, z.y);
    }
    __syncthreads();
    // write [baseline][channel]: TC consecutive channels per baseline
    for (int i = threadIdx.x; i < 64 * TC; i += 256) {
        const int b = i / TC, c = i % TC;
        out[(size_t)(b0 + b) * C + c0 + c] = tile[b][c];
    }
}

int main()
{
    const int C = 4096, B = 32768;
    float2 *vis; float *out;
    CHECK(hipMalloc(&vis, (size_t)C * B * 8)); CHECK(hipMalloc(&out, (size_t)C * B * 4));
    CHECK(hipMemset(vis, 1, (size_t)C * B * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
#define RUN(TC) do { \
    hipLaunchKernelGGL(amp_t<TC>, dim3(B / 64, C / TC), dim3(256), 0, 0, vis, out, C, B); \
    CHECK(hipDeviceSynchronize()); CHECK(hipEventRecord(e0)); \
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(amp_t<TC>, dim3(B / 64, C / TC), dim3(256), 0, 0, vis, out, C, B); \
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10; \
    printf("tile %3d ch x 64 bl: %.3f ms  (%.2f TB/s of 12 B/sample)\n", TC, ms, (double)C * B * 12 / ms / 1e9); } while (0)
    RUN(64); RUN(128); RUN(256);
    return 0;
}
