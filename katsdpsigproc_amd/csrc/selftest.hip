// Self-test entry points: expose the arithmetic building blocks of the kernels so
// that the test-suite can compare them with IEEE / numpy results (exhaustively for
// the restricted-range square root). Not part of any reference interface.
#include "rank.h"

__global__ void selftest_sqrt12_kernel(float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ksp_sqrt_1_2(__uint_as_float(0x3f800000u + (unsigned)i));
}

__global__ void selftest_abs_kernel(const float *re, const float *im, float *out, int n)
{
    // the forms the loaders pick per batch, here per pair of elements: two samples that both
    // qualify for the short division go through its packed form, a single one through the
    // scalar form, anything else through the general one
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const bool ok0 = ksp_abs_range_key(re[i], im[i]) & KSP_ABS_RANGE_BIT;
    const bool ok1 = i + 1 < n && (ksp_abs_range_key(re[i + 1], im[i + 1]) & KSP_ABS_RANGE_BIT);
    if (ok0 && ok1) {
        ksp_abs_c64_inrange_x2(re[i], im[i], re[i + 1], im[i + 1], out[i], out[i + 1]);
        return;
    }
    out[i] = ok0 ? ksp_abs_c64_inrange(re[i], im[i]) : ksp_abs_c64(re[i], im[i]);
    if (i + 1 < n)
        out[i + 1] = ok1 ? ksp_abs_c64_inrange(re[i + 1], im[i + 1]) : ksp_abs_c64(re[i + 1], im[i + 1]);
}

extern "C" int ksp_selftest_sqrt12(int device, void *stream, float *out, int n)
{
    KSP_REQUIRE(out != nullptr && n >= 0 && n <= (1 << 23) + 1, "bad arguments");
    KSP_CHECK(hipSetDevice(device));
    if (n == 0) return 0;
    hipLaunchKernelGGL(selftest_sqrt12_kernel, dim3(ksp_divup(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, out, n);
    KSP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ksp_selftest_abs(int device, void *stream, const float *re, const float *im,
                                float *out, int n)
{
    KSP_REQUIRE(re != nullptr && im != nullptr && out != nullptr && n >= 0, "bad arguments");
    KSP_CHECK(hipSetDevice(device));
    if (n == 0) return 0;
    hipLaunchKernelGGL(selftest_abs_kernel, dim3(ksp_divup(n, 512)), dim3(256), 0,
                       (hipStream_t)stream, re, im, out, n);
    KSP_LAUNCH_CHECK();
    return 0;
}

// ---- rank library (rank.h), the counterpart of reference test/test_rank.mako:37-113 ----
// out[q] = number of data values strictly below q, 0 <= q < m, counted by one
// 256-thread workgroup whose threads hold VT values each (NaN padding never counts).
template <int VT>
__global__ __launch_bounds__(KSP_RANK_THREADS) void selftest_rank_kernel(
    const float *__restrict__ data, int n, int *__restrict__ out, int m)
{
    __shared__ RankScratch scratch;
    const int t = threadIdx.x;
    float v[VT];
#pragma unroll
    for (int i = 0; i < VT; i++) {
        const int c = i * KSP_RANK_THREADS + t;
        v[i] = (c < n) ? data[c] : __builtin_nanf("");
    }
    for (int q = 0; q < m; q++) {
        const unsigned pivot = __float_as_uint((float)q);
        int c = 0;
#pragma unroll
        for (int i = 0; i < VT; i++) c += __float_as_uint(v[i]) < pivot;
        c = block_sum(c, &scratch);
        if (t == 0) out[q] = c;
    }
}

// out[0] = smallest, out[1] = largest non-NaN value (NaN if there is none)
template <int VT>
__global__ __launch_bounds__(KSP_RANK_THREADS) void selftest_minmax_kernel(
    const float *__restrict__ data, int n, float *__restrict__ out)
{
    __shared__ RankScratch scratch;
    const int t = threadIdx.x;
    float v[VT];
#pragma unroll
    for (int i = 0; i < VT; i++) {
        const int c = i * KSP_RANK_THREADS + t;
        v[i] = (c < n) ? data[c] : __builtin_nanf("");
    }
    const float lo = block_fmin(v, &scratch);
    const float hi = block_fmax(v, &scratch);
    if (t == 0) {
        out[0] = lo;
        out[1] = hi;
    }
}

extern "C" int ksp_selftest_rank(int device, void *stream, const float *data, int *out, int n,
                                 int m)
{
    KSP_REQUIRE(data != nullptr && out != nullptr && n >= 0 && n <= 8 * KSP_RANK_THREADS && m >= 0,
                "bad arguments");
    KSP_CHECK(hipSetDevice(device));
    hipLaunchKernelGGL(selftest_rank_kernel<8>, dim3(1), dim3(KSP_RANK_THREADS), 0,
                       (hipStream_t)stream, data, n, out, m);
    KSP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ksp_selftest_minmax(int device, void *stream, const float *data, float *out, int n)
{
    KSP_REQUIRE(data != nullptr && out != nullptr && n >= 1 && n <= 8 * KSP_RANK_THREADS,
                "bad arguments");
    KSP_CHECK(hipSetDevice(device));
    hipLaunchKernelGGL(selftest_minmax_kernel<8>, dim3(1), dim3(KSP_RANK_THREADS), 0,
                       (hipStream_t)stream, data, n, out);
    KSP_LAUNCH_CHECK();
    return 0;
}
