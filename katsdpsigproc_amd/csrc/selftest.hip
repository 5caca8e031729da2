// Self-test entry points: expose the arithmetic building blocks of the kernels so
// that the test-suite can compare them with IEEE / numpy results (exhaustively for
// the restricted-range square root). Not part of any reference interface.
#include "ksp_common.h"

__global__ void selftest_sqrt12_kernel(float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ksp_sqrt_1_2(__uint_as_float(0x3f800000u + (unsigned)i));
}

__global__ void selftest_abs_kernel(const float *re, const float *im, float *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = ksp_abs_c64(re[i], im[i]);
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
    hipLaunchKernelGGL(selftest_abs_kernel, dim3(ksp_divup(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, re, im, out, n);
    KSP_LAUNCH_CHECK();
    return 0;
}
