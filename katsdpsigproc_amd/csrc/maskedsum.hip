// maskedsum: out[col] = sum_row mask[row] * in[row][col] (stands in for reference
// maskedsum.mako:38-68, which runs one thread per column over all rows).
//
// On MI355X one thread per column leaves most of the chip idle for the matrix
// widths this is used with (4096 columns = 64 waves for 256 CUs), so the rows are
// split as well: a 1024-thread workgroup covers 16 columns (one 128-byte segment
// of complex64 per row) x 64 row phases; each thread accumulates rows
// r = phase, phase + 64, ... with fma in float32, and the 64 partial sums of a
// column are combined in LDS in a fixed tree order, so the result is deterministic
// (but, like any float32 sum, order-dependent: the reference test allows
// rtol = 1e-6, test/test_maskedsum.py:67). HBM-bound at 8 bytes per element.
#include "ksp_common.h"

#define MS_COLS 16
#define MS_PHASES 64

template <bool USE_AMP>
__global__ __launch_bounds__(MS_COLS *MS_PHASES) void maskedsum_kernel(
    const float2 *__restrict__ in, const float *__restrict__ mask, void *__restrict__ out,
    int in_stride, int n_rows, int n_cols)
{
    __shared__ float2 part[MS_PHASES][MS_COLS + 1];
    const int lc = threadIdx.x % MS_COLS;
    const int phase = threadIdx.x / MS_COLS;
    const int col = blockIdx.x * MS_COLS + lc;
    float2 acc = make_float2(0.0f, 0.0f);
    if (col < n_cols) {
        const float2 *p = in + col;
#pragma unroll 4
        for (int row = phase; row < n_rows; row += MS_PHASES) {
            const float2 c = p[(size_t)row * in_stride];
            const float m = mask[row];
            if (USE_AMP) {
                // reference: fma(mask, sqrt(x*x + y*y), acc)  (maskedsum.mako:60)
                const float a = __builtin_sqrtf(__fadd_rn(__fmul_rn(c.x, c.x), __fmul_rn(c.y, c.y)));
                acc.x = __fmaf_rn(m, a, acc.x);
            } else {
                acc.x = __fmaf_rn(m, c.x, acc.x);
                acc.y = __fmaf_rn(m, c.y, acc.y);
            }
        }
    }
    part[phase][lc] = acc;
    __syncthreads();
#pragma unroll
    for (int half = MS_PHASES / 2; half > 0; half >>= 1) {
        if (phase < half) {
            float2 a = part[phase][lc], b = part[phase + half][lc];
            a.x = __fadd_rn(a.x, b.x);
            a.y = __fadd_rn(a.y, b.y);
            part[phase][lc] = a;
        }
        __syncthreads();
    }
    if (phase == 0 && col < n_cols) {
        if (USE_AMP)
            ((float *)out)[col] = part[0][lc].x;
        else
            ((float2 *)out)[col] = part[0][lc];
    }
}

extern "C" int ksp_maskedsum_float(int device, void *stream, const void *in, const float *mask,
                                   void *out, int in_stride, int n_rows, int n_cols,
                                   int use_amplitudes)
{
    KSP_REQUIRE(in != nullptr && mask != nullptr && out != nullptr, "NULL buffer");
    KSP_REQUIRE(n_rows >= 0 && n_cols >= 0 && in_stride >= n_cols, "bad shape");
    if (n_cols == 0) return 0;
    KSP_CHECK(hipSetDevice(device));
    dim3 grid(ksp_divup(n_cols, MS_COLS));
    if (use_amplitudes)
        hipLaunchKernelGGL(maskedsum_kernel<true>, grid, dim3(MS_COLS * MS_PHASES), 0,
                           (hipStream_t)stream, (const float2 *)in, mask, out, in_stride, n_rows,
                           n_cols);
    else
        hipLaunchKernelGGL(maskedsum_kernel<false>, grid, dim3(MS_COLS * MS_PHASES), 0,
                           (hipStream_t)stream, (const float2 *)in, mask, out, in_stride, n_rows,
                           n_cols);
    KSP_LAUNCH_CHECK();
    return 0;
}
