// percentile5: per row, [min, max, 25 %, 75 %, 50 %] ("lower" element, no
// interpolation) of |x| over a column range (stands in for reference
// percentile.mako:115-140).
//
// One 256-thread workgroup owns one row; the row lives in registers (VT values per
// thread, strided so that global reads are fully coalesced), and the three rank
// queries share one 31-pass bit-wise search (rank.h). Complex input is reduced with
// numpy's own |z| formula, so the result is bit-identical to
// np.percentile(np.abs(x), ..., method="lower") -- the reference kernel ranks
// re^2+im^2 and is only 1e-6-close (reference test/test_percentile.py:86-90).
// HBM-bound at 4 (8 for complex) bytes per element once the search is hidden by
// other resident workgroups.
#include "rank.h"

template <int VT, bool IS_AMP>
__global__ __launch_bounds__(KSP_RANK_THREADS) void percentile5_kernel(
    const void *__restrict__ in, float *__restrict__ out, int rows, int in_stride, int out_stride,
    int first_col, int n_cols)
{
    __shared__ RankScratch scratch;
    const int row = blockIdx.x;
    const int t = threadIdx.x;
    float v[VT];
#pragma unroll
    for (int i = 0; i < VT; i++) {
        const int c = i * KSP_RANK_THREADS + t;
        float a = __builtin_nanf("");
        if (c < n_cols) {
            const size_t idx = (size_t)row * in_stride + first_col + c;
            if (IS_AMP)
                a = ((const float *)in)[idx];
            else {
                const float2 z = ((const float2 *)in)[idx];
                a = ksp_abs_c64(z.x, z.y);
            }
        }
        v[i] = a;
    }
    const float lo = block_fmin(v, &scratch);
    const float hi = block_fmax(v, &scratch);
    float p25, p75, p50;
    block_select3(v, (n_cols - 1) / 4, ((n_cols - 1) * 3) / 4, (n_cols - 1) / 2, p25, p75, p50,
                  &scratch);
    if (t == 0) {
        out[0 * (size_t)out_stride + row] = lo;
        out[1 * (size_t)out_stride + row] = hi;
        out[2 * (size_t)out_stride + row] = p25;
        out[3 * (size_t)out_stride + row] = p75;
        out[4 * (size_t)out_stride + row] = p50;
    }
}

template <bool IS_AMP>
static int launch_percentile(hipStream_t s, const void *in, float *out, int rows, int in_stride,
                             int out_stride, int first_col, int n_cols)
{
    const int vt = ksp_divup(n_cols, KSP_RANK_THREADS);
#define KSP_P5(VT)                                                                              \
    hipLaunchKernelGGL((percentile5_kernel<VT, IS_AMP>), dim3(rows), dim3(KSP_RANK_THREADS), 0, \
                       s, in, out, rows, in_stride, out_stride, first_col, n_cols)
    if (vt <= 1)
        KSP_P5(1);
    else if (vt <= 2)
        KSP_P5(2);
    else if (vt <= 4)
        KSP_P5(4);
    else if (vt <= 8)
        KSP_P5(8);
    else if (vt <= 16)
        KSP_P5(16);
    else if (vt <= 24)
        KSP_P5(24);
    else if (vt <= 32)
        KSP_P5(32);
    else if (vt <= 48)
        KSP_P5(48);
    else if (vt <= 64)
        KSP_P5(64);
    else {
        ksp_set_error("ksp_percentile5_float: %d columns exceed the supported maximum of %d",
                      n_cols, 64 * KSP_RANK_THREADS);
        return (int)hipErrorInvalidValue;
    }
#undef KSP_P5
    KSP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ksp_percentile5_float(int device, void *stream, const void *in, float *out,
                                     int rows, int in_stride, int out_stride, int first_col,
                                     int n_cols, int is_amplitude)
{
    KSP_REQUIRE(in != nullptr && out != nullptr, "NULL buffer");
    KSP_REQUIRE(rows >= 0 && n_cols > 0 && first_col >= 0, "bad shape");
    KSP_REQUIRE(first_col + n_cols <= in_stride, "column range exceeds the row stride");
    KSP_REQUIRE(out_stride >= rows, "out_stride smaller than rows");
    if (rows == 0) return 0;
    KSP_CHECK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    return is_amplitude
               ? launch_percentile<true>(s, in, out, rows, in_stride, out_stride, first_col, n_cols)
               : launch_percentile<false>(s, in, out, rows, in_stride, out_stride, first_col, n_cols);
}
