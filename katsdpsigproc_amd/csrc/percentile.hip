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
#include "bitplane.h"
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

__device__ __forceinline__ float key_to_float(unsigned k)
{
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

// ----------------------------------------------------------------------------
// Rows of 1025 .. 4096 columns: ONE WAVEFRONT per row, lane l holds columns
// [64 l, 64 l + 64) of the range. Values become 32-bit keys that order like the floats
// (sign bit flipped for positive, all bits for negative values), the keys are
// transposed into 32 bit planes per lane, and the three order statistics come from
// three bit-plane searches (bitplane.h) -- no LDS, no barriers.
template <bool IS_AMP>
__global__ __launch_bounds__(256) void percentile5_wave_kernel(const void *__restrict__ in,
                                                               float *__restrict__ out, int rows,
                                                               int in_stride, int out_stride,
                                                               int first_col, int n_cols,
                                                               int vec_ok)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;  // whole wavefronts leave together
    // lane l holds columns [64 l, 64 l + 64): 256 (512) contiguous bytes per lane, read
    // with 16-byte loads when the row allows it (measured faster than the same bytes
    // interleaved over the lanes, and much faster than 4-byte loads)
    const size_t base = (size_t)row * in_stride + first_col + lane * 64;
    float amp[64];
    const bool full = lane * 64 + 64 <= n_cols;
    if (vec_ok && full) {
        if (IS_AMP) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float4 q = *(const float4 *)((const float *)in + base + 4 * i);
                amp[4 * i] = q.x;
                amp[4 * i + 1] = q.y;
                amp[4 * i + 2] = q.z;
                amp[4 * i + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const float4 q = *(const float4 *)((const float2 *)in + base + 2 * i);
                amp[2 * i] = ksp_abs_c64(q.x, q.y);
                amp[2 * i + 1] = ksp_abs_c64(q.z, q.w);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 64; i++) {
            amp[i] = 0.0f;
            if (lane * 64 + i < n_cols) {
                if (IS_AMP)
                    amp[i] = ((const float *)in)[base + i];
                else {
                    const float2 z = ((const float2 *)in)[base + i];
                    amp[i] = ksp_abs_c64(z.x, z.y);
                }
            }
        }
    }
    unsigned key[64];
    unsigned kmin = 0xffffffffu, kmax = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        unsigned k = 0xffffffffu;  // columns beyond the range: above every real key
        if (full || lane * 64 + i < n_cols) {
            const unsigned u = __float_as_uint(amp[i]);
            k = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
            kmin = min(kmin, k);
            kmax = max(kmax, k);
        }
        key[i] = k;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, off, 64));
        kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, off, 64));
    }
    unsigned hi[32], lo[32];  // inverted planes of key bits 16..31 / 0..15
#pragma unroll
    for (int i = 0; i < 32; i++) {
        hi[i] = ~__builtin_amdgcn_perm(key[2 * i + 1], key[2 * i], 0x07060302u);
        lo[i] = ~__builtin_amdgcn_perm(key[2 * i + 1], key[2 * i], 0x05040100u);
    }
    transpose_bits32(hi);
    transpose_bits32(lo);
    auto plane = [&](int bit, int half) __attribute__((always_inline)) -> unsigned {
        return bit >= 16 ? hi[16 * half + bit - 16] : lo[16 * half + bit];
    };
    // The three order statistics are searched TOGETHER, bit pair by bit pair: the
    // searches are independent, so each one's cross-lane reduction (a chain of dependent
    // DPP steps) runs in the shadow of the other two instead of three chains end to end.
    PlaneSearch<decltype(plane)> s25((n_cols - 1) / 4, plane);
    PlaneSearch<decltype(plane)> s75(((n_cols - 1) * 3) / 4, plane);
    PlaneSearch<decltype(plane)> s50((n_cols - 1) / 2, plane);
// (one bit per step and search: measured 23 % faster than two-bit steps)
#define KSP_P5_STEP(HI, LO)         \
    s25.template step1<HI>();       \
    s75.template step1<HI>();       \
    s50.template step1<HI>();       \
    s25.template step1<LO>();       \
    s75.template step1<LO>();       \
    s50.template step1<LO>()
    KSP_P5_STEP(31, 30);
    KSP_P5_STEP(29, 28);
    KSP_P5_STEP(27, 26);
    KSP_P5_STEP(25, 24);
    KSP_P5_STEP(23, 22);
    KSP_P5_STEP(21, 20);
    KSP_P5_STEP(19, 18);
    KSP_P5_STEP(17, 16);
    KSP_P5_STEP(15, 14);
    KSP_P5_STEP(13, 12);
    KSP_P5_STEP(11, 10);
    KSP_P5_STEP(9, 8);
    KSP_P5_STEP(7, 6);
    KSP_P5_STEP(5, 4);
    KSP_P5_STEP(3, 2);
    KSP_P5_STEP(1, 0);
#undef KSP_P5_STEP
    const float p25 = key_to_float(s25.prefix);
    const float p75 = key_to_float(s75.prefix);
    const float p50 = key_to_float(s50.prefix);
    if (lane == 0) {
        out[0 * (size_t)out_stride + row] = key_to_float(kmin);
        out[1 * (size_t)out_stride + row] = key_to_float(kmax);
        out[2 * (size_t)out_stride + row] = p25;
        out[3 * (size_t)out_stride + row] = p75;
        out[4 * (size_t)out_stride + row] = p50;
    }
}

template <bool IS_AMP>
static int launch_percentile(hipStream_t s, const void *in, float *out, int rows, int in_stride,
                             int out_stride, int first_col, int n_cols)
{
    if (n_cols > 1024 && n_cols <= 4096) {
        // 16-byte loads: every lane's first element 16-byte aligned
        const int per16 = IS_AMP ? 4 : 2;
        const int vec_ok = (in_stride % per16 == 0) && (first_col % per16 == 0) && ((uintptr_t)in % 16 == 0);
        hipLaunchKernelGGL((percentile5_wave_kernel<IS_AMP>), dim3(ksp_divup(rows, 4)), dim3(256), 0, s,
                           in, out, rows, in_stride, out_stride, first_col, n_cols, vec_ok);
        KSP_LAUNCH_CHECK();
        return 0;
    }
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
