// Shared host/device helpers for the gfx950 kernels. Wavefront = 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/katsdpsigproc_hip.h"

#define KSP_WAVE 64

void ksp_set_error(const char *fmt, ...);

#define KSP_CHECK(expr)                                                                    \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            ksp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                       \
            (void)hipGetLastError(); /* reported here: do not leave it for a later launch check */ \
            return (int)_e;                                                                \
        }                                                                                  \
    } while (0)

#define KSP_REQUIRE(cond, msg)                                              \
    do {                                                                    \
        if (!(cond)) {                                                      \
            ksp_set_error("invalid argument: %s (%s)", msg, #cond);         \
            return (int)hipErrorInvalidValue;                               \
        }                                                                   \
    } while (0)

#define KSP_LAUNCH_CHECK() KSP_CHECK(hipGetLastError())

static inline int ksp_divup(int a, int b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// Correctly rounded float32 square root for 1 <= x <= 2 (NaN passes through): one
// Newton step on x * rsq(x) with the residual taken exactly by an fma. The general
// expansion the compiler emits spends a dozen instructions on scaling and special
// cases that cannot occur in this interval; this one is checked against IEEE sqrt
// for EVERY float32 in [1, 2] (tests/test_gpu_ops.py, ksp_selftest_sqrt12).
__device__ __forceinline__ float ksp_sqrt_1_2(float x)
{
    const float q = __builtin_amdgcn_rsqf(x);
    const float g = __fmul_rn(x, q);
    const float h = __fmul_rn(0.5f, q);
    const float r = __fmaf_rn(-g, g, x);
    return __fmaf_rn(h, r, g);
}

// numpy's complex64 abs: mx * sqrt(fma(r, r, 1)), r = mn / mx, with IEEE
// division and square root (pinned by tests/golden abs probe; the reference
// host path computes np.abs(vis), rfi/host.py:137).
// |x| bit patterns order like unsigned integers with the NaNs on top, so max/min on
// the patterns give mx/mn with a NaN operand propagating by itself.
__device__ __forceinline__ float ksp_abs_c64(float re, float im)
{
    const unsigned ur = __float_as_uint(re) & 0x7fffffffu;
    const unsigned ui = __float_as_uint(im) & 0x7fffffffu;
    const unsigned umx = max(ur, ui), umn = min(ur, ui);
    const float mx = __uint_as_float(umx), mn = __uint_as_float(umn);
    // divide by at least the smallest denormal: 0 / 0 becomes 0 / tiny = 0 (-> |0| = 0)
    const float r = __fdiv_rn(mn, __uint_as_float(max(umx, 1u)));
    const float t = __fmaf_rn(r, r, 1.0f);  // in [1, 2]
    float a = __fmul_rn(mx, ksp_sqrt_1_2(t));
    // (inf, inf) and (inf, NaN) arrive here as NaN; hypot says inf
    if (umn == 0x7f800000u) a = __builtin_inff();
    return a;
}

// The same value for max(|re|, |im|) in [2^-63, 2^65) with the division written out:
// the instruction sequence the compiler emits for an IEEE division (reciprocal, one
// Newton step on it, quotient, two residual corrections) without the operand scaling
// before it and the special-case fix-up after it, four half-rate instructions that do
// nothing in this range -- v_div_scale leaves operands alone unless the denominator or
// 1/denominator is subnormal or the quotient is below 2^-126 (numerator below 2^-103),
// and a quotient that small only has to stay small: r < 2^-12 gives fma(r, r, 1) = 1
// either way. No zero, infinity or NaN gets here, so the guards of the general form go
// too. ksp_abs_in_range() says whether a sample qualifies: callers accumulate the keys
// of a batch with AND and take this form when bit 30 survives in every lane.
__device__ __forceinline__ unsigned ksp_abs_range_key(float re, float im)
{
    const unsigned ur = __float_as_uint(re) & 0x7fffffffu;
    const unsigned ui = __float_as_uint(im) & 0x7fffffffu;
    // exponent field e -> e + 64: bit 30 is set exactly for 64 <= e < 192
    return max(ur, ui) + (64u << 23);
}
constexpr unsigned KSP_ABS_RANGE_BIT = 1u << 30;

__device__ __forceinline__ float ksp_abs_c64_inrange(float re, float im)
{
    const unsigned ur = __float_as_uint(re) & 0x7fffffffu;
    const unsigned ui = __float_as_uint(im) & 0x7fffffffu;
    const float mx = __uint_as_float(max(ur, ui)), mn = __uint_as_float(min(ur, ui));
    const float y0 = __builtin_amdgcn_rcpf(mx);
    const float y = __fmaf_rn(__fmaf_rn(-mx, y0, 1.0f), y0, y0);
    const float q0 = __fmul_rn(mn, y);
    const float q1 = __fmaf_rn(__fmaf_rn(-mx, q0, mn), y, q0);
    const float r = __fmaf_rn(__fmaf_rn(-mx, q1, mn), y, q1);
    const float t = __fmaf_rn(r, r, 1.0f);  // in [1, 2]
    return __fmul_rn(mx, ksp_sqrt_1_2(t));
}

// Wave-wide (64-lane) reductions with every lane receiving the result.
// Wavefront vote as a scalar mask: v_cmp writes the lane mask straight into a scalar
// register pair, so "any lane" is one scalar compare (HIP's __any/__ballot go through
// a v_cndmask/v_cmp pair first).
__device__ __forceinline__ unsigned long long ksp_ballot(bool x)
{
    return __builtin_amdgcn_ballot_w64(x);
}
__device__ __forceinline__ bool ksp_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0; }

// Two of them at once with packed float32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: two IEEE
// operations per lane and instruction, same results): everything but the two reciprocals,
// the two reciprocal square roots and the integer min/max is shared by the pair.
typedef float ksp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ksp_abs_c64_inrange_x2(float re0, float im0, float re1, float im1,
                                                       float &a0, float &a1)
{
    const unsigned ur0 = __float_as_uint(re0) & 0x7fffffffu, ui0 = __float_as_uint(im0) & 0x7fffffffu;
    const unsigned ur1 = __float_as_uint(re1) & 0x7fffffffu, ui1 = __float_as_uint(im1) & 0x7fffffffu;
    const ksp_f32x2 mx = {__uint_as_float(max(ur0, ui0)), __uint_as_float(max(ur1, ui1))};
    const ksp_f32x2 mn = {__uint_as_float(min(ur0, ui0)), __uint_as_float(min(ur1, ui1))};
    const ksp_f32x2 one = {1.0f, 1.0f}, half = {0.5f, 0.5f};
    const ksp_f32x2 y0 = {__builtin_amdgcn_rcpf(mx.x), __builtin_amdgcn_rcpf(mx.y)};
    const ksp_f32x2 y = __builtin_elementwise_fma(__builtin_elementwise_fma(-mx, y0, one), y0, y0);
    const ksp_f32x2 q0 = mn * y;
    const ksp_f32x2 q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(-mx, q0, mn), y, q0);
    const ksp_f32x2 r = __builtin_elementwise_fma(__builtin_elementwise_fma(-mx, q1, mn), y, q1);
    const ksp_f32x2 t = __builtin_elementwise_fma(r, r, one);  // in [1, 2]
    // (ksp_sqrt_1_2 on both)
    const ksp_f32x2 q = {__builtin_amdgcn_rsqf(t.x), __builtin_amdgcn_rsqf(t.y)};
    const ksp_f32x2 g = t * q;
    const ksp_f32x2 h = half * q;
    const ksp_f32x2 e = __builtin_elementwise_fma(-g, g, t);
    const ksp_f32x2 sq = __builtin_elementwise_fma(h, e, g);
    const ksp_f32x2 a = mx * sq;
    a0 = a.x;
    a1 = a.y;
}

// |z| of the N pairs of visibilities a lane holds (v.x + j v.y, v.z + j v.w): the short
// division when every magnitude of the batch, in every lane of the wavefront, is an
// ordinary one, otherwise the general form for all of them.
// PACKED selects the two-at-once form of the short division (measured 3.6 % faster for the
// kernel without input flags, 1 % slower when the other wavefront of the SIMD runs the
// sorted-window median, i.e. with input flags).
// Returns whether the batch took the short division (then every amplitude is finite).
// `nan_watch` accumulates the largest bit pattern among the amplitudes of batches that took
// the general form (patterns above 0x7f800000 are NaNs).
template <int N, bool PACKED = true>
__device__ __forceinline__ bool ksp_abs_c64_batch(const float4 (&v)[N], float (&amp)[N][2],
                                                  unsigned &nan_watch)
{
    unsigned key = ~0u;
#pragma unroll
    for (int u = 0; u < N; u++)
        key &= ksp_abs_range_key(v[u].x, v[u].y) & ksp_abs_range_key(v[u].z, v[u].w);
    const bool ordinary = !ksp_any((key & KSP_ABS_RANGE_BIT) == 0);
    if (ordinary) {
#pragma unroll
        for (int u = 0; u < N; u++) {
            if constexpr (PACKED) {
                ksp_abs_c64_inrange_x2(v[u].x, v[u].y, v[u].z, v[u].w, amp[u][0], amp[u][1]);
            } else {
                amp[u][0] = ksp_abs_c64_inrange(v[u].x, v[u].y);
                amp[u][1] = ksp_abs_c64_inrange(v[u].z, v[u].w);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < N; u++) {
            amp[u][0] = ksp_abs_c64(v[u].x, v[u].y);
            amp[u][1] = ksp_abs_c64(v[u].z, v[u].w);
            // (only here can an amplitude be NaN; see the callers' `umax`)
            nan_watch = max(nan_watch, max(__float_as_uint(amp[u][0]), __float_as_uint(amp[u][1])));
        }
    }
    return ordinary;
}

// |z| of N visibilities of a lane (e.g. N consecutive rows of its baseline): the short
// division in packed pairs when every magnitude, in every lane, is an ordinary one,
// otherwise the general form for all of them.
template <int N>
__device__ __forceinline__ void ksp_abs_c64_rows(const float2 (&z)[N], float (&a)[N])
{
    unsigned key = ~0u;
#pragma unroll
    for (int k = 0; k < N; k++) key &= ksp_abs_range_key(z[k].x, z[k].y);
    if (!ksp_any((key & KSP_ABS_RANGE_BIT) == 0)) {
#pragma unroll
        for (int k = 0; k + 1 < N; k += 2)
            ksp_abs_c64_inrange_x2(z[k].x, z[k].y, z[k + 1].x, z[k + 1].y, a[k], a[k + 1]);
        if (N & 1) a[N - 1] = ksp_abs_c64_inrange(z[N - 1].x, z[N - 1].y);
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) a[k] = ksp_abs_c64(z[k].x, z[k].y);
    }
}


// Wavefront reductions on the DPP network (no LDS round trip, result wave-uniform):
// four row shifts leave each row's total in its lane 15, row_bcast:15 / row_bcast:31
// carry the totals across rows into lane 63.
#define KSP_DPP(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, 0xf, true)
__device__ __forceinline__ int ksp_wave_sum_dpp(int v)
{
    v += KSP_DPP(v, 0x111, 0xf);  // row_shr:1
    v += KSP_DPP(v, 0x112, 0xf);  // row_shr:2
    v += KSP_DPP(v, 0x114, 0xf);  // row_shr:4
    v += KSP_DPP(v, 0x118, 0xf);  // row_shr:8
    v += KSP_DPP(v, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    v += KSP_DPP(v, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
// Two wavefront sums at once: the two DPP chains are interleaved so that each
// instruction fills the wait states the other chain needs between dependent DPP steps.
__device__ __forceinline__ void ksp_wave_sum2_dpp(int &a, int &b)
{
    a += KSP_DPP(a, 0x111, 0xf);
    b += KSP_DPP(b, 0x111, 0xf);
    a += KSP_DPP(a, 0x112, 0xf);
    b += KSP_DPP(b, 0x112, 0xf);
    a += KSP_DPP(a, 0x114, 0xf);
    b += KSP_DPP(b, 0x114, 0xf);
    a += KSP_DPP(a, 0x118, 0xf);
    b += KSP_DPP(b, 0x118, 0xf);
    a += KSP_DPP(a, 0x142, 0xa);
    b += KSP_DPP(b, 0x142, 0xa);
    a += KSP_DPP(a, 0x143, 0xc);
    b += KSP_DPP(b, 0x143, 0xc);
    a = __builtin_amdgcn_readlane(a, 63);
    b = __builtin_amdgcn_readlane(b, 63);
}

// Inclusive prefix sum over the lanes of a wavefront (DPP scan: shifts within the
// rows of 16, then the row totals carried across with row_bcast:15 / row_bcast:31).
__device__ __forceinline__ int ksp_wave_scan_dpp(int v)
{
    v += KSP_DPP(v, 0x111, 0xf);  // row_shr:1
    v += KSP_DPP(v, 0x112, 0xf);  // row_shr:2
    v += KSP_DPP(v, 0x114, 0xf);  // row_shr:4
    v += KSP_DPP(v, 0x118, 0xf);  // row_shr:8  -> inclusive scan inside each row
    v += KSP_DPP(v, 0x142, 0xa);  // rows 1, 3 += total of rows 0, 2
    v += KSP_DPP(v, 0x143, 0xc);  // rows 2, 3 += total of rows 0..1
    return v;
}

__device__ __forceinline__ unsigned ksp_wave_or_dpp(unsigned u)
{
    int v = (int)u;
    v |= KSP_DPP(v, 0x111, 0xf);
    v |= KSP_DPP(v, 0x112, 0xf);
    v |= KSP_DPP(v, 0x114, 0xf);
    v |= KSP_DPP(v, 0x118, 0xf);
    v |= KSP_DPP(v, 0x142, 0xa);
    v |= KSP_DPP(v, 0x143, 0xc);
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int ksp_wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float ksp_wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ float ksp_wave_min(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ double ksp_wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ double ksp_wave_min(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
#endif
