// Rank / selection primitives for one 256-thread workgroup (4 wavefronts) whose
// threads each hold VT positive floats in registers. They play the role of the
// reference's rank.mako + wg_reduce.mako macro library (reference:
// rank.mako:31-267, wg_reduce.mako:97-201), re-thought for 64-wide wavefronts:
// per-lane counts are summed with cross-lane shuffles inside a wave and only the
// four wave totals cross LDS, so a rank query costs one barrier pair instead of
// a log2(workgroup) LDS rake.
//
// Values are ordered by their IEEE-754 bit patterns, which is the natural order
// for non-negative floats (same assumption as rank.mako:188-189). Slots beyond
// the data are padded with NaN, whose pattern (0x7fc00000) sorts above every
// finite value and +inf, so "count of keys < pivot" never counts padding for any
// pivot that is a real value.
#pragma once
#include "ksp_common.h"

#define KSP_RANK_THREADS 256
#define KSP_RANK_WAVES (KSP_RANK_THREADS / KSP_WAVE)

struct RankScratch {
    int isum[3][KSP_RANK_WAVES];
    float fred[KSP_RANK_WAVES];
};

// Sum three per-thread counts over the workgroup; every thread gets the totals.
__device__ __forceinline__ void block_sum3(int &a, int &b, int &c, RankScratch *s)
{
    a = ksp_wave_sum(a);
    b = ksp_wave_sum(b);
    c = ksp_wave_sum(c);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // previous users of the scratch are done
    if ((threadIdx.x & 63) == 0) {
        s->isum[0][wave] = a;
        s->isum[1][wave] = b;
        s->isum[2][wave] = c;
    }
    __syncthreads();
    a = b = c = 0;
#pragma unroll
    for (int w = 0; w < KSP_RANK_WAVES; w++) {
        a += s->isum[0][w];
        b += s->isum[1][w];
        c += s->isum[2][w];
    }
}

__device__ __forceinline__ int block_sum(int a, RankScratch *s)
{
    a = ksp_wave_sum(a);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s->isum[0][wave] = a;
    __syncthreads();
    a = 0;
#pragma unroll
    for (int w = 0; w < KSP_RANK_WAVES; w++) a += s->isum[0][w];
    return a;
}

template <bool IS_MAX>
__device__ __forceinline__ float block_minmax(float v, RankScratch *s)
{
    v = IS_MAX ? ksp_wave_max(v) : ksp_wave_min(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s->fred[wave] = v;
    __syncthreads();
    v = s->fred[0];
#pragma unroll
    for (int w = 1; w < KSP_RANK_WAVES; w++) v = IS_MAX ? fmaxf(v, s->fred[w]) : fminf(v, s->fred[w]);
    return v;
}

// Smallest / largest non-NaN value (fminf/fmaxf ignore NaN operands), NaN if none
// (reference rank.mako:57-84).
template <int VT>
__device__ __forceinline__ float block_fmin(const float (&v)[VT], RankScratch *s)
{
    float m = v[0];
#pragma unroll
    for (int i = 1; i < VT; i++) m = fminf(m, v[i]);
    return block_minmax<false>(m, s);
}

template <int VT>
__device__ __forceinline__ float block_fmax(const float (&v)[VT], RankScratch *s)
{
    float m = v[0];
#pragma unroll
    for (int i = 1; i < VT; i++) m = fmaxf(m, v[i]);
    return block_minmax<true>(m, s);
}

// Values of rank r0, r1, r2 (0-based, "number of keys strictly below") found
// together by a bit-wise binary search on the float bit pattern (reference
// rank.mako:197-207 finds one rank per 31-pass search; sharing the passes
// between three ranks costs three compares per key instead of three searches).
template <int VT>
__device__ __forceinline__ void block_select3(const float (&v)[VT], int r0, int r1, int r2,
                                              float &o0, float &o1, float &o2, RankScratch *s)
{
    unsigned cur0 = 0, cur1 = 0, cur2 = 0;
    for (int bit = 30; bit >= 0; bit--) {
        const unsigned t0 = cur0 | (1u << bit), t1 = cur1 | (1u << bit), t2 = cur2 | (1u << bit);
        int c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
        for (int i = 0; i < VT; i++) {
            const unsigned k = __float_as_uint(v[i]);
            c0 += k < t0;
            c1 += k < t1;
            c2 += k < t2;
        }
        block_sum3(c0, c1, c2, s);
        if (c0 <= r0) cur0 = t0;
        if (c1 <= r1) cur1 = t1;
        if (c2 <= r2) cur2 = t2;
    }
    o0 = __uint_as_float(cur0);
    o1 = __uint_as_float(cur1);
    o2 = __uint_as_float(cur2);
}

// Single-rank search; if halfway, the mean of ranks `rank` and `rank - 1`
// computed as (a + b) * 0.5f in float32 (reference rank.mako:209-218, and what
// numpy.median does for a float32 array).
template <int VT>
__device__ __forceinline__ float block_select(const float (&v)[VT], int rank, bool halfway,
                                              RankScratch *s)
{
    unsigned cur = 0;
    for (int bit = 30; bit >= 0; bit--) {
        const unsigned t = cur | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < VT; i++) c += __float_as_uint(v[i]) < t;
        c = block_sum(c, s);
        if (c <= rank) cur = t;
    }
    float result = __uint_as_float(cur);
    if (halfway) {  // workgroup-uniform condition
        int c = 0;
        float below = 0.0f;
#pragma unroll
        for (int i = 0; i < VT; i++) {
            const bool lt = __float_as_uint(v[i]) < cur;
            c += lt;
            below = lt ? fmaxf(below, v[i]) : below;
        }
        c = block_sum(c, s);
        below = block_minmax<true>(below, s);
        // if fewer than `rank` keys are strictly below, rank-1 is a duplicate of result
        const float prev = (c == rank) ? below : result;
        result = __fmul_rn(__fadd_rn(result, prev), 0.5f);
    }
    return result;
}

// Median of the non-zero values of a row of `n` non-negative floats held VT per thread
// (NaN beyond the data), float32 mean of the middle two for an even count, NaN when
// every value is zero (reference rank.mako:253-267: zeros are counted and the target
// rank shifted past them instead of removing them).
template <int VT>
__device__ __forceinline__ float block_median_non_zero(const float (&v)[VT], int n, RankScratch *s)
{
    int zeros = 0;
#pragma unroll
    for (int i = 0; i < VT; i++) zeros += (v[i] == 0.0f);
    zeros = block_sum(zeros, s);
    // zeros sort first, so the median of the non-zero values has rank (n + zeros) / 2
    const int rank2 = n + zeros;
    float med = block_select(v, rank2 / 2, !(rank2 & 1), s);
    if (zeros == n) med = __builtin_nanf("");  // numpy: median of nothing
    return med;
}
