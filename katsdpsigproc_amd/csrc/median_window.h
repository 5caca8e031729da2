// Sliding-window median of WIDTH samples held as a SORTED register array.
//
// This replaces the reference kernel's per-thread `samples[WIDTH]`/`rank[WIDTH]`
// shift register (reference: rfi/background_median_filter.mako:79-145), which
// costs O(WIDTH) compares plus a private-array shuffle per sample. Here a step is
//   remove(oldest):  L[i] = s[i] < out ? s[i] : s[i+1]        (WIDTH-1 cmp+select)
//   insert(newest):  s[i] = med3(L[i-1], in, L[i])            (WIDTH v_med3_f32)
// which is branch-free, data-independent, and keeps every array index static
// after unrolling, so nothing spills to scratch.
//
// Samples that must not take part (flagged, NaN, or outside the band -- the host
// path masks them and uses min_periods=1, reference rfi/host.py:138-148) are
// represented by +-infinity PADDING, split so that (#+inf - #-inf) is always 0 or
// 1. The valid samples then sit centred in the sorted array: with an odd number of
// valid samples the median is s[H]; with an even number it is the mean of s[H-1]
// and s[H] (computed in float64 like pandas does). No padding ever needs to
// change sides: see step().
#pragma once
#include "ksp_common.h"

template <int WIDTH>
struct MedianWindow {
    static constexpr int H = WIDTH / 2;
    float s[WIDTH];  // sorted ascending
    int n_neg;       // paddings stored as -inf
    int n_pos;       // paddings stored as +inf

    __device__ __forceinline__ void reset()
    {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = i < H ? -__builtin_inff() : __builtin_inff();
        n_neg = H;
        n_pos = H + 1;
    }

    // Replace `out` (the sample that leaves; NaN if it was padding) by `in` (NaN if
    // the entering sample is invalid).
    __device__ __forceinline__ void step(float out, float in)
    {
        const float inf = __builtin_inff();
        const bool out_valid = out == out;
        const bool in_valid = in == in;
        // A leaving padding is taken from the +inf side when that side is ahead,
        // otherwise from the -inf side; either keeps 0 <= n_pos - n_neg <= 1.
        const bool take_pos = n_pos > n_neg;
        const float vo = out_valid ? out : (take_pos ? inf : -inf);
        n_pos -= (!out_valid && take_pos);
        n_neg -= (!out_valid && !take_pos);
        const bool give_pos = n_pos == n_neg;
        const float vi = in_valid ? in : (give_pos ? inf : -inf);
        n_pos += (!in_valid && give_pos);
        n_neg += (!in_valid && !give_pos);

        float L[WIDTH - 1];
#pragma unroll
        for (int i = 0; i < WIDTH - 1; i++) L[i] = (s[i] < vo) ? s[i] : s[i + 1];
        s[0] = fminf(L[0], vi);
#pragma unroll
        for (int i = 1; i < WIDTH - 1; i++) s[i] = __builtin_amdgcn_fmed3f(L[i - 1], vi, L[i]);
        s[WIDTH - 1] = fmaxf(L[WIDTH - 2], vi);
    }

    __device__ __forceinline__ int n_valid() const { return WIDTH - n_neg - n_pos; }

    // Median of the valid samples, as float64 (requires n_valid() >= 1).
    __device__ __forceinline__ double median() const
    {
        const double hi = (double)s[H];
        if (WIDTH == 1) return hi;
        const double lo = (double)s[H > 0 ? H - 1 : 0];
        return (n_pos == n_neg) ? hi : (lo + hi) * 0.5;
    }
};
