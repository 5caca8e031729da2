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
// change sides: a leaving padding is taken from the side that is ahead, an entering
// one goes to the side that is behind.
//
// That invariant leaves ONE bit of state per lane: `odd` = "one more +inf than -inf"
// (equivalently: the number of valid samples is even). A lane-boolean lives in a
// scalar register pair as a wavefront mask, so updating it is scalar-unit work and
// the vector unit only runs the compares/selects of the sorted array itself. The
// caller supplies, next to each leaving/entering sample, whether it takes part.
#pragma once
#include "ksp_common.h"

template <int WIDTH>
struct SortedWindow {
    static constexpr int H = WIDTH / 2;
    float s[WIDTH];  // sorted ascending
    bool odd;        // #(+inf paddings) == #(-inf paddings) + 1
    // deviations computed so far that are not zero but round to a float32 zero (+-2^-150: the
    // mean of two subnormal samples against a third); the host counts them among the non-zero
    // deviations of the MAD (rfi/host.py:161), so mad_noise() is told how many there are
    int tiny = 0;
    float pinf, ninf;  // +-infinity, opaque to the optimiser (see reset)

    __device__ __forceinline__ void reset()
    {
        // Kept opaque so that med3(a, b, +-inf) stays ONE v_med3_f32: folded to
        // fminf/fmaxf it would drag a NaN-canonicalising v_max_f32 x, x, x along.
        pinf = __builtin_inff();
        ninf = -__builtin_inff();
        asm volatile("" : "+v"(pinf), "+v"(ninf));
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = i < H ? ninf : pinf;
        odd = true;
    }

    // `in_ok`: the entering sample takes part (otherwise its value is ignored and a
    // padding enters on the side that keeps the balance)
    __device__ __forceinline__ void insert(const float (&L)[WIDTH - 1], float in, bool in_ok)
    {
        const float vi = in_ok ? in : (odd ? ninf : pinf);
        odd = (odd == in_ok);  // toggles when a padding entered
        s[0] = __builtin_amdgcn_fmed3f(L[0], vi, ninf);  // min
#pragma unroll
        for (int i = 1; i < WIDTH - 1; i++) s[i] = __builtin_amdgcn_fmed3f(L[i - 1], vi, L[i]);
        s[WIDTH - 1] = __builtin_amdgcn_fmed3f(L[WIDTH - 2], vi, pinf);  // max
    }

    // Replace a leaving PADDING (statically known, e.g. the reset state during warm-up)
    // by `in`: +inf paddings sit at the top, -inf paddings at the bottom, so removal
    // is a shift selected by `odd` -- no compares.
    __device__ __forceinline__ void step_pad_out(float in, bool in_ok)
    {
        float L[WIDTH - 1];
#pragma unroll
        for (int i = 0; i < WIDTH - 1; i++) L[i] = odd ? s[i] : s[i + 1];
        odd = !odd;
        insert(L, in, in_ok);
    }

    // Replace `out` by `in`; *_ok tell whether the sample takes part (a padding's
    // value is ignored).
    __device__ __forceinline__ void step(float out, bool out_ok, float in, bool in_ok)
    {
        // a leaving padding is taken from the +inf side when that side is ahead
        const float vo = out_ok ? out : (odd ? pinf : ninf);
        odd = (odd == out_ok);
        float L[WIDTH - 1];
#pragma unroll
        for (int i = 0; i < WIDTH - 1; i++) L[i] = (s[i] < vo) ? s[i] : s[i + 1];
        insert(L, in, in_ok);
    }

    // float32(centre - median of the valid samples), the median and the subtraction
    // evaluated in float64 as the host path does. With an odd number of valid samples
    // the median is the float32 s[H] and float32(double(x) - double(m)) equals the
    // float32 difference x - m (the double difference is exact or differs from the
    // larger operand by far less than half a float32 ulp), so one v_sub_f32 suffices;
    // the float64 mean is only needed by lanes with an even count, and the whole
    // wavefront skips it when no lane has one.
    __device__ __forceinline__ float deviation(float x)
    {
        float d = x - s[H];
        if (odd) {
            // the empty asm keeps this a real branch (skipped when no lane is odd)
            // instead of eight speculated float64 instructions per step
            asm volatile("");
            // x - (lo + hi) / 2 with one rounding: the sum is exact in float64, halving it
            // is exact, so the fused form equals the host's two steps
            const double dd = __fma_rn(-0.5, (double)s[H > 0 ? H - 1 : 0] + (double)s[H], (double)x);
            d = (float)dd;
            tiny += (d == 0.0f && dd != 0.0);
        }
        return d;
    }
};
