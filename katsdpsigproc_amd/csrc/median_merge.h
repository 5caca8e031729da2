// Sliding-window median by MERGING, for data without masked samples.
//
// Cut the lane's samples into blocks of W (= window width). The window of output j
// is then a SUFFIX of block m = j / W (its last W - t samples, t = j % W) followed by
// a PREFIX of block m + 1 (its first t samples). Keep both sorted:
//   * the suffix lists of a block are built once, by inserting its samples from the
//     back (W (W - 1) / 2 + ... min/med3/max, all lists are kept: 91 registers for
//     W = 13);
//   * the prefix list grows by one insertion per output;
//   * the median (rank H of the union of two sorted lists X, Y) is
//         min over i + j = H + 1 of max(X[i - 1], Y[j - 1])        (X[-1] = Y[-1] = -inf),
//     at most H + 1 terms, fewer near the block ends.
// That is about 21 min/max/med3 per output where the sorted-window step
// (median_window.h) needs 12 compare/select pairs, 13 med3 and the scalar-mask wait
// states between them; nothing here goes through a scalar register.
//
// Samples that do not exist (beyond the band, first and last lane only) are +-inf in
// alternation away from the band edge, which keeps #(+inf) - #(-inf) in {0, 1} for
// every window; with one more +inf the number of valid samples is even and the median
// is the float64 mean of ranks H - 1 and H, as in SortedWindow. Any OTHER masked
// sample (flagged, NaN) would need a rank that varies per lane: callers use the
// sorted-window path for such data.
#pragma once
#include <utility>

#include "ksp_common.h"

template <int... Is, class F>
__device__ __forceinline__ void ksp_static_for_impl(std::integer_sequence<int, Is...>, F &&f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
// f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <int N, class F>
__device__ __forceinline__ void ksp_static_for(F &&f)
{
    ksp_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// single-instruction min/max (inputs are never NaN here, so no canonicalisation)
__device__ __forceinline__ float ksp_vmin3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int R, int W>
struct MergeMedian {
    static constexpr int H = W / 2;
    static_assert(W % 2 == 1 && H <= R, "odd window, reaching at most one lane away");
    // suffix list t (the block's samples t .. W-1, sorted) lives at S[off(t)], length W - t
    static constexpr int off(int t) { return t * W - t * (t - 1) / 2; }
    static constexpr int S_SIZE = W * (W + 1) / 2;

    float pinf, ninf;  // opaque +-inf (med3 with them is min / max in one instruction)
    int tiny = 0;      // non-zero deviations that round to a float32 zero (see SortedWindow::tiny)

    __device__ __forceinline__ float vmin(float a, float b) const
    {
        return __builtin_amdgcn_fmed3f(a, b, ninf);
    }
    __device__ __forceinline__ float vmax(float a, float b) const
    {
        return __builtin_amdgcn_fmed3f(a, b, pinf);
    }

    // dst[0 .. N] = src[0 .. N) with v inserted (both sorted); N >= 1
    template <int N>
    __device__ __forceinline__ void insert(const float *src, float v, float *dst) const
    {
        float out[N + 1];
        out[0] = vmin(src[0], v);
#pragma unroll
        for (int i = 1; i < N; i++) out[i] = __builtin_amdgcn_fmed3f(src[i - 1], v, src[i]);
        out[N] = vmax(src[N - 1], v);
#pragma unroll
        for (int i = 0; i <= N; i++) dst[i] = out[i];
    }

    // element of 0-based rank K in the union of sorted X (A elements) and Y (B elements)
    template <int A, int B, int K>
    __device__ __forceinline__ float rank(const float *X, const float *Y) const
    {
        constexpr int LO = (K + 1 - B) > 0 ? (K + 1 - B) : 0;
        constexpr int HI = (K + 1) < A ? (K + 1) : A;
        constexpr int N = HI - LO + 1;
        static_assert(N >= 1, "rank outside the union");
        float term[N];
#pragma unroll
        for (int n = 0; n < N; n++) {
            const int i = LO + n, j = K + 1 - i;
            term[n] = (i == 0) ? Y[j - 1] : (j == 0) ? X[i - 1] : vmax(X[i - 1], Y[j - 1]);
        }
        float r = term[0];
        int n = 1;
#pragma unroll
        for (; n + 1 < N; n += 2) r = ksp_vmin3(r, term[n], term[n + 1]);
        if (n < N) r = vmin(r, term[n]);
        return r;
    }

    // `run` -> the lane's first sample in the LDS row (neighbouring lanes' runs sit
    // RUN_GAP words beyond the run's ends); dev[j], j < R, and their maximum are
    // produced for the lane's R output channels.
    template <int RUN_GAP>
    __device__ __forceinline__ void run_lane(const float *run, int lane, float (&dev)[R],
                                             float &dmax)
    {
        run_src([&](int i) { return run[i]; }, [&](int i) { return run[i - RUN_GAP]; },
                [&](int i) { return run[i + RUN_GAP]; }, lane == 0, lane == 63, dev, dmax);
    }

    // General form: in(i) is the lane's own sample i (0 <= i < R), left(i) (i < 0) and
    // right(i) (i >= R) those of the neighbouring runs -- asked for only where such a
    // run exists (`first` / `last`: this run is the first / last of the band); beyond the
    // band the +-inf stand-ins are supplied here.
    template <class In, class Left, class Right>
    __device__ __forceinline__ void run_src(In &&in, Left &&left, Right &&right, bool first,
                                            bool last, float (&dev)[R], float &dmax)
    {
        pinf = __builtin_inff();
        ninf = -__builtin_inff();
        asm volatile("" : "+v"(pinf), "+v"(ninf));
        // sample i of the lane, -H <= i < R + H
        auto xs = [&](int i) -> float {
            if (i >= 0 && i < R) return in(i);
            if (i < 0) return !first ? left(i) : (((-i) & 1) ? pinf : ninf);
            return !last ? right(i) : (((i - R) & 1) ? ninf : pinf);
        };
        dmax = ninf;
        constexpr int STAGES = (R + W - 1) / W;
        float cur[W];  // block m, unsorted
#pragma unroll
        for (int k = 0; k < W; k++) cur[k] = xs(-H + k);
        ksp_static_for<STAGES>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            float S[S_SIZE];
            S[off(W - 1)] = cur[W - 1];
            ksp_static_for<W - 1>([&](auto u_) {
                constexpr int t = W - 2 - decltype(u_)::value;  // W-2 .. 0
                insert<W - 1 - t>(&S[off(t + 1)], cur[t], &S[off(t)]);
            });
            float P[W];    // sorted prefix of block m + 1
            float nxt[W];  // block m + 1, unsorted, as far as read
            ksp_static_for<W>([&](auto t_) {
                constexpr int t = decltype(t_)::value;
                constexpr int j = m * W + t;  // output channel (relative)
                if constexpr (j < R) {
                    if constexpr (t >= 1) {
                        nxt[t - 1] = xs(-H + (m + 1) * W + t - 1);
                        if constexpr (t == 1)
                            P[0] = nxt[0];
                        else
                            insert<t - 1>(P, nxt[t - 1], P);
                    }
                    const float med = rank<W - t, t, H>(&S[off(t)], P);
                    const float xc = (t + H < W) ? cur[t + H] : nxt[t + H - W];
                    float d = xc - med;
                    // windows that reach beyond the band by an odd number of samples
                    // (first / last lane only) hold an even number of valid ones
                    constexpr bool left_odd = j < H && ((H - j) & 1);
                    constexpr bool right_odd = j + H >= R && ((j + H - R + 1) & 1);
                    if constexpr (left_odd || right_odd) {
                        const float lo = rank<W - t, t, H - 1>(&S[off(t)], P);
                        if (left_odd ? first : last) {
                            const double dd = (double)xc - ((double)lo + (double)med) * 0.5;
                            d = (float)dd;
                            tiny += (d == 0.0f && dd != 0.0);
                        }
                    }
                    dmax = vmax(dmax, d);
                    dev[j] = d;
                }
            });
            if constexpr (m + 1 < STAGES) {
                // the rest of block m + 1 (only what later windows can reach)
                ksp_static_for<W>([&](auto k_) {
                    constexpr int k = decltype(k_)::value;
                    constexpr int i = -H + (m + 1) * W + k;
                    constexpr bool have = (m * W + k + 1 < R) && (k + 1 < W);  // read in the loop above
                    if constexpr (!have) nxt[k] = (i < R + H) ? xs(i < R + H ? i : 0) : pinf;
                });
#pragma unroll
                for (int k = 0; k < W; k++) cur[k] = nxt[k];
            }
        });
    }
};
