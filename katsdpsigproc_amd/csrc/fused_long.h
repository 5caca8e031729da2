// Fused flagger for bands of more than 4096 channels (the reference script's 8192- and
// 10240-channel presets, scripts/rfiflagtest.py:190-195).
//
// Same decomposition as flagger_fused_kernel.h -- a strip of adjacent baselines in LDS,
// one wavefront per baseline, everything after the load wave-local -- but a lane now
// owns NR runs of 64 channels instead of one: run g * 64 + lane for g = 0 .. NR - 1
// ("group" g covers channels [4096 g, 4096 g + 4096)), so that inside a group the
// neighbours of a lane's run are still the neighbouring lanes'. The LDS image of a
// baseline is NR times larger, which leaves room for one workgroup per CU (4 baselines
// up to 9088 channels, 3 up to 12288); with one wavefront per SIMD the register file
// holds the NR * 64 float32 deviations per lane. The building blocks (sorted-window and
// merging median, exact recomputation, candidate ranking) are those of fused_common.h;
// the MAD's bit-plane search and SumThreshold are restated here over NR groups.
#pragma once
#include "fused_common.h"

#define LONG_RUN 68  // floats per 64-channel run in LDS (4 words of padding, as FusedLayout)

__device__ __forceinline__ int long_index(int c) { return (c >> 6) * LONG_RUN + (c & 63); }

// ---------------------------------------------------------------------------------
// Load: one baseline per lane (S lanes per row, 64 rows per pass), vis -> amplitude ->
// LDS rows [baseline][channel]; channels C .. 64 runs - 1 are filled with NaN. Handles
// every input form (complex / amplitude, any input-flags mode, ragged last strip).
// Returns whether this lane produced an amplitude that takes no part.
template <int S>
__device__ __forceinline__ bool load_strip_long(const FusedParams &p, float *lds, int row_floats,
                                                int runs, int b0, int tid)
{
    constexpr int LB = 4, NB = 3;
    constexpr int RSTEP = 64;  // = threads / S
    constexpr int BATCH = RSTEP * LB;
    const int C = p.channels;
    const int q = tid % S;
    const int r0 = tid / S;
    const int bl = min(b0 + q, p.baselines - 1);  // ragged strip: duplicate the last baseline
    float *myrow = lds + q * row_floats;
    unsigned umax = 0;
    auto request = [&](float2 (&raw)[LB], unsigned (&fl)[LB], int rbase) {
#pragma unroll
        for (int u = 0; u < LB; u++) {
            const int row = min(rbase + r0 + u * RSTEP, C - 1);
            if (p.is_amplitude)
                raw[u] = make_float2(((const float *)p.vis)[(size_t)row * p.vis_stride + bl], 0.0f);
            else
                raw[u] = ((const float2 *)p.vis)[(size_t)row * p.vis_stride + bl];
            fl[u] = 0;
            if (p.flags_mode == KSP_FLAGS_CHANNEL)
                fl[u] = p.in_flags[row];
            else if (p.flags_mode == KSP_FLAGS_FULL)
                fl[u] = p.in_flags[(size_t)row * p.in_flags_stride + bl];
        }
    };
    auto finish = [&](const float2 (&raw)[LB], const unsigned (&fl)[LB], int rbase) {
#pragma unroll
        for (int u = 0; u < LB; u++) {
            const int row = rbase + r0 + u * RSTEP;
            float a = p.is_amplitude ? raw[u].x : ksp_abs_c64(raw[u].x, raw[u].y);
            if (fl[u]) a = __builtin_nanf("");
            if (row < C) {
                umax = max(umax, __float_as_uint(a));
                myrow[long_index(row)] = a;
            }
        }
    };
    float2 buf[NB][LB];
    unsigned fl[NB][LB];
#pragma unroll
    for (int k = 0; k < NB - 1; k++)
        if (k * BATCH < C) request(buf[k], fl[k], k * BATCH);
    for (int rb = 0; rb < C; rb += NB * BATCH) {
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int cur = rb + k * BATCH;
            if (cur < C) {
                const int ahead = cur + (NB - 1) * BATCH;
                if (ahead < C) request(buf[(k + NB - 1) % NB], fl[(k + NB - 1) % NB], ahead);
                finish(buf[k], fl[k], cur);
            }
        }
    }
    for (int row = C + r0; row < 64 * runs; row += RSTEP) myrow[long_index(row)] = __builtin_nanf("");
    // (negative amplitudes of an amplitude input sort above 0x80000000: not "masked")
    return (umax & 0x7fffffffu) > 0x7f800000u;
}

// Fast form for complex input and a whole strip of 4 baselines: two lanes per row, each
// loading a pair of baselines (16 bytes), 128 rows per pass -- the loader of the
// 4096-channel kernel (load_strip_fast) with this kernel's row layout.
template <int MODE>
__device__ __forceinline__ bool load_strip_long_pairs(const FusedParams &p, float *lds,
                                                      int row_floats, int runs, int b0, int tid)
{
    constexpr int LB = 4, NB = 3;
    constexpr int RSTEP = 128;
    constexpr int BATCH = RSTEP * LB;
    const int C = p.channels;
    const int q = tid & 1;
    const int r0 = tid >> 1;
    const int bl = b0 + 2 * q;
    const float2 *vis = (const float2 *)p.vis + bl;
    const size_t stride = (size_t)p.vis_stride;
    float *row_a = lds + (2 * q) * row_floats, *row_b = lds + (2 * q + 1) * row_floats;
    unsigned umax = 0, flag_or = 0;
    auto request = [&](float4 (&raw)[LB], unsigned (&fl)[LB], int rbase) {
#pragma unroll
        for (int u = 0; u < LB; u++) {
            const int row = min(rbase + r0 + u * RSTEP, C - 1);
            raw[u] = *(const float4 *)(vis + (size_t)row * stride);
            fl[u] = 0;
            if (MODE == KSP_FLAGS_CHANNEL)
                fl[u] = p.in_flags[row];
            else if (MODE == KSP_FLAGS_FULL)
                fl[u] = *(const unsigned short *)(p.in_flags + (size_t)row * p.in_flags_stride + bl);
        }
    };
    auto finish = [&](const float4 (&raw)[LB], const unsigned (&fl)[LB], int rbase) {
        float amp[LB][2];
        ksp_abs_c64_batch<LB, MODE == KSP_FLAGS_NONE>(raw, amp, umax);
#pragma unroll
        for (int u = 0; u < LB; u++) {
            const int row = rbase + r0 + u * RSTEP;
            float a0 = amp[u][0], a1 = amp[u][1];
            if (MODE == KSP_FLAGS_CHANNEL) {
                if (fl[u]) a0 = a1 = __builtin_nanf("");
            } else if (MODE == KSP_FLAGS_FULL) {
                if (fl[u] & 0xffu) a0 = __builtin_nanf("");
                if (fl[u] >> 8) a1 = __builtin_nanf("");
            }
            // (only the general |z| can yield a NaN by itself: watched there; a flagged
            // sample shows in its flag)
            if (MODE != KSP_FLAGS_NONE) flag_or |= fl[u];
            if (row < C) {
                const int idx = long_index(row);
                row_a[idx] = a0;
                row_b[idx] = a1;
            }
        }
    };
    float4 buf[NB][LB];
    unsigned fl[NB][LB];
#pragma unroll
    for (int k = 0; k < NB - 1; k++)
        if (k * BATCH < C) request(buf[k], fl[k], k * BATCH);
    for (int rb = 0; rb < C; rb += NB * BATCH) {
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int cur = rb + k * BATCH;
            if (cur < C) {
                const int ahead = cur + (NB - 1) * BATCH;
                if (ahead < C) request(buf[(k + NB - 1) % NB], fl[(k + NB - 1) % NB], ahead);
                finish(buf[k], fl[k], cur);
            }
        }
    }
    for (int row = C + r0; row < 64 * runs; row += RSTEP) {
        row_a[long_index(row)] = __builtin_nanf("");
        row_b[long_index(row)] = __builtin_nanf("");
    }
    return umax > 0x7f800000u || flag_or != 0;
}

// ---------------------------------------------------------------------------------
// MAD over NR groups of 64 deviations per lane: 1.4826 x median of the non-zero |dev| of
// the whole baseline. The search is mad_noise()'s bit-plane search with every count
// summed over the groups; sample (g, j) of a lane is channel ((g * 64 + lane) << 6) + j.
template <int NR, int WIDTH, int LIST_CAP, class Fetch>
__device__ __forceinline__ double mad_noise_long(const float (&dev)[NR][64], int lane,
                                                 double *list, Fetch &&fetch, int tiny = 0)
{
    auto chan0 = [&](int g) { return ((g * 64 + lane) << 6); };
    auto dv = [&](int g, int j) -> float {
        float x = dev[g][j];
        asm("" : "+v"(x));
        return x;
    };
    auto key_of = [&](int g, int j) -> unsigned {
        return min((__float_as_uint(dv(g, j)) & 0x7fffffffu) + 0xffffu, 0x7fffffffu) >> 16;
    };
    // 15-bit keys (rounded up: key 0 = exact zero), transposed into inverted bit planes
    unsigned np[NR][32];
#pragma unroll
    for (int g = 0; g < NR; g++) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            // (signed patterns: see mad_noise in fused_common.h)
            const unsigned a = __float_as_uint(dev[g][2 * i]) + 0xffffu;
            const unsigned b = __float_as_uint(dev[g][2 * i + 1]) + 0xffffu;
            np[g][i] = ~__builtin_amdgcn_perm(b, a, 0x07060302u) & 0x7fff7fffu;
        }
        transpose_bits32(np[g]);
    }
    const int total = NR * 64 * 64;
    int zeros;
    {
        int z = 0;
#pragma unroll
        for (int g = 0; g < NR; g++) {
            unsigned z0 = np[g][0], z1 = np[g][16];
#pragma unroll
            for (int b = 1; b < 15; b++) {
                z0 &= np[g][b];
                z1 &= np[g][16 + b];
            }
            z += __popc(z0) + __popc(z1);
        }
        zeros = ksp_wave_sum_dpp(z);
    }
    // (`tiny` deviations of +-2^-150, float32 zeros that the host counts as non-zero: see
    // mad_noise in fused_common.h)
    const int zeros_keys = zeros;
    if (ksp_any(tiny != 0)) zeros -= ksp_wave_sum_dpp(tiny);
    if (zeros == total) return __builtin_nan("");  // numpy: median of nothing
    const int rank2 = total + zeros;  // zeros sort first (reference rank.mako:261-266)
    const int rank = rank2 / 2;
    if (rank < zeros_keys) return 0x1p-150 * FUSED_MAD_NORMAL;
    unsigned K = 0;
    int below_bin = 0;
    unsigned eq0[NR], eq1[NR];
#pragma unroll
    for (int g = 0; g < NR; g++) eq0[g] = eq1[g] = 0xffffffffu;
    // one bit per step (measured faster than two-bit steps in every bit-plane search here):
    // the keys that match the prefix and have this bit clear are counted over the groups
    auto step1 = [&](int bit) {
        unsigned z0[NR], z1[NR];
        int c = 0;
#pragma unroll
        for (int g = 0; g < NR; g++) {
            z0[g] = eq0[g] & np[g][bit];
            z1[g] = eq1[g] & np[g][16 + bit];
            c += __popc(z0[g]) + __popc(z1[g]);
        }
        c = below_bin + ksp_wave_sum_dpp(c);
        const bool take = c <= rank;
        K |= take ? (1u << bit) : 0u;
        below_bin = take ? c : below_bin;
#pragma unroll
        for (int g = 0; g < NR; g++) {
            eq0[g] = take ? (eq0[g] ^ z0[g]) : z0[g];
            eq1[g] = take ? (eq1[g] ^ z1[g]) : z1[g];
        }
    };
#pragma unroll
    for (int bit = 14; bit >= 0; bit--) step1(bit);
    int mine = 0;
#pragma unroll
    for (int g = 0; g < NR; g++) mine += __popc(eq0[g]) + __popc(eq1[g]);
    int in_bin = ksp_wave_sum_dpp(mine);
    const bool even = !(rank2 & 1);
    auto bin_mask = [&](int g, unsigned key) -> unsigned long long {
        unsigned long long m = 0;
#pragma unroll
        for (int j = 0; j < 64; j++)
            if (key_of(g, j) == key) m |= 1ull << j;
        return m;
    };
    // every sample of every group in `mask[g]`: exact |deviation| handed to f
    auto each = [&](const unsigned long long (&mask)[NR], auto &&f) {
#pragma unroll
        for (int g = 0; g < NR; g++) {
            unsigned long long todo = mask[g];
            while (__any(todo != 0)) {
                const bool has = todo != 0;
                const int j = has ? __ffsll((long long)todo) - 1 : 0;
                todo &= todo - 1;
                const double x = fabs(exact_dev<WIDTH>(chan0(g) + j, fetch));
                if (has) f(x);
            }
        }
    };
    // the largest exact value among the samples whose float32 |dev| is the largest one
    // below pattern `limit_pat` (all samples if limit_pat == 0xffffffff and `keymax` >= 0:
    // then restricted to key bin `keymax`)
    auto largest_below = [&](unsigned limit_pat, int only_key) -> double {
        float b32 = 0.0f;
#pragma unroll
        for (int g = 0; g < NR; g++)
#pragma unroll
            for (int j = 0; j < 64; j++) {
                const float a = fabsf(dv(g, j));
                const bool in = (only_key < 0) ? (__float_as_uint(a) < limit_pat)
                                               : ((int)key_of(g, j) == only_key);
                b32 = in ? fmaxf(b32, a) : b32;
            }
        b32 = ksp_wave_max(b32);
        unsigned long long top[NR];
#pragma unroll
        for (int g = 0; g < NR; g++) {
            top[g] = 0;
#pragma unroll
            for (int j = 0; j < 64; j++) {
                const bool in = (only_key < 0) || ((int)key_of(g, j) == only_key);
                if (in && fabsf(dv(g, j)) == b32) top[g] |= 1ull << j;
            }
        }
        double m = 0.0;
        each(top, [&](double x) { m = fmax(m, x); });
        return ksp_wave_max(m);
    };
    int r = rank - below_bin;  // 0-based rank inside the bin
    double xk, prev;
    bool have_prev;
    if (in_bin <= 64) {
        // usual case: one candidate per lane through the list, exact values ranked by
        // lane broadcast (as mad_noise 3a)
        int *lc = (int *)list;
        const int incl = ksp_wave_scan_dpp(mine);
        const int n = __builtin_amdgcn_readlane(incl, 63);
        int pos = incl - mine;
#pragma unroll
        for (int g = 0; g < NR; g++) {
            unsigned h0 = eq0[g], h1 = eq1[g];
            while (h0 | h1) {  // divergent: as many rounds as the busiest lane has hits
                if (h0) {
                    lc[pos++] = chan0(g) + 2 * (__ffs((int)h0) - 1);
                    h0 &= h0 - 1;
                } else {
                    lc[pos++] = chan0(g) + 2 * (__ffs((int)h1) - 1) + 1;
                    h1 &= h1 - 1;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int c = lc[lane < n ? lane : 0];
        const double x = fabs(exact_dev<WIDTH>(c, fetch));
        rank_lanes64(x, n, r, lane, xk, prev, have_prev);
        __builtin_amdgcn_wave_barrier();
    } else {
        // Crowded bin (quantised data): narrow it to ONE float32 value with an exact search
        // on the full patterns, then select among those samples' float64 values by
        // bisection on their bit patterns (as mad_noise does beyond its list capacity).
        const unsigned first = ((K - 1) << 16) + 1u;  // key K covers first .. K << 16
        unsigned offs = 0;
        int below = below_bin;
        for (int bit = 15; bit >= 0; bit--) {
            const unsigned test = first + (offs | (1u << bit));
            int c = 0;
#pragma unroll
            for (int g = 0; g < NR; g++)
#pragma unroll
                for (int j = 0; j < 64; j++) c += (__float_as_uint(dv(g, j)) & 0x7fffffffu) < test;
            c = ksp_wave_sum(c);
            if (c <= rank) {
                offs |= 1u << bit;
                below = c;
            }
        }
        const unsigned cur = first + offs;
        unsigned long long cand[NR];
        int nc = 0;
#pragma unroll
        for (int g = 0; g < NR; g++) {
            cand[g] = 0;
#pragma unroll
            for (int j = 0; j < 64; j++)
                if ((__float_as_uint(dv(g, j)) & 0x7fffffffu) == cur) cand[g] |= 1ull << j;
            nc += __popcll(cand[g]);
        }
        in_bin = ksp_wave_sum(nc);
        r = rank - below;
        double lo = __builtin_inf(), hi = 0.0;
        each(cand, [&](double x) {
            lo = fmin(lo, x);
            hi = fmax(hi, x);
        });
        lo = ksp_wave_min(lo);
        hi = ksp_wave_max(hi);
        xk = lo;
        prev = lo;
        if (lo != hi) {
            const unsigned long long base = (unsigned long long)__double_as_longlong(lo);
            const unsigned long long span = (unsigned long long)__double_as_longlong(hi) - base;
            unsigned long long o = 0;
            int below_k = 0;
            for (int bit = 63 - __clzll((long long)span); bit >= 0; bit--) {
                const unsigned long long test = base + (o | (1ull << bit));
                int c = 0;
                each(cand, [&](double x) { c += (unsigned long long)__double_as_longlong(x) < test; });
                c = ksp_wave_sum(c);
                if (c <= r) {
                    o |= 1ull << bit;
                    below_k = c;
                }
            }
            xk = __longlong_as_double((long long)(base + o));
            prev = xk;
            if (even && r >= 1 && below_k == r) {
                double m = 0.0;
                each(cand, [&](double x) { m = (x < xk) ? fmax(m, x) : m; });
                prev = ksp_wave_max(m);
            }
        }
        have_prev = r >= 1;
        if (even && !have_prev) {
            prev = largest_below(cur, -1);
            have_prev = true;
        }
    }
    if (even && !have_prev) {
        // r == 0: the lower median is the largest value of the highest non-empty bin below K
        int k2 = -1;
#pragma unroll
        for (int g = 0; g < NR; g++)
#pragma unroll
            for (int j = 0; j < 64; j++) {
                const int kj = (int)key_of(g, j);
                k2 = (kj < (int)K) ? max(k2, kj) : k2;
            }
        k2 = wave_max_int(k2);
        // (no non-zero key below the bin: the value below it is one of the +-2^-150)
        prev = k2 <= 0 ? 0x1p-150 : largest_below(0xffffffffu, k2);
    }
    if (even) xk = (xk + prev) / 2.0;  // float64 mean, as numpy.median
    return xk * FUSED_MAD_NORMAL;
}

// ---------------------------------------------------------------------------------
// Thresholds over NR groups: fl[g] = flag mask of the lane's run in group g (`dev` is
// clobbered). Same
// reasoning as threshold_flags() (fused_common.h); the run after a lane's run in group g
// is lane + 1's in the same group or, for lane 63, lane 0's in group g + 1.
template <int NR, int WIDTH, class Fetch>
__device__ __forceinline__ void threshold_flags_long(const FusedParams &p, float (&dev)[NR][64],
                                                     float dmax, double noise64, int lane, int C,
                                                     Fetch &&fetch, unsigned long long (&fl)[NR])
{
#pragma unroll
    for (int g = 0; g < NR; g++) fl[g] = 0;
    auto chan0 = [&](int g) { return ((g * 64 + lane) << 6); };
    auto inband = [&](int g) -> unsigned long long {
        const int c0 = chan0(g);
        return (c0 + 64 <= C) ? ~0ull : (c0 >= C ? 0ull : ((1ull << (C - c0)) - 1));
    };
    if (p.threshold_kind == KSP_THRESHOLD_SIMPLE) {
        const double thr = p.n_sigma * noise64;  // float64 product (host.py:182)
        if (!__any((double)dmax * (1.0 + 0x1p-23) > thr)) return;
#pragma unroll
        for (int g = 0; g < NR; g++) {
            unsigned long long unsure = 0;
#pragma unroll
            for (int j = 0; j < 64; j++) {
                const double d = (double)dev[g][j];
                const double slack = fabs(d) * 0x1p-23;
                if (d - slack > thr)
                    fl[g] |= 1ull << j;
                else if (d + slack > thr)
                    unsure |= 1ull << j;
            }
            while (__any(unsure != 0)) {
                const bool has = unsure != 0;
                const int j = has ? __ffsll((long long)unsure) - 1 : 0;
                unsure &= unsure - 1;
                const double d = exact_dev<WIDTH>(chan0(g) + j, fetch);
                if (has && d > thr) fl[g] |= 1ull << j;
            }
            fl[g] &= inband(g);
        }
        return;
    }
    const double t1 = p.n_sigma * noise64;  // host.py:252
    constexpr int MAXW = 4;
    float thr[MAXW];
    float thr_min = __builtin_inff();
    bool thr_nan = false;
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        thr[k] = (float)(t1 * p.scales[k < KSP_MAX_WINDOWS ? k : 0]);  // host.py:235
        if (k < p.n_windows) {
            thr_min = fminf(thr_min, thr[k]);
            thr_nan |= (thr[k] != thr[k]);
        }
    }
    const double cand = (double)thr_min * (1.0 - 0x1p-20);
    const bool positive = thr_min > 0.0f;
    if (thr_nan || !__any(!positive || ((double)dmax >= cand))) return;

    // gt0: float32 deviation > thr_0 (decides window 1); ge: deviation >= thr_min ("hot")
    unsigned long long gt0[NR], ge[NR];
    bool weak = false;
#pragma unroll
    for (int g = 0; g < NR; g++) {
        unsigned g_lo = 0, g_hi = 0, e_lo = 0, e_hi = 0;
        const float t0 = thr[0], tm = thr_min;
#pragma unroll
        for (int j = 31; j >= 0; j--) {
            asm("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(g_lo) : "v"(dev[g][j]), "v"(t0) : "vcc");
            asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(e_lo) : "v"(dev[g][j]), "v"(tm) : "vcc");
        }
#pragma unroll
        for (int j = 63; j >= 32; j--) {
            asm("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(g_hi) : "v"(dev[g][j]), "v"(t0) : "vcc");
            asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(e_hi) : "v"(dev[g][j]), "v"(tm) : "vcc");
        }
        gt0[g] = (((unsigned long long)g_hi << 32) | g_lo) & inband(g);
        ge[g] = (((unsigned long long)e_hi << 32) | e_lo) & inband(g);
        weak |= (ge[g] & ~gt0[g]) != 0;
    }
    if (positive && !__any(weak)) {
#pragma unroll
        for (int g = 0; g < NR; g++) fl[g] = gt0[g];
        return;
    }
    // General case (a weak sample somewhere in the baseline): window sums, as
    // threshold_flags(), with the neighbouring run found across groups. The deviations
    // are overwritten in place by the substitution of flagged samples (host.py:237):
    // nothing needs them afterwards, and a copy would not fit the register file.
    float (&d)[NR][64] = dev;
    // value / mask of the run that follows this lane's run of group g
    auto next_f = [&](int g, float (&v)[NR][64], int m) -> float {
        const float same = __shfl_down(v[g][m], 1, 64);
        const float wrap = ksp_bcast(v[g + 1 < NR ? g + 1 : g][m], 0);
        return lane < 63 ? same : wrap;
    };
    auto next_m = [&](int g, const unsigned long long (&v)[NR]) -> unsigned long long {
        const unsigned long long same = __shfl_down(v[g], 1, 64);
        const unsigned long long wrap = g + 1 < NR ? (unsigned long long)__shfl(v[g + 1 < NR ? g + 1 : g], 0, 64) : 0ull;
        return lane < 63 ? same : wrap;
    };
    auto prev_m = [&](int g, const unsigned long long (&v)[NR]) -> unsigned long long {
        const unsigned long long same = __shfl_up(v[g], 1, 64);
        const unsigned long long wrap = g > 0 ? (unsigned long long)__shfl(v[g > 0 ? g - 1 : 0], 63, 64) : 0ull;
        return lane > 0 ? same : wrap;
    };
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        if (k >= p.n_windows) break;
        const int w = 1 << k;
        const float thrf = thr[k];
        const double limit = (double)__fmul_rn(thrf, (float)w);  // host.py:242
#pragma unroll
        for (int g = 0; g < NR; g++)
#pragma unroll
            for (int j = 0; j < 64; j++)
                if ((fl[g] >> j) & 1) d[g][j] = thrf;  // host.py:237
        unsigned long long hits[NR];
#pragma unroll
        for (int g = 0; g < NR; g++) {
            float ext[7];
#pragma unroll
            for (int m = 0; m < 7; m++) ext[m] = next_f(g, d, m);
            const unsigned nfl = (unsigned)(next_m(g, fl) & 0x7fu);
            unsigned long long need = ~0ull;
            if (positive) {
                unsigned long long hot[NR];
#pragma unroll
                for (int gg = 0; gg < NR; gg++) hot[gg] = ge[gg] & ~fl[gg];
                const unsigned long long hot_next = next_m(g, hot);
                unsigned long long reach = hot[g];
#pragma unroll
                for (int m = 1; m < w; m++) reach |= (hot[g] >> m) | (hot_next << (64 - m));
                need = ((unsigned long long)ksp_wave_or_dpp((unsigned)(reach >> 32)) << 32) |
                       ksp_wave_or_dpp((unsigned)reach);
            }
            unsigned long long h = 0, unsure = 0;
            const int c0 = chan0(g);
#pragma unroll
            for (int j = 0; j < 64; j++) {
                if (!((need >> j) & 1)) continue;  // wave-uniform
                double s = 0.0, mag = 0.0;
#pragma unroll
                for (int m = 0; m < w; m++) {
                    const int jj = j + m;
                    const float v = (jj < 64) ? d[g][jj % 64] : ext[(jj >= 64) ? (jj - 64) % 7 : 0];
                    const bool sub = (jj < 64) ? ((fl[g] >> (jj % 64)) & 1)
                                               : ((nfl >> ((jj >= 64) ? (jj - 64) % 7 : 0)) & 1);
                    s += (double)v;
                    mag += sub ? 0.0 : fabs((double)v);
                }
                const bool valid = (c0 + j + w <= C);
                const double slack = mag * 0x1p-23;
                if (valid) {
                    if (s - slack > limit)
                        h |= 1ull << j;
                    else if (s + slack > limit)
                        unsure |= 1ull << j;
                }
            }
            while (__any(unsure != 0)) {
                const bool has = unsure != 0;
                const int j = has ? __ffsll((long long)unsure) - 1 : 0;
                unsure &= unsure - 1;
                double s = 0.0;
                for (int m = 0; m < w; m++) {
                    const int jj = j + m;
                    const bool sub = (jj < 64) ? ((fl[g] >> jj) & 1) : ((nfl >> (jj - 64)) & 1);
                    const double x = exact_dev<WIDTH>(c0 + jj, fetch);
                    s += sub ? (double)thrf : x;
                }
                if (has && s > limit) h |= 1ull << j;
            }
            hits[g] = h;
        }
        // dilation: a hit at j flags j .. j + w - 1 (into the following run if need be)
#pragma unroll
        for (int g = 0; g < NR; g++) {
            unsigned pin = (unsigned)(prev_m(g, hits) >> 57) & 0x7fu;  // hits of the 7 positions before the run
            unsigned long long own = hits[g];
            if (w >= 2) { own |= own << 1; pin |= pin << 1; }
            if (w >= 4) { own |= own << 2; pin |= pin << 2; }
            if (w >= 8) { own |= own << 4; pin |= pin << 4; }
            fl[g] |= (own | (unsigned long long)(pin >> 7)) & inband(g);
        }
    }
}
