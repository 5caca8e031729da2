// Building blocks of the fused flagger kernel (flagger_fused.hip): strip layout in
// LDS, cooperative strip load, and the wave-local median / MAD / threshold phases.
// See DESIGN.md section 4.1 for the reasoning.
#pragma once
#include <stdlib.h>

#include "median_window.h"

#define FUSED_THREADS 512
#define FUSED_STRIP 8
#define FUSED_MAD_NORMAL 1.4826

struct FusedParams {
    const void *vis;
    const uint8_t *in_flags;
    uint8_t *flags;
    float *deviations;
    float *noise;
    int channels, baselines;
    int vis_stride, in_flags_stride, flags_stride, dev_stride;
    int is_amplitude, flags_mode, threshold_kind, n_windows, flag_value;
    int n_strips;
    int debug_stop;  // diagnostic only (env KSP_FUSED_DEBUG_STOP): 0 = run everything
    double n_sigma;
    double scales[KSP_MAX_WINDOWS];
};

// LDS image of one strip: 8 rows (one per baseline) of float32. Lane l of the owning
// wavefront works on channels [l*R, (l+1)*R): its run starts at word l*RUN. RUN is ODD
// so that the 64 lanes of a wavefront, all reading the same offset of their own run, hit
// 32 different banks; the 9 pad words per run hold the first 6 deviations (see
// dev_slot). Rows are 8 words apart modulo 32 so that the baselines written by one
// lane group during the load fall in different banks. After the rows: one small
// candidate list per wavefront for the MAD.
template <int R>
struct FusedLayout {
    static constexpr int PAD = 9;
    static constexpr int RUN = R + PAD;
    static constexpr int ROW = ((64 * RUN + 31) / 32) * 32 + 8;
    static constexpr int LDS_FLOATS = FUSED_STRIP * ROW;
    static constexpr int LIST_DOUBLES = 128;                             // candidates
    static constexpr int LIST_STRIDE = LIST_DOUBLES + LIST_DOUBLES / 2;  // + as many ints
    static constexpr size_t LDS_BYTES =
        sizeof(float) * LDS_FLOATS + sizeof(double) * LIST_STRIDE * FUSED_STRIP;
    // word of the AMPLITUDE of channel c within a row
    __device__ static __forceinline__ int amp_slot(int c) { return (c / R) * RUN + (c % R); }
    // word of the DEVIATION of channel c within a row: deviations overwrite amplitudes
    // in place once these are dead, except for the first 6 channels of a run, which the
    // lane below still reads late (as the tail of its own windows) and which therefore
    // go to the pad words.
    __device__ static __forceinline__ int dev_slot(int c)
    {
        const int j = c % R;
        return (c / R) * RUN + (j < 6 ? R + j : j);
    }
};
static_assert(FusedLayout<64>::LDS_BYTES <= 160 * 1024, "LDS image must fit one CU");

// blockIdx -> strip, XCD-aware: workgroups b, b+8, ... share an XCD, so the 8 strips
// that make up one 512-byte input line are given to workgroups of one XCD, where their
// half-lines meet in L2. Speed only -- any bijection is correct.
__device__ __forceinline__ int strip_of(int id, int n_strips)
{
    const int full = (n_strips / 64) * 64;
    if (id >= full) return id;
    const int xcd = id & 7, i = id >> 3;
    return ((i >> 3) * 8 + xcd) * 8 + (i & 7);
}

__device__ __forceinline__ float amp_with_flags(const FusedParams &p, float re, float im,
                                                int row, int bl)
{
    float a = ksp_abs_c64(re, im);
    if (p.flags_mode == KSP_FLAGS_CHANNEL) {
        if (p.in_flags[row]) a = __builtin_nanf("");
    } else if (p.flags_mode == KSP_FLAGS_FULL) {
        if (p.in_flags[(size_t)row * p.in_flags_stride + bl]) a = __builtin_nanf("");
    }
    return a;
}

// ---------------------------------------------------------------------------------
// Strip load, split in two halves so that the memory requests of strip k+1 can be in
// flight while strip k is being processed: `request` issues every 16-byte load of a
// lane (a pair of baselines = a quarter of a 64-byte row segment, every 128th row);
// `finish` turns them into numpy's |z| and stores the float32 amplitudes in LDS.
// NROWS = ceil(C / 128) requests per lane stay in registers in between.
template <int R>
struct StripLoader {
    using LY = FusedLayout<R>;
    static constexpr int RSTEP = FUSED_THREADS / 4;                   // rows covered per pass
    static constexpr int NROWS = (64 * R + RSTEP - 1) / RSTEP;        // rows per lane
#ifndef KSP_PREFETCH_ROWS
#define KSP_PREFETCH_ROWS 0
#endif
    // rows requested ahead of time and held in registers while the previous strip is
    // processed; the rest is loaded when the strip is finished (registers are finite:
    // a spilled request would have to be waited for at once, which defeats the purpose)
    static constexpr int NPRE = NROWS < KSP_PREFETCH_ROWS ? NROWS : KSP_PREFETCH_ROWS;
    float4 raw[NPRE];

    // fast path: complex64 input and both baselines of the pair exist
    __device__ __forceinline__ static bool plain(const FusedParams &p, int b0, int tid)
    {
        return !p.is_amplitude && (b0 + 2 * (tid & 3) + 1 < p.baselines);
    }

    __device__ __forceinline__ void request(const FusedParams &p, int b0, int tid)
    {
        const int bl = b0 + 2 * (tid & 3);
        const int r0 = tid >> 2;
        const bool ok = plain(p, b0, tid);
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            const int row = r0 + u * RSTEP;
            raw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && row < p.channels)
                raw[u] = *(const float4 *)((const float2 *)p.vis + (size_t)row * p.vis_stride + bl);
        }
    }

    // Input flags are NOT applied here (the median phase masks the samples as it reads
    // them), which keeps this fully unrolled block small.
    __device__ __forceinline__ void finish(const FusedParams &p, float *lds, int b0, int tid)
    {
        const int q = tid & 3;
        const int bl = b0 + 2 * q;
        const int r0 = tid >> 2;
        const bool ok0 = bl < p.baselines, ok1 = bl + 1 < p.baselines;
        const bool ok = plain(p, b0, tid);
        if (ok) {
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                const int row = r0 + u * RSTEP;
                if (row >= p.channels) break;
                const int slot = LY::amp_slot(row);
                lds[(2 * q) * LY::ROW + slot] = ksp_abs_c64(raw[u].x, raw[u].y);
                lds[(2 * q + 1) * LY::ROW + slot] = ksp_abs_c64(raw[u].z, raw[u].w);
            }
            // the rows that were not requested ahead: batches of 8 requests (rolled loop)
            constexpr int LB = 8;
#pragma unroll 1
            for (int ub = NPRE; ub < NROWS; ub += LB) {
                float4 late[LB];
#pragma unroll
                for (int u = 0; u < LB; u++) {
                    const int row = r0 + (ub + u) * RSTEP;
                    late[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < p.channels)
                        late[u] = *(const float4 *)((const float2 *)p.vis +
                                                    (size_t)row * p.vis_stride + bl);
                }
#pragma unroll
                for (int u = 0; u < LB; u++) {
                    const int row = r0 + (ub + u) * RSTEP;
                    if (row < p.channels) {
                        const int slot = LY::amp_slot(row);
                        lds[(2 * q) * LY::ROW + slot] = ksp_abs_c64(late[u].x, late[u].y);
                        lds[(2 * q + 1) * LY::ROW + slot] = ksp_abs_c64(late[u].z, late[u].w);
                    }
                }
            }
            return;
        }
        // amplitude input, or the ragged last strip: plain loads, no pipelining
#pragma unroll 1
        for (int row = r0; row < p.channels; row += RSTEP) {
            float a0 = __builtin_nanf(""), a1 = __builtin_nanf("");
            if (p.is_amplitude) {
                const float *src = (const float *)p.vis + (size_t)row * p.vis_stride + bl;
                if (ok0) a0 = src[0];
                if (ok1) a1 = src[1];
            } else if (ok0) {
                const float2 v = ((const float2 *)p.vis)[(size_t)row * p.vis_stride + bl];
                a0 = ksp_abs_c64(v.x, v.y);
            }
            const int slot = LY::amp_slot(row);
            lds[(2 * q) * LY::ROW + slot] = a0;
            lds[(2 * q + 1) * LY::ROW + slot] = a1;
        }
    }
};

// ---------------------------------------------------------------------------------
// Median phase. Lane l slides the sorted window (median_window.h) over its run of R
// channels of the wavefront's baseline. The loop is ROLLED in blocks of WIDTH steps
// (the ring of the last WIDTH samples then keeps static register indices), which keeps
// the code small enough for the instruction cache; amplitudes come from LDS by dynamic
// address, and each deviation -- computed in float64 as the host does, then rounded to
// float32 -- is stored back to LDS over an amplitude that is no longer needed
// (FusedLayout::dev_slot). Rounding is monotone, so order statistics can be located on
// the float32 values; the few samples whose exact value decides a result are
// recomputed in float64 on demand (exact_dev). Returns the exact largest deviation of
// the lane's run.
template <int R, int WIDTH>
__device__ __forceinline__ double median_phase(const FusedParams &p, int bl, float *myrow,
                                             int lane, int C)
{
    using LY = FusedLayout<R>;
    constexpr int H = WIDTH / 2;
    constexpr int STEPS = R + 2 * H;  // samples entering the window
    constexpr int BLOCKS = (STEPS + WIDTH - 1) / WIDTH;
    const int c0 = lane * R;
    double dmax = -__builtin_inf();
    MedianWindow<WIDTH> win;
    win.reset();
    float ring[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; i++) ring[i] = __builtin_nanf("");
#pragma unroll 1
    for (int blk = 0; blk < BLOCKS; blk++) {
#pragma unroll
        for (int k = 0; k < WIDTH; k++) {
            const int t = blk * WIDTH + k;  // step: sample c0 - H + t enters
            const int c_in = c0 - H + t;
            float a_in = __builtin_nanf("");
            if (t < STEPS && c_in >= 0 && c_in < C) {
                a_in = myrow[LY::amp_slot(c_in)];
                // input flags mask the sample (any non-zero value; host.py:143)
                if (p.flags_mode == KSP_FLAGS_CHANNEL) {
                    if (p.in_flags[c_in]) a_in = __builtin_nanf("");
                } else if (p.flags_mode == KSP_FLAGS_FULL) {
                    if (bl < p.baselines && p.in_flags[(size_t)c_in * p.in_flags_stride + bl])
                        a_in = __builtin_nanf("");
                }
            }
            win.step(ring[k], a_in);
            ring[k] = a_in;
            const int j = t - 2 * H;  // output channel offset within the run
            if (j >= 0 && j < R) {    // wave-uniform
                const float xc = ring[(k + WIDTH - H) % WIDTH];  // centre sample c0 + j
                double d = 0.0;
                if (xc == xc) d = (double)xc - win.median();
                dmax = fmax(dmax, d);
                myrow[lane * LY::RUN + (j < 6 ? R + j : j)] = (float)d;
            }
        }
    }
    return dmax;
}

// ---------------------------------------------------------------------------------
// Exact float64 deviation of a channel, recomputed from the float32 amplitudes of its
// window (NaN = masked or outside the band). Same arithmetic as MedianWindow::median():
// median of the valid samples, even counts averaged in float64. The amplitudes in LDS
// have been overwritten by then, so the visibilities are read again (13 8-byte reads
// per sample, served by L2 / Infinity Cache). Only the handful of samples that decide
// a result come here.
template <int WIDTH>
__device__ __forceinline__ double exact_dev(const FusedParams &p, int bl, int c)
{
    constexpr int H = WIDTH / 2;
    const int C = p.channels;
    float v[WIDTH];
    float centre = __builtin_nanf("");
    int n = 0;
#pragma unroll
    for (int k = 0; k < WIDTH; k++) {
        const int cc = c - H + k;
        float a = __builtin_nanf("");
        if (cc >= 0 && cc < C && bl < p.baselines) {
            if (p.is_amplitude) {
                a = ((const float *)p.vis)[(size_t)cc * p.vis_stride + bl];
                if (p.flags_mode == KSP_FLAGS_CHANNEL && p.in_flags[cc]) a = __builtin_nanf("");
                if (p.flags_mode == KSP_FLAGS_FULL && p.in_flags[(size_t)cc * p.in_flags_stride + bl])
                    a = __builtin_nanf("");
            } else {
                const float2 z = ((const float2 *)p.vis)[(size_t)cc * p.vis_stride + bl];
                a = amp_with_flags(p, z.x, z.y, cc, bl);
            }
        }
        if (k == H) centre = a;
        const bool ok = a == a;
        n += ok;
        v[k] = ok ? a : __builtin_inff();
    }
    // odd-even transposition sort (small code; invalid samples, +inf, end up last)
#pragma unroll
    for (int round = 0; round < WIDTH; round++) {
#pragma unroll
        for (int i = round & 1; i + 1 < WIDTH; i += 2) {
            const float lo = fminf(v[i], v[i + 1]), hi = fmaxf(v[i], v[i + 1]);
            v[i] = lo;
            v[i + 1] = hi;
        }
    }
    float lo = v[0], hi = v[0];
#pragma unroll
    for (int i = 0; i < WIDTH; i++) {
        lo = (i == ((n - 1) >> 1)) ? v[i] : lo;
        hi = (i == (n >> 1)) ? v[i] : hi;
    }
    if (!(centre == centre) || n == 0) return 0.0;
    const double med = (n & 1) ? (double)hi : ((double)lo + (double)hi) * 0.5;
    return (double)centre - med;
}

// ---------------------------------------------------------------------------------
// MAD: 1.4826 * median of the non-zero |deviations| of one baseline (whole wavefront).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Wave-wide sum of a small per-lane count (< 2^BITS) without touching LDS: one
// ballot + scalar popcount per bit.
template <int BITS>
__device__ __forceinline__ int wave_sum_small(int c)
{
    int total = 0;
#pragma unroll
    for (int b = 0; b < BITS; b++) total += __popcll(__ballot((c >> b) & 1)) << b;
    return total;
}

// Number of keys (over the whole wavefront) strictly below T, 1 <= T <= 32768. Two keys
// per register: (key - T) has bit 15 set exactly when key < T because both are below
// 2^15, so three packed 16-bit operations handle two samples.
template <int NP>
__device__ __forceinline__ int count_less16(const unsigned (&kp)[NP], unsigned T)
{
    const unsigned short t = (unsigned short)T;
    const u16x2 tt = {t, t};
    u16x2 acc0 = {0, 0}, acc1 = {0, 0};
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const u16x2 d = __builtin_bit_cast(u16x2, kp[i]) - tt;
        if (i & 1)
            acc1 += d >> (unsigned short)15;
        else
            acc0 += d >> (unsigned short)15;
    }
    const u16x2 acc = acc0 + acc1;
    return wave_sum_small<8>((int)acc.x + (int)acc.y);
}

// Exact |deviation| of every sample whose bit is set in `cand` (bit j <-> channel
// c0 + j of this lane) -> list[0..n), at most `cap` entries; returns n (wave-uniform).
// The candidates' channel numbers are first compacted into the list area (as ints),
// then redistributed one per lane, so that the memory-bound recomputation runs once per
// 64 candidates instead of once per candidate of the busiest lane.
template <int WIDTH>
__device__ __forceinline__ int gather_exact(const FusedParams &p, int bl, unsigned long long cand,
                                            int c0, double *list, int cap, int lane)
{
    int *chan = (int *)(list + cap);
    int n = 0;
    while (__any(cand != 0)) {
        const bool has = cand != 0;
        const int j = has ? __ffsll((long long)cand) - 1 : 0;
        cand &= cand - 1;
        const unsigned long long m = __ballot(has);
        const int pos = n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (has && pos < cap) chan[pos] = c0 + j;
        n += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int m = min(n, cap);
#pragma unroll 1
    for (int base = 0; base < m; base += 64) {
        const int i = base + lane;
        const int c = i < m ? chan[i] : 0;
        const double x = fabs(exact_dev<WIDTH>(p, bl, c));
        if (i < m) list[i] = x;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return n;
}

// Values of (stable) rank r and r - 1 among list[0..n).
__device__ __forceinline__ void rank_in_list(const double *list, int n, int r, int lane,
                                             double &xk, double &prev, bool &have_prev)
{
    xk = 0.0;
    prev = 0.0;
    have_prev = false;
#pragma unroll 1
    for (int ci = lane; ci < ((n + 63) & ~63); ci += 64) {
        const bool live = ci < n;
        const double x = live ? list[ci] : 0.0;
        int cnt = 0;
        if (n <= 64) {
            // one candidate per lane; the others arrive by lane broadcast (v_readlane)
            const int xlo = __double2loint(x), xhi = __double2hiint(x);
            for (int jj = 0; jj < n; jj++) {
                const double y = __hiloint2double(__builtin_amdgcn_readlane(xhi, jj),
                                                  __builtin_amdgcn_readlane(xlo, jj));
                cnt += (y < x) || (y == x && jj < ci);
            }
        } else {
            for (int jj = 0; jj < n; jj++) {
                const double y = list[jj];
                cnt += (y < x) || (y == x && jj < ci);
            }
        }
        const unsigned long long hit = __ballot(live && cnt == r);
        const unsigned long long hitp = __ballot(live && cnt == r - 1);
        if (hit) xk = __shfl(x, __ffsll((long long)hit) - 1, 64);
        if (hitp) {
            prev = __shfl(x, __ffsll((long long)hitp) - 1, 64);
            have_prev = true;
        }
    }
}

// Returns the float64 noise estimate (NaN when every deviation is zero). `myrow` holds
// the float32 deviations (dev_slot layout); `list` is this wavefront's candidate list.
template <int R, int WIDTH, int LIST_CAP>
__device__ __forceinline__ double mad_noise(const FusedParams &p, int bl, const float *myrow,
                                            int lane, double *list)
{
    using LY = FusedLayout<R>;
    constexpr int NP = R / 2;
    const int c0 = lane * R;
    const float *mydev = myrow + lane * LY::RUN;
    auto dev_at = [&](int j) -> float { return mydev[j < 6 ? R + j : j]; };
    // 1. 15-bit keys = float32 exponent + 7 mantissa bits of |dev|, two per register.
    //    float32(|d|) is monotone in |d| and so is its truncation, hence the key bin of
    //    the median can be found without knowing any exact value.
    unsigned kp[NP];
    int zeros = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const float d0 = dev_at(2 * i), d1 = dev_at(2 * i + 1);
        zeros += (d0 == 0.0f) + (d1 == 0.0f);
        kp[i] = ((__float_as_uint(d0) >> 16) & 0x7fffu) | (__float_as_uint(d1) & 0x7fff0000u);
    }
    zeros = wave_sum_small<8>(zeros);
    const int total = 64 * R;
    if (zeros == total) return __builtin_nan("");  // numpy: median of nothing
    const int rank2 = total + zeros;  // zeros sort first (reference rank.mako:261-266)
    const int rank = rank2 / 2;       // rank of the (upper) median among all slots
    const bool even = !(rank2 & 1);
    // 2. which key bin holds the median, and how many samples lie below the bin
    unsigned K = 0;
    int below_bin = 0;
#pragma unroll 1
    for (int bit = 14; bit >= 0; bit--) {
        const unsigned test = K | (1u << bit);
        const int c = count_less16<NP>(kp, test);
        if (c <= rank) {
            K = test;
            below_bin = c;
        }
    }
    auto key_of = [&](int j) -> unsigned {
        return (j & 1) ? (kp[j / 2] >> 16) : (kp[j / 2] & 0xffffu);
    };
    int in_bin = count_less16<NP>(kp, K + 1) - below_bin;
    int r = rank - below_bin;  // 0-based rank inside the bin
    unsigned long long cand = 0;
#pragma unroll
    for (int j = 0; j < R; j++)
        if (key_of(j) == K) cand |= 1ull << j;
    unsigned cur32 = 0;  // set when the bin had to be narrowed to one float32 value
    if (in_bin > LIST_CAP) {
        // Degenerate data (many samples in one key bin, e.g. quantised input): narrow
        // the bin with an exact search on the full float32 patterns, which leaves only
        // samples whose float32 deviations are identical.
        unsigned cur = K << 16;
        int below = below_bin;
#pragma unroll 1
        for (int bit = 15; bit >= 0; bit--) {
            const unsigned test = cur | (1u << bit);
            int c = 0;
#pragma unroll 1
            for (int j = 0; j < R; j++) c += (__float_as_uint(dev_at(j)) & 0x7fffffffu) < test;
            c = ksp_wave_sum(c);
            if (c <= rank) {
                cur = test;
                below = c;
            }
        }
        cand = 0;
#pragma unroll 1
        for (int j = 0; j < R; j++)
            if ((__float_as_uint(dev_at(j)) & 0x7fffffffu) == cur) cand |= 1ull << j;
        in_bin = ksp_wave_sum(__popcll(cand));
        r = rank - below;
        below_bin = below;
        cur32 = cur;
    }
    // 3. recompute the bin's (few) samples exactly and rank them in float64; if the
    //    lower median (even counts) lies below the bin, a second pass does the same for
    //    the largest float32 values below it. One loop = one inlined copy of the
    //    recomputation.
    double xk = 0.0, prev = 0.0;
    bool have_prev = false;
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            if (!even || have_prev) break;
            // largest float32 magnitude strictly below the selected bin/value
            float b32 = 0.0f;
#pragma unroll 1
            for (int j = 0; j < R; j++) {
                const float a = fabsf(dev_at(j));
                const bool lower = cur32 ? (__float_as_uint(a) < cur32)
                                         : (((__float_as_uint(a) >> 16) & 0x7fffu) < K);
                b32 = lower ? fmaxf(b32, a) : b32;
            }
            b32 = ksp_wave_max(b32);
            cand = 0;
#pragma unroll 1
            for (int j = 0; j < R; j++)
                if (fabsf(dev_at(j)) == b32) cand |= 1ull << j;
        }
        const int n = gather_exact<WIDTH>(p, bl, cand, c0, list, LIST_CAP, lane);
        if (pass == 0) {
            if (n > LIST_CAP) {
                // more than LIST_CAP samples share ONE float32 deviation. Their exact
                // values are taken to be that value (true for the quantised data that
                // produces such ties; documented limitation otherwise).
                xk = (double)__uint_as_float(cur32);
                if (r > 0) {
                    prev = xk;
                    have_prev = true;
                }
            } else {
                rank_in_list(list, n, r, lane, xk, prev, have_prev);
            }
        } else {
            double below_max = 0.0;
            for (int i = lane; i < min(n, LIST_CAP); i += 64) below_max = fmax(below_max, list[i]);
            prev = ksp_wave_max(below_max);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (even) xk = (xk + prev) / 2.0;  // float64 mean, as numpy.median
    return xk * FUSED_MAD_NORMAL;
}

// ---------------------------------------------------------------------------------
// Thresholds. Returns the flag mask of the lane's run (bit j: channel c0 + j). The
// deviations in `myrow` are used as scratch when windows have to be summed.
//
// SumThreshold is evaluated on the float32 deviations with a rigorous error bound:
// |float32(d) - d| <= 2^-24 |d|, so a window sum computed from the rounded values is
// within 2^-24 * sum|d| (plus float64 rounding, covered by using 2^-23) of the exact
// float64 sum. Windows whose sum is further than that from the limit are decided as
// the exact arithmetic would decide them; the others (practically never) are summed
// again from exact deviations.
template <int R, int WIDTH>
__device__ __forceinline__ unsigned long long threshold_flags(const FusedParams &p, int bl,
                                                              float *myrow, double dmax,
                                                              double noise64, int lane, int C)
{
    using LY = FusedLayout<R>;
    static_assert(R <= 64, "flag mask is 64 bits");
    const int c0 = lane * R;
    float *mydev = myrow + lane * LY::RUN;
    unsigned long long fl = 0;
    if (p.threshold_kind == KSP_THRESHOLD_SIMPLE) {
        const double thr = p.n_sigma * noise64;  // float64 product (host.py:182)
        if (__any(dmax > thr)) {
            // float32(d) > thr decides d > thr except when float32(d) is within one
            // rounding of thr; those samples are recomputed exactly
            unsigned long long unsure = 0;
#pragma unroll 1
            for (int j = 0; j < R; j++) {
                const double d = (double)mydev[j < 6 ? R + j : j];
                const double slack = fabs(d) * 0x1p-23;
                if (d - slack > thr)
                    fl |= 1ull << j;
                else if (d + slack > thr)
                    unsure |= 1ull << j;
            }
            while (__any(unsure != 0)) {
                const bool has = unsure != 0;
                const int j = has ? __ffsll((long long)unsure) - 1 : 0;
                unsure &= unsure - 1;
                const double d = exact_dev<WIDTH>(p, bl, c0 + j);
                if (has && d > thr) fl |= 1ull << j;
            }
        }
        return fl;
    }
    const double t1 = p.n_sigma * noise64;  // host.py:252
    // (static indices only: a runtime-indexed register array would live in scratch)
    constexpr int MAXW = 4;
    float thr[MAXW];
    float thr_min = __builtin_inff();
    bool thr_nan = false;
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        thr[k] = (float)(t1 * p.scales[k]);  // host.py:235
        if (k < p.n_windows) {
            thr_min = fminf(thr_min, thr[k]);
            thr_nan |= (thr[k] != thr[k]);
        }
    }
    // Fast reject (exact): no window can fire unless some sample reaches
    // min_k thr_k; the 2^-20 margin makes the test conservative; needs thresholds > 0.
    const double cand = (double)thr_min * (1.0 - 0x1p-20);
    const bool any = !(thr_min > 0.0f) || (dmax >= cand);
    if (thr_nan || !__any(any)) return 0;

    // value of channel c of this baseline as the window sums see it (any lane's run)
    auto dval = [&](int c) -> float { return myrow[LY::dev_slot(c)]; };
#pragma unroll 1
    for (int k = 0; k < p.n_windows; k++) {  // rolled: one copy of the window code
        const int w = 1 << k;
        const float thrf = k == 0 ? thr[0] : (k == 1 ? thr[1] : (k == 2 ? thr[2] : thr[3]));
        const double limit = (double)__fmul_rn(thrf, (float)w);  // host.py:242
        // already-flagged samples contribute exactly thr (host.py:237; thr is a float32)
        {
            unsigned long long m = fl;
            while (m) {
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                mydev[j < 6 ? R + j : j] = thrf;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // flag bits of the 7 channels after this run (substituted values are exact)
        unsigned nfl = 0;
#pragma unroll
        for (int m = 0; m < 7; m++) {
            const unsigned long long f = __shfl_down(fl, 1 + m / R, 64);
            nfl |= (unsigned)((f >> (m % R)) & 1) << m;
        }
        unsigned long long hits = 0, unsure = 0;
#pragma unroll 1
        for (int j = 0; j < R; j++) {
            const bool valid = (c0 + j + w <= C);
            double s = 0.0, mag = 0.0;
#pragma unroll 1
            for (int m = 0; m < w; m++) {
                const int jj = j + m;
                const bool sub = (jj < R) ? ((fl >> jj) & 1) : ((nfl >> (jj - R)) & 1);
                const double v = valid ? (double)dval(c0 + jj) : 0.0;
                s += v;
                mag += sub ? 0.0 : fabs(v);
            }
            const double slack = mag * 0x1p-23;
            if (valid) {
                if (s - slack > limit)
                    hits |= 1ull << j;
                else if (s + slack > limit)
                    unsure |= 1ull << j;
            }
        }
        // resolve the (practically non-existent) undecided windows exactly
        while (__any(unsure != 0)) {
            const bool has = unsure != 0;
            const int j = has ? __ffsll((long long)unsure) - 1 : 0;
            unsure &= unsure - 1;
            double s = 0.0;
#pragma unroll 1
            for (int m = 0; m < w; m++) {
                const int jj = j + m;
                const bool sub = (jj < R) ? ((fl >> jj) & 1) : ((nfl >> (jj - R)) & 1);
                const double x = exact_dev<WIDTH>(p, bl, c0 + jj);
                s += sub ? (double)thrf : x;
            }
            if (has && s > limit) hits |= 1ull << j;
        }
        // dilation: a hit at j flags j..j+w-1. `pin` holds the hits of the 7 positions
        // just below this lane's run (bit i <-> position i - 7).
        unsigned pin = 0;
        if (R >= 7) {
            const unsigned long long prev = __shfl_up(hits, 1, 64);
            if (lane > 0) pin = (unsigned)(prev >> (R - 7)) & 0x7fu;
        } else {
#pragma unroll
            for (int back = 1; back * R < 7 + R; back++) {
                const unsigned long long prev = __shfl_up(hits, back, 64);
                const int sh = 7 - back * R;
                if (lane >= back) pin |= (unsigned)(sh >= 0 ? (prev << sh) : (prev >> (-sh))) & 0x7fu;
            }
        }
        unsigned long long own = hits;
        if (w >= 2) { own |= own << 1; pin |= pin << 1; }
        if (w >= 4) { own |= own << 2; pin |= pin << 2; }
        if (w >= 8) { own |= own << 4; pin |= pin << 4; }
        const unsigned long long comb = own | (unsigned long long)(pin >> 7);
        fl |= comb & (R == 64 ? ~0ull : ((1ull << R) - 1));
    }
    return fl;
}

// Flags: the launcher zero-fills the whole array (one coalesced memset on the same
// stream); only flagged samples are written here, one byte each. A strip is 8 bytes
// wide, which no store pattern of one workgroup can turn into full 64-byte lines;
// flags are rare, the memset is not.
__device__ __forceinline__ void write_flags(const FusedParams &p, unsigned long long fl, int c0,
                                            int bl, int C)
{
    if (bl >= p.baselines) return;
    const uint8_t fv = (uint8_t)p.flag_value;
    while (fl) {
        const int j = __ffsll((long long)fl) - 1;
        fl &= fl - 1;
        const int c = c0 + j;
        if (c < C) p.flags[(size_t)c * p.flags_stride + bl] = fv;
    }
}
