// Fused single-pass RFI flagger for MI355X (gfx950).
//
// The reference runs five kernels (background -> transpose -> madnz_t ->
// threshold_sum -> transpose; reference rfi/device.py:1152-1164) and moves 31 bytes
// per sample through device memory. Here one launch reads each visibility once
// (8 B) and writes each flag once (1 B); everything in between stays on chip.
//
// Work decomposition ("strip" = 8 adjacent baselines x all channels):
//   * one 512-thread workgroup (8 wavefronts) per strip, one workgroup per CU;
//   * LOAD: all 512 threads read the strip as 64-byte row segments (4 lanes x
//     16 B per channel row), compute numpy's |z| and park the float32 amplitudes
//     in LDS, transposed to [baseline][channel] with a padded, conflict-free
//     layout (136 KiB of the CU's 160 KiB);
//   * from then on wavefront w owns baseline w and lane l owns a run of R
//     consecutive channels, so the sliding median (median_window.h), the MAD
//     selection and SumThreshold are all wave-local: no barriers, cross-lane
//     traffic only through shuffles;
//   * deviations are kept in float64 registers (R per lane): the host path is
//     float64 after the amplitude (reference rfi/host.py:148-163, 235-245), and
//     flags must be bit-identical to it;
//   * MAD: 31-pass bit-wise search on float32-rounded keys (rounding is monotone,
//     so the k-th smallest key is the rounded k-th smallest value), then the
//     exact float64 value is recovered from the (normally single) tied element;
//   * SumThreshold: a window can only fire if some sample reaches the smallest
//     threshold, so a wavefront first tests max(dev) against it (exact argument
//     in DESIGN.md) and only runs the float64 window sums when that is possible;
//   * flags are staged through LDS and written as 8-byte row segments; the
//     blockIdx -> strip map puts the 8 strips that share each 64-byte output line
//     on one XCD so that their partial lines merge in that XCD's L2.
//
// Roofline: HBM, 9 algorithmic bytes per sample (8 read + 1 written).
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

template <int R>
struct FusedLayout {
    static constexpr int RUN = R + 4;                 // lane run padded to keep b128 reads aligned
    static constexpr int ROW = 64 * RUN + 8;          // +8: rows of different baselines 2-way at worst
    static constexpr int LDS_FLOATS = FUSED_STRIP * ROW;
    __device__ static __forceinline__ int index(int c) { return (c / R) * RUN + (c % R); }
};

__device__ __forceinline__ double shfl_down_f64(double v, int delta)
{
    return __shfl_down(v, delta, 64);
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Monotone 15-bit key of a non-negative double: float32-style exponent (8 bits) and
// the top 7 mantissa bits, clamped. Zero maps to key 0.
__device__ __forceinline__ unsigned key16_of(double a)
{
    const int hi = __double2hiint(a);
    int k = (hi >> 13) - (896 << 7);  // rebias 11-bit exponent to 8 bits
    k = min(max(k, 0), 32767);
    return (unsigned)k;
}

// Wave-wide sum of a small per-lane count (< 2^BITS) without touching LDS: one
// ballot + scalar popcount per bit.
template <int BITS>
__device__ __forceinline__ int wave_sum_small(int c)
{
    int total = 0;
#pragma unroll
    for (int b = 0; b < BITS; b++)
        total += __popcll(__ballot((c >> b) & 1)) << b;
    return total;
}

// Selection keys live in the wave's (by then dead) amplitude row in LDS, two 16-bit
// keys per word, laid out so that lane l reads chunk t as one 16-byte access at
// word (t * 64 + l) * 4: consecutive lanes, consecutive 16-byte slots, no conflicts.
template <int R>
struct KeyStore {
    static constexpr int CHUNKS = (R / 2 + 3) / 4;       // 16-byte chunks per lane
    static constexpr int WORDS = CHUNKS * 64 * 4;        // words used in the row
    static constexpr int PER_CHUNK = (R / 2 >= 4) ? 4 : R / 2;  // packed words per chunk
};

// Number of keys (over the whole wave) strictly below T, 1 <= T <= 32768. Two keys
// per word: (key - T) has bit 15 set exactly when key < T because both are below
// 2^15, so three packed 16-bit operations handle two samples.
template <int R>
__device__ __forceinline__ int count_less16(const uint4 *keys, int lane, unsigned T)
{
    using KS = KeyStore<R>;
    const unsigned short t = (unsigned short)T;
    const u16x2 tt = {t, t};
    u16x2 acc0 = {0, 0}, acc1 = {0, 0};
#pragma unroll
    for (int c = 0; c < KS::CHUNKS; c++) {
        const uint4 kk = keys[c * 64 + lane];
        const unsigned w[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
        for (int e = 0; e < KS::PER_CHUNK; e++) {
            const u16x2 d = __builtin_bit_cast(u16x2, w[e]) - tt;
            if (e & 1)
                acc1 += d >> (unsigned short)15;
            else
                acc0 += d >> (unsigned short)15;
        }
    }
    const u16x2 acc = acc0 + acc1;
    const int cnt = (int)acc.x + (int)acc.y;
    return wave_sum_small<8>(cnt);
}

template <int R, int WIDTH>
__global__ __launch_bounds__(FUSED_THREADS, 2) void flagger_fused_kernel(const FusedParams p)
{
    using LY = FusedLayout<R>;
    constexpr int H = WIDTH / 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int C = p.channels;

    // ---- blockIdx -> strip, XCD-aware (speed only; any bijection is correct) ----
    int strip = blockIdx.x;
    {
        const int full = (p.n_strips / 64) * 64;
        if ((int)blockIdx.x < full) {
            const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
            strip = ((i >> 3) * 8 + xcd) * 8 + (i & 7);
        }
    }
    const int b0 = strip * FUSED_STRIP;

    // ---- LOAD: vis -> amplitude -> LDS [baseline][channel] -----------------------
    // Each lane owns a pair of baselines (16 B of a 64-byte row segment) and every
    // 128th row. Loads are issued in batches of LB rows before any is consumed, so a
    // CU keeps 512 x LB x 16 B in flight.
    {
        constexpr int LB = (R >= 32) ? R / 2 : 8;  // R=64: all 32 rows of a lane at once
        constexpr int RSTEP = FUSED_THREADS / 4;  // rows covered per pass
        const int q = tid & 3;    // which pair of baselines
        const int r0 = tid >> 2;  // row within a pass of 128 rows
        const int bl = b0 + 2 * q;
        const bool ok0 = bl < p.baselines, ok1 = bl + 1 < p.baselines;
        const bool plain = !p.is_amplitude && ok1;  // the common, fully vectorised case
        for (int rbase = r0; rbase < C; rbase += RSTEP * LB) {
            float4 raw[LB];
            if (plain) {
#pragma unroll
                for (int u = 0; u < LB; u++) {
                    const int row = rbase + u * RSTEP;
                    raw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < C)
                        raw[u] = *(const float4 *)((const float2 *)p.vis +
                                                   (size_t)row * p.vis_stride + bl);
                }
            }
#pragma unroll
            for (int u = 0; u < LB; u++) {
                const int row = rbase + u * RSTEP;
                if (row >= C) break;
                float a0 = __builtin_nanf(""), a1 = __builtin_nanf("");
                if (plain) {
                    a0 = ksp_abs_c64(raw[u].x, raw[u].y);
                    a1 = ksp_abs_c64(raw[u].z, raw[u].w);
                } else if (p.is_amplitude) {
                    const float *src = (const float *)p.vis + (size_t)row * p.vis_stride + bl;
                    if (ok1) {
                        const float2 v = *(const float2 *)src;
                        a0 = v.x;
                        a1 = v.y;
                    } else if (ok0)
                        a0 = src[0];
                } else if (ok0) {
                    const float2 v = ((const float2 *)p.vis)[(size_t)row * p.vis_stride + bl];
                    a0 = ksp_abs_c64(v.x, v.y);
                }
                if (p.flags_mode == KSP_FLAGS_CHANNEL) {
                    if (p.in_flags[row]) a0 = a1 = __builtin_nanf("");
                } else if (p.flags_mode == KSP_FLAGS_FULL) {
                    const uint8_t *f = p.in_flags + (size_t)row * p.in_flags_stride + bl;
                    if (ok0 && f[0]) a0 = __builtin_nanf("");
                    if (ok1 && f[1]) a1 = __builtin_nanf("");
                }
                const int idx = LY::index(row);
                lds[(2 * q) * LY::ROW + idx] = a0;
                lds[(2 * q + 1) * LY::ROW + idx] = a1;
            }
        }
    }
    __syncthreads();
    if (p.debug_stop == 1) return;

    // ---- per-baseline, wave-local part --------------------------------------------
    const int bl = b0 + wave;
    float *myrow = lds + wave * LY::ROW;
    const int c0 = lane * R;
    auto amp_at = [&](int c) -> float {
        return (c >= 0 && c < C) ? myrow[LY::index(c)] : __builtin_nanf("");
    };

    // |deviation| in float64 registers, signs in a bit mask: the MAD search can then
    // order samples by the high word of the IEEE pattern without a separate key array.
    double adev[R];
    unsigned long long neg = 0;      // bit j: deviation of channel c0 + j is negative
    double dmax = -__builtin_inf();  // largest signed deviation of this lane
    {
        MedianWindow<WIDTH> win;
        win.reset();
        float ring[WIDTH];
#pragma unroll
        for (int i = 0; i < WIDTH; i++) ring[i] = __builtin_nanf("");
        // warm-up: samples c0-H .. c0+H-1 (ring slots 0 .. 2H-1)
#pragma unroll
        for (int k = 0; k < 2 * H; k++) {
            const float a = amp_at(c0 - H + k);
            win.step(ring[k % WIDTH], a);
            ring[k % WIDTH] = a;
        }
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int k = 2 * H + j;  // step number; entering sample is c0 + H + j
            const float a = (j + H < R) ? myrow[lane * LY::RUN + j + H] : amp_at(c0 + H + j);
            const float a_in = (c0 + H + j < C) ? a : __builtin_nanf("");
            win.step(ring[k % WIDTH], a_in);
            ring[k % WIDTH] = a_in;
            const float xc = ring[(k + WIDTH - H) % WIDTH];  // centre sample c0 + j
            double d = 0.0;
            if (xc == xc) d = (double)xc - win.median();
            dmax = fmax(dmax, d);
            if (d < 0.0) neg |= 1ull << j;
            adev[j] = fabs(d);
        }
    }
    auto signed_dev = [&](int j) -> double { return ((neg >> j) & 1) ? -adev[j] : adev[j]; };
    if (p.debug_stop == 2) {
        double acc = dmax;
#pragma unroll
        for (int j = 0; j < R; j++) acc += adev[j];
        if (acc == 12345.678 && p.noise) p.noise[0] = (float)neg;  // keep the work alive
        return;
    }

    // ---- MAD: median of non-zero |dev| -------------------------------------------
    // 1. 15-pass bit-wise search on packed 16-bit keys finds the key bin K that holds
    //    the median and the number of samples below the bin;
    // 2. the (few) samples of bin K are copied to this wave's LDS row and ranked
    //    exactly in float64;
    // 3. degenerate data (hundreds of samples in one bin) falls back to an exact
    //    search on the float64 bit patterns.
    double noise64;
    {
        using KS = KeyStore<R>;
        uint4 *keys = (uint4 *)myrow;  // the amplitude row is dead from here on
        int zeros = 0;
#pragma unroll
        for (int c = 0; c < KS::CHUNKS; c++) {
            unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < KS::PER_CHUNK; e++) {
                const int j = (c * 4 + e) * 2;
                zeros += (adev[j] == 0.0) + (adev[j + 1] == 0.0);
                w[e] = key16_of(adev[j]) | (key16_of(adev[j + 1]) << 16);
            }
            keys[c * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        zeros = wave_sum_small<8>(zeros);
        if (p.debug_stop == 31) { if (zeros == -7 && p.noise) p.noise[0] = 1.f; return; }
        const int total = 64 * R;
        if (zeros == total) {
            noise64 = __builtin_nan("");  // numpy: median of nothing
        } else {
            const int rank2 = total + zeros;
            const int rank = rank2 / 2;  // rank of the (upper) median among all slots
            const bool even = !(rank2 & 1);
            unsigned K = 0;
            int below_bin = 0;
            for (int bit = 14; bit >= 0; bit--) {
                const unsigned test = K | (1u << bit);
                const int c = count_less16<R>(keys, lane, test);
                if (c <= rank) {
                    K = test;
                    below_bin = c;
                }
            }
            if (p.debug_stop == 32) { if (K == 99999u && p.noise) p.noise[0] = (float)below_bin; return; }
            const int in_bin = count_less16<R>(keys, lane, K + 1) - below_bin;
            const int r = rank - below_bin;  // 0-based rank inside the bin
            double xk, prev;
            bool have_prev = false;
            // candidate list: float64, behind the keys in the same row
            double *list = (double *)(myrow + KS::WORDS);
            constexpr int LIST_CAP = (LY::ROW - KS::WORDS) / 2;
            constexpr int MAX_LIST = LIST_CAP < 512 ? LIST_CAP : 512;
            if (in_bin <= MAX_LIST) {
                int base = 0;
#pragma unroll
                for (int c = 0; c < KS::CHUNKS; c++) {
                    const uint4 kk = keys[c * 64 + lane];
                    const unsigned w[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
                    for (int e = 0; e < KS::PER_CHUNK; e++) {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int j = (c * 4 + e) * 2 + h;
                            const unsigned k16 = h ? (w[e] >> 16) : (w[e] & 0xffffu);
                            const bool is = (k16 == K);
                            const unsigned long long m = __ballot(is);
                            if (m) {  // wave-uniform, rarely taken
                                const int pos = base + __builtin_amdgcn_mbcnt_hi(
                                                           (unsigned)(m >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                                if (is) list[pos] = adev[j];
                                base += __popcll(m);
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (p.debug_stop == 33) { if (base == -7 && p.noise) p.noise[0] = (float)list[0]; return; }
                // exact stable rank of every listed value; rank r is the upper median
                xk = 0.0;
                prev = 0.0;
                if (in_bin <= 64) {
                    // one candidate per lane; the others arrive by lane broadcast
                    const bool live = lane < in_bin;
                    const double x = live ? list[lane] : 0.0;
                    int cnt = 0;
                    const int xlo = __double2loint(x), xhi = __double2hiint(x);
                    for (int jj = 0; jj < in_bin; jj++) {
                        // uniform source lane: v_readlane_b32, no LDS round trip
                        const double y = __hiloint2double(__builtin_amdgcn_readlane(xhi, jj),
                                                          __builtin_amdgcn_readlane(xlo, jj));
                        cnt += (y < x) || (y == x && jj < lane);
                    }
                    const unsigned long long hit = __ballot(live && cnt == r);
                    const unsigned long long hitp = __ballot(live && cnt == r - 1);
                    xk = __shfl(x, __ffsll((long long)hit) - 1, 64);
                    if (hitp) {
                        prev = __shfl(x, __ffsll((long long)hitp) - 1, 64);
                        have_prev = true;
                    }
                } else {
                    for (int ci = lane; ci < ((in_bin + 63) & ~63); ci += 64) {
                        const bool live = ci < in_bin;
                        const double x = live ? list[ci] : 0.0;
                        int cnt = 0;
                        for (int jj = 0; jj < in_bin; jj++) {
                            const double y = list[jj];
                            cnt += (y < x) || (y == x && jj < ci);
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
                __builtin_amdgcn_wave_barrier();
                if (p.debug_stop == 34) { if (xk == -7.0 && p.noise) p.noise[0] = (float)prev; return; }
                if (even && !have_prev) {
                    // r == 0: the lower median is the largest value below the bin
                    // (keys are re-read from LDS, not recomputed: keeping 64 keys
                    // alive in registers across the search would spill)
                    double below = 0.0;
#pragma unroll
                    for (int c = 0; c < KS::CHUNKS; c++) {
                        const uint4 kk = keys[c * 64 + lane];
                        const unsigned w[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
                        for (int e = 0; e < KS::PER_CHUNK; e++) {
                            const int j = (c * 4 + e) * 2;
                            below = ((w[e] & 0xffffu) < K) ? fmax(below, adev[j]) : below;
                            below = ((w[e] >> 16) < K) ? fmax(below, adev[j + 1]) : below;
                        }
                    }
                    prev = ksp_wave_max(below);
                }
                if (p.debug_stop == 35) { if (xk == -7.0 && p.noise) p.noise[0] = (float)prev; return; }
                if (even) xk = (xk + prev) / 2.0;  // float64 mean, as numpy.median
            } else {
                // exact search on the high words of the float64 patterns (slow, rare)
                unsigned cur = 0;
                for (int bit = 30; bit >= 0; bit--) {
                    const unsigned test = cur | (1u << bit);
                    int c = 0;
#pragma unroll
                    for (int j = 0; j < R; j++) c += (unsigned)__double2hiint(adev[j]) < test;
                    c = ksp_wave_sum(c);
                    if (c <= rank) cur = test;
                }
                int less = 0, ties = 0;
                double tmin = __builtin_inf(), tmax = 0.0;
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const unsigned hi = (unsigned)__double2hiint(adev[j]);
                    less += hi < cur;
                    ties += hi == cur;
                    tmin = hi == cur ? fmin(tmin, adev[j]) : tmin;
                    tmax = hi == cur ? fmax(tmax, adev[j]) : tmax;
                }
                less = ksp_wave_sum(less);
                ties = ksp_wave_sum(ties);
                tmin = ksp_wave_min(tmin);
                tmax = ksp_wave_max(tmax);
                xk = tmin;
                if (tmin != tmax) {
                    // several distinct values share the high word: walk them in order
                    int rr = rank - less;  // 0-based rank inside the tie set
                    double curv = -1.0;
                    for (int it = 0; it < ties; it++) {
                        double nxt = __builtin_inf();
#pragma unroll
                        for (int j = 0; j < R; j++) {
                            const bool tie = (unsigned)__double2hiint(adev[j]) == cur;
                            if (tie && adev[j] > curv) nxt = fmin(nxt, adev[j]);
                        }
                        nxt = ksp_wave_min(nxt);
                        int cnt = 0;
#pragma unroll
                        for (int j = 0; j < R; j++) cnt += (adev[j] == nxt);
                        cnt = ksp_wave_sum(cnt);
                        xk = nxt;
                        if (rr < cnt) break;
                        rr -= cnt;
                        curv = nxt;
                    }
                }
                if (!(rank2 & 1)) {
                    // even count: mean with the next value down (float64, as numpy.median)
                    int c = 0;
                    double below = 0.0;
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        const bool lt = adev[j] < xk;
                        c += lt;
                        below = lt ? fmax(below, adev[j]) : below;
                    }
                    c = ksp_wave_sum(c);
                    below = ksp_wave_max(below);
                    const double lower = (c == rank) ? below : xk;
                    xk = (xk + lower) / 2.0;
                }
            }
            noise64 = xk * FUSED_MAD_NORMAL;
        }
    }
    if (p.debug_stop == 36) { if (noise64 == -7.0 && p.noise) p.noise[0] = 1.f; return; }
    if (lane == 0 && p.noise != nullptr && bl < p.baselines) p.noise[bl] = (float)noise64;

    // ---- optional deviations output: stage float32 in this wave's LDS row
    // (after MAD, which borrows the row for its candidate list) ---------
    if (p.deviations != nullptr) {
#pragma unroll
        for (int j = 0; j < R; j++) myrow[lane * LY::RUN + j] = (float)signed_dev(j);
    }

    if (p.debug_stop == 3) return;

    // ---- thresholds ----------------------------------------------------------------
    unsigned long long fl = 0;  // bit j: channel c0 + j flagged
    static_assert(R <= 64, "flag mask is 64 bits");
    if (p.threshold_kind == KSP_THRESHOLD_SIMPLE) {
        const double thr = p.n_sigma * noise64;  // float64 product (host.py:182)
        if (__any(dmax > thr)) {
#pragma unroll
            for (int j = 0; j < R; j++)
                if (signed_dev(j) > thr) fl |= 1ull << j;
        }
    } else {
        const double t1 = p.n_sigma * noise64;  // host.py:252
        float thr[KSP_MAX_WINDOWS];
        float thr_min = __builtin_inff();
        bool thr_nan = false;
        for (int k = 0; k < p.n_windows; k++) {
            thr[k] = (float)(t1 * p.scales[k]);  // host.py:235
            thr_min = fminf(thr_min, thr[k]);
            thr_nan |= (thr[k] != thr[k]);
        }
        // Fast reject. A w-sample float64 sum of values all below m is at most
        // w*m, so no window can exceed w*thr_k unless some sample
        // reaches min_k thr_k. The 2^-20 margin makes the test conservative; the
        // bound needs positive thresholds.
        const double cand = (double)thr_min * (1.0 - 0x1p-20);
        const bool any = !(thr_min > 0.0f) || (dmax >= cand);
        if (!thr_nan && __any(any)) {
            // in place: adev becomes the signed, substituted working copy
#pragma unroll
            for (int j = 0; j < R; j++) adev[j] = signed_dev(j);
            for (int k = 0; k < p.n_windows; k++) {
                const int w = 1 << k;
                const double thrd = (double)thr[k];
                const double limit = (double)__fmul_rn(thr[k], (float)w);  // host.py:242
#pragma unroll
                for (int j = 0; j < R; j++)
                    if ((fl >> j) & 1) adev[j] = thrd;  // host.py:237
                // the next lanes' first 7 values (w - 1 <= 7 are used)
                double ext[7];
#pragma unroll
                for (int m = 0; m < 7; m++) ext[m] = shfl_down_f64(adev[m % R], 1 + m / R);
                unsigned long long hits = 0;
#pragma unroll
                for (int j = 0; j < R; j++) {
                    double s = 0.0;
#pragma unroll
                    for (int m = 0; m < 8; m++)
                        if (m < w) s += (j + m < R) ? adev[(j + m) % R] : ext[(j + m >= R) ? (j + m - R) % 7 : 0];
                    const bool valid = (c0 + j + w <= C);
                    if (valid && s > limit) hits |= 1ull << j;
                }
                // dilation: a hit at j flags j..j+w-1. `pin` holds the hits of the 7
                // positions just below this lane's run (bit i <-> position i - 7).
                unsigned pin = 0;
                if (R >= 7) {
                    const unsigned long long prev = __shfl_up(hits, 1, 64);
                    if (lane > 0) pin = (unsigned)(prev >> (R - 7)) & 0x7fu;
                } else {
#pragma unroll
                    for (int back = 1; back * R < 7 + R; back++) {
                        const unsigned long long prev = __shfl_up(hits, back, 64);
                        const int sh = 7 - back * R;
                        if (lane >= back)
                            pin |= (unsigned)(sh >= 0 ? (prev << sh) : (prev >> (-sh))) & 0x7fu;
                    }
                }
                unsigned long long own = hits;
                if (w >= 2) { own |= own << 1; pin |= pin << 1; }
                if (w >= 4) { own |= own << 2; pin |= pin << 2; }
                if (w >= 8) { own |= own << 4; pin |= pin << 4; }
                const unsigned long long comb = own | (unsigned long long)(pin >> 7);
                fl |= comb & (R == 64 ? ~0ull : ((1ull << R) - 1));
            }
        }
    }

    // ---- outputs ---------------------------------------------------------------------
    if (p.debug_stop == 4) {
        if (fl == 0x123456789abcull && p.noise) p.noise[0] = 1.0f;
        return;
    }
    if (p.deviations != nullptr) {
        __syncthreads();  // every wave has staged its float32 deviations
        // rows now hold float32 deviations; write them as [channel][8 baselines]
        const int q = tid & 3, r0 = tid >> 2;
        const int blq = b0 + 2 * q;
        for (int row = r0; row < C; row += FUSED_THREADS / 4) {
            const int idx = LY::index(row);
            const float v0 = lds[(2 * q) * LY::ROW + idx];
            const float v1 = lds[(2 * q + 1) * LY::ROW + idx];
            float *dst = p.deviations + (size_t)row * p.dev_stride + blq;
            if (blq + 1 < p.baselines && (p.dev_stride & 1) == 0)
                *(float2 *)dst = make_float2(v0, v1);
            else {
                if (blq < p.baselines) dst[0] = v0;
                if (blq + 1 < p.baselines) dst[1] = v1;
            }
        }
        __syncthreads();
    }
    // flags: the launcher zero-fills the whole flags array with one coalesced memset
    // before this kernel (same stream), so only flagged samples are written here, one
    // byte each. A strip is 8 bytes wide, which no store pattern of a single
    // workgroup can turn into full 64-byte lines; flags are rare, the memset is not.
    if (bl < p.baselines) {
        const uint8_t fv = (uint8_t)p.flag_value;
        unsigned long long m = fl;
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int c = c0 + j;
            if (c < C) p.flags[(size_t)c * p.flags_stride + bl] = fv;
        }
    }
}

template <int R, int WIDTH>
static int launch_fused(hipStream_t s, const FusedParams &p)
{
    using LY = FusedLayout<R>;
    const size_t lds_bytes = sizeof(float) * LY::LDS_FLOATS;
    auto kern = flagger_fused_kernel<R, WIDTH>;
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        KSP_CHECK(hipFuncSetAttribute((const void *)kern,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    // all flags start at zero; the kernel only writes the (rare) non-zero ones
    KSP_CHECK(hipMemsetAsync(p.flags, 0, (size_t)(p.channels - 1) * p.flags_stride + p.baselines,
                             s));
    hipLaunchKernelGGL(kern, dim3(p.n_strips), dim3(FUSED_THREADS), lds_bytes, s, p);
    KSP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ksp_flagger_fused_supported(int channels, int width, int n_windows)
{
    return channels >= 1 && channels <= 4096 && width == 13 && n_windows >= 1 && n_windows <= 4;
}

extern "C" int ksp_flagger_fused(int device, void *stream, const void *vis,
                                 const uint8_t *in_flags, uint8_t *flags, float *deviations,
                                 float *noise, int channels, int baselines, int vis_stride,
                                 int in_flags_stride, int flags_stride, int dev_stride, int width,
                                 int is_amplitude, int flags_mode, int threshold_kind,
                                 double n_sigma, const double *scales64, int n_windows,
                                 int flag_value)
{
    KSP_REQUIRE(vis != nullptr && flags != nullptr, "NULL buffer");
    KSP_REQUIRE(channels >= 1 && baselines >= 0, "bad shape");
    KSP_REQUIRE(vis_stride >= baselines && flags_stride >= baselines, "stride smaller than row");
    KSP_REQUIRE(deviations == nullptr || dev_stride >= baselines, "bad dev_stride");
    KSP_REQUIRE(flags_mode >= KSP_FLAGS_NONE && flags_mode <= KSP_FLAGS_FULL, "bad flags_mode");
    KSP_REQUIRE(flags_mode == KSP_FLAGS_NONE || in_flags != nullptr, "in_flags is NULL");
    KSP_REQUIRE(flags_mode != KSP_FLAGS_FULL || in_flags_stride >= baselines, "bad in_flags_stride");
    KSP_REQUIRE(threshold_kind == KSP_THRESHOLD_SIMPLE || threshold_kind == KSP_THRESHOLD_SUM,
                "bad threshold_kind");
    KSP_REQUIRE(threshold_kind == KSP_THRESHOLD_SIMPLE || scales64 != nullptr, "scales64 is NULL");
    if (threshold_kind == KSP_THRESHOLD_SIMPLE && n_windows < 1) n_windows = 1;
    if (!ksp_flagger_fused_supported(channels, width, n_windows)) {
        ksp_set_error("ksp_flagger_fused: unsupported configuration (channels=%d width=%d "
                      "n_windows=%d); use the per-stage kernels", channels, width, n_windows);
        return (int)hipErrorNotSupported;
    }
    // 16-byte loads of baseline pairs need even strides and an aligned base
    KSP_REQUIRE((vis_stride & 1) == 0, "vis_stride must be even");
    KSP_REQUIRE(((uintptr_t)vis & 15) == 0, "vis must be 16-byte aligned");
    if (baselines == 0) return 0;
    KSP_CHECK(hipSetDevice(device));

    FusedParams p;
    p.vis = vis;
    p.in_flags = in_flags;
    p.flags = flags;
    p.deviations = deviations;
    p.noise = noise;
    p.channels = channels;
    p.baselines = baselines;
    p.vis_stride = vis_stride;
    p.in_flags_stride = in_flags_stride;
    p.flags_stride = flags_stride;
    p.dev_stride = dev_stride;
    p.is_amplitude = is_amplitude;
    p.flags_mode = flags_mode;
    p.threshold_kind = threshold_kind;
    p.n_windows = n_windows;
    p.flag_value = flag_value;
    p.n_strips = ksp_divup(baselines, FUSED_STRIP);
    {
        const char *dbg = getenv("KSP_FUSED_DEBUG_STOP");
        p.debug_stop = dbg ? atoi(dbg) : 0;
    }
    p.n_sigma = n_sigma;
    for (int k = 0; k < KSP_MAX_WINDOWS; k++)
        p.scales[k] = (scales64 != nullptr && k < n_windows) ? scales64[k] : 0.0;

    hipStream_t s = (hipStream_t)stream;
    if (channels <= 64 * 4) return launch_fused<4, 13>(s, p);
    if (channels <= 64 * 16) return launch_fused<16, 13>(s, p);
    return launch_fused<64, 13>(s, p);
}
