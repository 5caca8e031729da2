// Building blocks shared by the fused flagger kernels (flagger_fused.hip):
// strip layout in LDS, cooperative strip load, the wave-local median / MAD /
// threshold phases, flag output. See DESIGN.md section 4.1 for the reasoning.
#pragma once
#include <stdlib.h>

#include "bitplane.h"
#include "median_merge.h"
#include "median_window.h"

#ifndef FUSED_MAD_BITS
#define FUSED_MAD_BITS 1  // bits decided per step of the MAD's key search (1 or 2)
#endif
#ifndef FUSED_STRIP
#define FUSED_STRIP 4  // baselines per workgroup (one wavefront each); 4 -> two workgroups per CU
#endif
#define FUSED_THREADS (64 * FUSED_STRIP)
#define FUSED_MAD_NORMAL 1.4826

// Diagnostics (phase probes, per-wavefront time stamps) exist only in builds made with
// KSP_EXTRA_HIPCC_FLAGS=-DKSP_DIAG (tools/phase_probe.py, tools/trace_phases.py); in the
// product build these fold to constants and the code behind them disappears.
#ifdef KSP_DIAG_REALTIME  // one 100 MHz clock for the whole chip (phases across CUs line up)
#define FUSED_DIAG_CLOCK() __builtin_amdgcn_s_memrealtime()
#else  // shader-clock cycles (fine-grained, per XCD)
#define FUSED_DIAG_CLOCK() __builtin_amdgcn_s_memtime()
#endif
#ifdef KSP_DIAG
#define FUSED_DIAG_STOP(p) ((p).debug_stop)
#define FUSED_DIAG_TRACE(p) ((p).trace)
#else
#define FUSED_DIAG_STOP(p) 0
#define FUSED_DIAG_TRACE(p) ((unsigned long long *)nullptr)
#endif

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
    // Schedule: workgroups 0 .. n_static-1 take strip strip_of(blockIdx); the remaining
    // dyn_blocks workgroups take the last n_dyn strips from a counter, first come first
    // served (n_dyn = 0 without a workspace). See flagger_fused.hip.
    int n_static, n_dyn, dyn_blocks;
    unsigned *work;  // [0] next dynamic strip, [1] dynamic workgroups finished; zero between launches
    int first_round;  // workgroups resident at once (2 per CU): the first dispatch round
#ifdef KSP_DIAG
    int debug_stop;  // diagnostic builds only (env KSP_FUSED_DEBUG_STOP): 0 = run everything
    unsigned long long *trace;  // diagnostic builds only (env KSP_FUSED_DEBUG_TRACE): phase time stamps
#endif
    double n_sigma;
    double scales[KSP_MAX_WINDOWS];
};

// LDS image of one strip: 8 rows (one per baseline) of float32 amplitudes. Lane l of
// the owning wavefront works on channels [l*R, (l+1)*R); its run is padded by 4 words
// so that 16-byte reads of consecutive lanes fall in consecutive 16-byte slots, and
// rows are offset by 8 words so that the 8 baselines written by one lane group hit
// different banks. After the rows: one candidate list per wavefront for the MAD.
template <int R>
struct FusedLayout {
    static constexpr int RUN = R + 4;
    static constexpr int ROW = 64 * RUN + 8;
    static constexpr int LDS_FLOATS = FUSED_STRIP * ROW;
    static constexpr int LIST_DOUBLES = 256;  // per wavefront
    static constexpr size_t LDS_BYTES =
        sizeof(float) * LDS_FLOATS + sizeof(double) * LIST_DOUBLES * FUSED_STRIP;
    __device__ static __forceinline__ int index(int c) { return (c / R) * RUN + (c % R); }
};

// blockIdx -> strip, XCD-aware: workgroups b, b+8, ... share an XCD, so the 8 strips
// that make up one 512-byte input line / 64-byte output line are given to workgroups
// of one XCD. Speed only -- any bijection is correct.
#ifndef FUSED_XCD_GROUP
#define FUSED_XCD_GROUP 8
#endif
__device__ __forceinline__ int strip_of(int id, int n_strips)
{
    constexpr int G = FUSED_XCD_GROUP;
    const int full = (n_strips / (8 * G)) * (8 * G);
    if (id >= full) return id;
    const int xcd = id & 7, i = id >> 3;
    return ((i / G) * 8 + xcd) * G + (i % G);
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

// Cooperative load of a whole strip: vis -> amplitude -> LDS rows. Each lane owns a
// pair of baselines (16 B of the row segment) and every RSTEP-th row. Requests are
// issued LB rows at a time, one batch ahead of the amplitude arithmetic (double
// buffered in registers).
template <int R>
__device__ __forceinline__ void load_strip(const FusedParams &p, float *lds, int b0, int tid)
{
    using LY = FusedLayout<R>;
    constexpr int LB = 4;
    constexpr int LPR = FUSED_STRIP / 2;            // lanes per row segment
    constexpr int RSTEP = FUSED_THREADS / LPR;      // rows covered per pass
    const int C = p.channels;
    const int q = tid % LPR;
    const int r0 = tid / LPR;
    const int bl = b0 + 2 * q;
    const bool ok0 = bl < p.baselines, ok1 = bl + 1 < p.baselines;
    const bool plain = !p.is_amplitude && ok1;
    // flags of the two samples a lane owns in a row: bit 0 / bit 8 (requested together
    // with the visibilities so that they do not serialise the loop)
    auto request = [&](float4 (&raw)[LB], unsigned (&fl)[LB], int rbase) {
#pragma unroll
        for (int u = 0; u < LB; u++) {
            const int row = rbase + u * RSTEP;
            raw[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            fl[u] = 0;
            if (plain && row < C) {
                raw[u] = *(const float4 *)((const float2 *)p.vis + (size_t)row * p.vis_stride + bl);
                // (only loads here: any arithmetic on the bytes would wait for them)
                if (p.flags_mode == KSP_FLAGS_CHANNEL) {
                    fl[u] = p.in_flags[row];
                } else if (p.flags_mode == KSP_FLAGS_FULL) {
                    fl[u] = *(const unsigned short *)(p.in_flags + (size_t)row * p.in_flags_stride + bl);
                }
            }
        }
    };
    auto finish = [&](const float4 (&raw)[LB], const unsigned (&fl)[LB], int rbase) {
#pragma unroll
        for (int u = 0; u < LB; u++) {
            const int row = rbase + u * RSTEP;
            if (row >= C) break;
            float a0 = __builtin_nanf(""), a1 = __builtin_nanf("");
            unsigned fl8;
            if (plain) {
                a0 = ksp_abs_c64(raw[u].x, raw[u].y);
                a1 = ksp_abs_c64(raw[u].z, raw[u].w);
                if (p.flags_mode == KSP_FLAGS_CHANNEL) fl8 = fl[u] ? 0x101u : 0u; else fl8 = fl[u];
                if (fl8 & 0xffu) a0 = __builtin_nanf("");
                if (fl8 >> 8) a1 = __builtin_nanf("");
            } else if (p.is_amplitude) {
                const float *src = (const float *)p.vis + (size_t)row * p.vis_stride + bl;
                if (ok0) a0 = src[0];
                if (ok1) a1 = src[1];
                if (p.flags_mode == KSP_FLAGS_CHANNEL) {
                    if (p.in_flags[row]) a0 = a1 = __builtin_nanf("");
                } else if (p.flags_mode == KSP_FLAGS_FULL) {
                    const uint8_t *f = p.in_flags + (size_t)row * p.in_flags_stride + bl;
                    if (ok0 && f[0]) a0 = __builtin_nanf("");
                    if (ok1 && f[1]) a1 = __builtin_nanf("");
                }
            } else if (ok0) {
                // last, odd baseline of a ragged strip
                const float2 v = ((const float2 *)p.vis)[(size_t)row * p.vis_stride + bl];
                a0 = amp_with_flags(p, v.x, v.y, row, bl);
            }
            const int idx = LY::index(row);
            lds[(2 * q) * LY::ROW + idx] = a0;
            lds[(2 * q + 1) * LY::ROW + idx] = a1;
        }
    };
    constexpr int BATCH = RSTEP * LB;
    float4 bufa[LB], bufb[LB];
    unsigned fla[LB], flb[LB];
    request(bufa, fla, r0);
    for (int rbase = r0; rbase < C; rbase += 2 * BATCH) {
        if (rbase + BATCH < C) request(bufb, flb, rbase + BATCH);
        finish(bufa, fla, rbase);
        if (rbase + BATCH < C) {
            if (rbase + 2 * BATCH < C) request(bufa, fla, rbase + 2 * BATCH);
            finish(bufb, flb, rbase + BATCH);
        }
    }
    // channels C .. 64 R - 1 do not exist; the median phase expects NaN there
    for (int row = C + r0; row < 64 * R; row += RSTEP) {
        const int idx = LY::index(row);
        lds[(2 * q) * LY::ROW + idx] = __builtin_nanf("");
        lds[(2 * q + 1) * LY::ROW + idx] = __builtin_nanf("");
    }
}

// Fast form of load_strip for complex input and a strip that lies wholly inside the
// array (all but possibly the last one): every lane issues the same straight-line
// sequence of loads -- rows past the end are clamped, their results dropped -- so the
// hardware counters, not conservative waits at branch joins, pace the double buffer.
// MODE is the input-flags mode, fixed per launch.
// Loader geometry (measured best on MI355X: more loads in flight per lane make the
// 32-byte pattern slower, fewer leave the wavefront waiting): a ring of FUSED_LOAD_NB
// batches of FUSED_LOAD_LB rows per lane.
#ifndef FUSED_LOAD_LB
#define FUSED_LOAD_LB 4
#endif
#ifndef FUSED_LOAD_NB
#define FUSED_LOAD_NB 3
#endif
// Returns whether this thread produced any amplitude that takes no part (NaN: flagged
// or NaN input) -- amplitudes are non-negative, so their bit patterns order like
// unsigned integers with the NaNs on top and one integer max per row keeps track.
template <int R, int MODE>
__device__ __forceinline__ bool load_strip_fast(const FusedParams &p, float *lds, int b0, int tid)
{
    using LY = FusedLayout<R>;
    constexpr int LB = FUSED_LOAD_LB;
    constexpr int LPR = FUSED_STRIP / 2;
    constexpr int RSTEP = FUSED_THREADS / LPR;
    constexpr int BATCH = RSTEP * LB;
    const int C = p.channels;
    const int q = tid % LPR;
    const int r0 = tid / LPR;
    const int bl = b0 + 2 * q;
    const float2 *vis = (const float2 *)p.vis + bl;
    const size_t stride = (size_t)p.vis_stride;
    unsigned umax = 0, flag_or = 0;
    auto request = [&](float4 (&raw)[LB], unsigned (&fl)[LB], int rbase) {
#pragma unroll
        for (int u = 0; u < LB; u++) {
            const int row = min(rbase + r0 + u * RSTEP, C - 1);
            raw[u] = *(const float4 *)(vis + (size_t)row * stride);
            fl[u] = 0;
            // (only loads here: any arithmetic on the bytes would wait for them)
            if (MODE == KSP_FLAGS_CHANNEL)
                fl[u] = p.in_flags[row];
            else if (MODE == KSP_FLAGS_FULL)
                fl[u] = *(const unsigned short *)(p.in_flags + (size_t)row * p.in_flags_stride + bl);
        }
    };
    auto finish = [&](const float4 (&raw)[LB], const unsigned (&fl)[LB], int rbase) {
        float amp[LB][2];
        if (FUSED_DIAG_STOP(p) == 11) {  // diagnostic: how long does the loader take without arithmetic?
#pragma unroll
            for (int u = 0; u < LB; u++) {
                amp[u][0] = raw[u].x;
                amp[u][1] = raw[u].z;
            }
        } else {
            ksp_abs_c64_batch<LB, MODE == KSP_FLAGS_NONE>(raw, amp, umax);
        }
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
                const int idx = LY::index(row);
                lds[(2 * q) * LY::ROW + idx] = a0;
                lds[(2 * q + 1) * LY::ROW + idx] = a1;
            }
        }
    };
    // ring of NB batches: NB - 1 are in flight while one is turned into amplitudes
    constexpr int NB = FUSED_LOAD_NB;
    float4 buf[NB][LB];
    unsigned fl[NB][LB];
#pragma unroll
    for (int k = 0; k < NB - 1; k++)
        if (k * BATCH < C) request(buf[k], fl[k], k * BATCH);
    for (int rb = 0; rb < C; rb += NB * BATCH) {  // wave-uniform bounds
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
    // channels C .. 64 R - 1 do not exist; the median phase expects NaN there
    for (int row = C + r0; row < 64 * R; row += RSTEP) {
        const int idx = LY::index(row);
        lds[(2 * q) * LY::ROW + idx] = __builtin_nanf("");
        lds[(2 * q + 1) * LY::ROW + idx] = __builtin_nanf("");
    }
    return umax > 0x7f800000u || flag_or != 0;
}

// ---------------------------------------------------------------------------------
// Median phase: lane l slides the sorted window over its run of R channels of the
// wavefront's baseline. Deviations are kept ROUNDED TO FLOAT32 in registers -- 64
// registers per lane instead of 128 -- together with their maximum. Rounding is
// monotone, so order statistics can be located on the float32 values; the few samples
// whose exact float64 value decides a result are recomputed on demand (exact_dev).
// The LDS row holds NaN for every sample that must not take part (flagged, NaN input,
// channels >= C); channels outside [0, 64 R) are never read.
// General form: the samples are supplied by `amp_rel(i)`, -H <= i < R + H (NaN = takes
// no part, including everything beyond the band).
template <int R, int WIDTH, class Src>
__device__ __forceinline__ void median_phase_src(Src &&amp_rel, float (&dev)[R], float &dmax,
                                                 int *tiny = nullptr)
{
    constexpr int H = WIDTH / 2;
    dmax = -__builtin_inff();
    SortedWindow<WIDTH> win;
    win.reset();
    // the last WIDTH samples (NaN = takes no part). Whether a sample takes part is
    // re-derived from its value when it leaves and when it is the centre: a compare
    // each, where remembering the lane masks would tie up 26 scalar registers.
    float ring[WIDTH];
#pragma unroll
    for (int j = -2 * H; j < R; j++) {
        const int k = 2 * H + j;  // step number; channel (relative) H + j enters
        const float a = amp_rel(H + j);
        if (k < WIDTH) {
            win.step_pad_out(a, a == a);  // what leaves is still the reset padding
        } else {
            float out = ring[k % WIDTH];
            asm("" : "+v"(out));  // opaque: do not carry the entry-time mask along
            win.step(out, out == out, a, a == a);
        }
        ring[k % WIDTH] = a;
        if (j >= 0) {
            float xc = ring[(k + WIDTH - H) % WIDTH];  // centre: relative channel j
            asm("" : "+v"(xc));
            float d = win.deviation(xc);
            d = (xc == xc) ? d : 0.0f;
            dmax = __builtin_amdgcn_fmed3f(dmax, d, win.pinf);  // max, no canonicalise
            dev[j] = d;
        }
    }
    if (tiny != nullptr) *tiny += win.tiny;
}

template <int R, int WIDTH>
__device__ __forceinline__ void median_phase(const float *myrow, int lane, float (&dev)[R],
                                             float &dmax, int *tiny = nullptr)
{
    using LY = FusedLayout<R>;
    constexpr int H = WIDTH / 2;
    const float *run = myrow + lane * LY::RUN;
    const float nan = __builtin_nanf("");
    // amplitude of channel lane*R + i, -H <= i < R + H. When the window reaches no
    // further than the neighbouring lanes' runs, which sit RUN = R + 4 words away,
    // every address is `run` plus a constant.
    auto amp_rel = [&](int i) -> float {
        if (i >= 0 && i < R) return run[i];
        if constexpr (H <= R) {
            if (i < 0) return lane > 0 ? run[i - (LY::RUN - R)] : nan;
            return lane < 63 ? run[i + (LY::RUN - R)] : nan;
        } else {
            const int c = lane * R + i;
            return (c >= 0 && c < 64 * R) ? myrow[LY::index(c)] : nan;
        }
    };
    median_phase_src<R, WIDTH>(amp_rel, dev, dmax, tiny);
}

// Exact float64 deviation of channel c, recomputed from the amplitudes of its window
// (fetch(c) returns the float32 amplitude, NaN if the sample is masked or outside the
// band). Same arithmetic as SortedWindow::deviation(): median of the valid samples, even
// counts averaged in float64. Used only for the handful of samples that decide a
// result: invalid -> +inf, insertion sort (k + 1 min/med3/max for the k-th sample).
template <int WIDTH, class Fetch>
__device__ __forceinline__ double exact_dev(int c, Fetch &&fetch)
{
    constexpr int H = WIDTH / 2;
    float pinf = __builtin_inff(), ninf = -__builtin_inff();
    asm volatile("" : "+v"(pinf), "+v"(ninf));  // opaque: med3 with them stays one instruction
    // (all samples first: where `fetch` reads global memory the loads are then in flight
    // together instead of one per insertion)
    float a[WIDTH];
#pragma unroll
    for (int k = 0; k < WIDTH; k++) a[k] = fetch(c - H + k);
    float v[WIDTH];
    int n = 0;
#pragma unroll
    for (int k = 0; k < WIDTH; k++) {
        const bool ok = a[k] == a[k];
        n += ok;
        const float x = ok ? a[k] : pinf;
        // insert x into the sorted v[0 .. k)
        float out[WIDTH];
        if (k == 0) {
            out[0] = x;
        } else {
            out[0] = __builtin_amdgcn_fmed3f(v[0], x, ninf);  // min
#pragma unroll
            for (int i = 1; i < k; i++) out[i] = __builtin_amdgcn_fmed3f(v[i - 1], x, v[i]);
            out[k] = __builtin_amdgcn_fmed3f(v[k - 1], x, pinf);  // max
        }
#pragma unroll
        for (int i = 0; i <= k; i++) v[i] = out[i];
    }
    const float centre = a[H];
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

__device__ __forceinline__ int wave_max_int(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}

// Number of keys (over the whole wavefront) strictly below T, 1 <= T <= 32768. Two
// 15-bit keys per register, counted with plain 32-bit integer instructions (full rate
// on this chip, where packed 16-bit ones run at half rate): per half-word,
// ((T - 1) | 0x8000) - key lies in [1, 0xffff] -- no borrow crosses into the other
// half -- and has bit 15 set exactly when key < T.
template <int NP>
__device__ __forceinline__ int count_less16(const unsigned (&kp)[NP], unsigned T)
{
    const unsigned tc = ((T - 1) | 0x8000u) * 0x10001u;
    unsigned acc0 = 0, acc1 = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const unsigned f = ((tc - kp[i]) >> 15) & 0x10001u;
        if (i & 1)
            acc1 += f;
        else
            acc0 += f;
    }
    const unsigned acc = acc0 + acc1;  // each half <= NP <= 32
    return wave_sum_small<7>((int)((acc & 0xffffu) + (acc >> 16)));
}

// Append the exact |deviation| of every sample whose bit is set in `cand` (bit j <->
// channel c0 + j of this lane) to `list`, at most `cap` entries; returns the new
// length (wave-uniform). All lanes stay active so that ballots see every lane.
template <int WIDTH, class Fetch>
__device__ __forceinline__ int gather_exact(unsigned long long cand, int c0, double *list,
                                            int base, int cap, Fetch &&fetch)
{
    while (__any(cand != 0)) {
        const bool has = cand != 0;
        const int j = has ? __ffsll((long long)cand) - 1 : 0;
        cand &= cand - 1;
        const double x = fabs(exact_dev<WIDTH>(c0 + j, fetch));
        const unsigned long long m = __ballot(has);
        const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (has && pos < cap) list[pos] = x;
        base += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return base;
}

// Values of (stable) rank r and r - 1 among list[0..n): each lane ranks the entries
// ci = lane, lane + 64, ... against all others.
__device__ __forceinline__ void rank_in_list(const double *list, int n, int r, int lane,
                                             double &xk, double &prev, bool &have_prev)
{
    xk = 0.0;
    prev = 0.0;
    have_prev = false;
    if (n <= 64) {
        // one candidate per lane; the others arrive by lane broadcast (v_readlane)
        const bool live = lane < n;
        const double x = live ? list[lane] : 0.0;
        const int xlo = __double2loint(x), xhi = __double2hiint(x);
        int cnt = 0;
        for (int jj = 0; jj < n; jj++) {
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
        return;
    }
    for (int ci = lane; ci < ((n + 63) & ~63); ci += 64) {
        const bool live = ci < n;
        const double x = live ? list[ci] : 0.0;
        int cnt = 0;
        for (int jj = 0; jj < n; jj++) {
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

// Broadcast from a wave-uniform lane number: one v_readlane_b32 (the LDS-based
// __shfl costs a ds_bpermute round trip each).
__device__ __forceinline__ int ksp_bcast(int v, int src_lane)
{
    return __builtin_amdgcn_readlane(v, src_lane);
}
__device__ __forceinline__ float ksp_bcast(float v, int src_lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
__device__ __forceinline__ double ksp_bcast(double v, int src_lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
                            __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}

// Common case of the MAD's last step: the median's key bin holds n <= 64 samples, one
// per lane (x = exact |deviation|, lanes >= n idle). Each lane ranks its value against
// the others by lane broadcast; returns the values of (stable) rank r and r - 1.
__device__ __forceinline__ void rank_lanes64(double x, int n, int r, int lane, double &xk,
                                             double &prev, bool &have_prev)
{
    const bool live = lane < n;
    // Unrolled in blocks of 8 with a wave-uniform exit: with the lane numbers constants,
    // a broadcast is two v_readlane with immediate lane selects and the tie-break a
    // constant lane mask (a loop over a run-time lane number stalls on every select:
    // 190 cycles per candidate, measured). Idle lanes hold NaN, which never counts.
    const double xs = live ? x : __builtin_nan("");
    int cnt = 0;
#pragma unroll
    for (int blk = 0; blk < 64; blk += 8) {
        if (blk >= n) break;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int jj = blk + u;
            const double y = ksp_bcast(xs, jj);
            cnt += (y < xs) || (y == xs && jj < lane);
        }
    }
    xk = ksp_bcast(x, __ffsll((long long)ksp_ballot(live && cnt == r)) - 1);
    have_prev = r >= 1;
    prev = have_prev ? ksp_bcast(x, __ffsll((long long)ksp_ballot(live && cnt == r - 1)) - 1) : 0.0;
}

// dev[j] for a lane-varying j: a binary tree of selects, one level per bit of j (6 compares
// and 63 v_cndmask; a register array cannot be indexed per lane).
__device__ __forceinline__ float ksp_select64(const float (&dev)[64], int j)
{
    // (v_cndmask written out: left to the optimiser, a select between two array elements
    // becomes an element with a run-time index and the arrays move to scratch memory)
    auto pick = [](unsigned long long mask, float a, float b) -> float {  // mask ? b : a
        float r;
        asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
        return r;
    };
    const unsigned long long m0 = ksp_ballot(j & 1), m1 = ksp_ballot(j & 2), m2 = ksp_ballot(j & 4),
                             m3 = ksp_ballot(j & 8), m4 = ksp_ballot(j & 16), m5 = ksp_ballot(j & 32);
    float t32[32], t16[16], t8[8], t4[4];
#pragma unroll
    for (int i = 0; i < 32; i++) t32[i] = pick(m0, dev[2 * i], dev[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 16; i++) t16[i] = pick(m1, t32[2 * i], t32[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 8; i++) t8[i] = pick(m2, t16[2 * i], t16[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 4; i++) t4[i] = pick(m3, t8[2 * i], t8[2 * i + 1]);
    return pick(m5, pick(m4, t4[0], t4[1]), pick(m4, t4[2], t4[3]));
}
template <int R>
__device__ __forceinline__ float ksp_select_dev(const float (&dev)[R], int j)
{
    if constexpr (R == 64)
        return ksp_select64(dev, j);
    else
        return 0.0f;  // (never asked for: HAVE_EXACT needs R == 64)
}

// Ranking of candidates whose float32 value is the exact one, on the bit patterns (non-negative
// floats order like the integers).
__device__ __forceinline__ void rank_list_u32(const unsigned *vals, unsigned x, int n, int r,
                                              int lane, unsigned &xk, unsigned &prev,
                                              bool &have_prev)
{
    // `vals` (LDS, 16-byte aligned, n <= 64 entries, padded with 0xffffffff to a multiple of
    // 4) holds every candidate; each lane compares its own x with all of them -- broadcast
    // reads and independent compares instead of a chain of lane broadcasts through scalar
    // registers. Equal values are interchangeable (equal float32 values that are exact are
    // equal exact values), so the value of rank k is held by any lane with
    // #(values < x) <= k < #(values <= x).
    const bool live = lane < n;
    int lt = 0, le = 0;
#pragma unroll
    for (int blk = 0; blk < 64; blk += 8) {
        if (blk >= n) break;
        const uint4 a = *(const uint4 *)(vals + blk), b = *(const uint4 *)(vals + blk + 4);
        const unsigned y[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int u = 0; u < 8; u++) {
            lt += y[u] < x;
            le += y[u] <= x;
        }
    }
    xk = (unsigned)__builtin_amdgcn_readlane(
        (int)x, __ffsll((long long)ksp_ballot(live && lt <= r && r < le)) - 1);
    have_prev = r >= 1;
    prev = have_prev ? (unsigned)__builtin_amdgcn_readlane(
                           (int)x, __ffsll((long long)ksp_ballot(live && lt <= r - 1 && r - 1 < le)) - 1)
                     : 0u;
}

// `list` is this wavefront's private candidate list in LDS (LIST_CAP doubles).
// HAVE_EXACT: the caller knows which deviations are exact as they stand -- bit j of `exact`
// set means float64(dev[j]) IS the host's float64 deviation of channel c0 + j (a float32
// difference that did not round, Sterbenz) -- and exact_dev() is then only asked for the
// others: the usual case needs no amplitude at all.
// `top_hint` (R == 64, persistent callers): in, the top 8 bits of the key bin the caller's
// previous baseline ended in, or -1; out, those of this one. The search then first checks,
// with two wave reductions instead of eight, whether the median lies under the same top
// bits (neighbouring baselines mostly have noise of the same binade), and only otherwise
// starts from the top.
// `tiny`: how many of the lane's deviations are +-2^-150 exactly -- not zero, but zero as
// float32 (SortedWindow::tiny). They have key 0 like the zeros and are told apart by count:
// the host ranks them as the smallest non-zero values.
// Returns the float64 noise estimate (NaN when every deviation is zero).
template <int R, int WIDTH, int LIST_CAP, bool HAVE_EXACT = false, class Fetch>
__device__ __forceinline__ double mad_noise(const float (&dev)[R], int lane, double *list,
                                            Fetch &&fetch, int debug_stop = 0,
                                            unsigned long long *trace = nullptr,
                                            unsigned long long exact = 0, int *top_hint = nullptr,
                                            int tiny = 0)
{
    static_assert(!HAVE_EXACT || R == 64, "exactness masks are 64 bits, one per sample of a lane");
    constexpr int NP = R / 2;
    const int c0 = lane * R;
    // dev[j] behind an opaque copy, for the rarely taken paths below: without it the
    // compiler shares their |dev| patterns and keys with the key generation and keeps
    // 128 values alive across the whole search for them
    auto dv = [&](int j) -> float {
        float x = dev[j];
        asm("" : "+v"(x));
        return x;
    };
    // 1. 15-bit keys, two per register: the top 16 bits of |dev|'s float32 pattern,
    //    ROUNDED UP (key = (pattern + 0xffff) >> 16), so that key 0 means exactly zero
    //    and the zeros can be counted from the bit planes below instead of with a
    //    compare per sample. float32(|d|) is monotone in |d| and so is this rounding,
    //    hence the key bin of the median can be found without knowing any exact value.
    //    A deviation is finite or the default NaN (of inf - inf): |pattern| <= 0x7fc00000,
    //    so the addition carries neither into the sign bit nor past it -- it is done on
    //    the signed pattern and the two sign bits are cleared after packing.
    unsigned kp[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const unsigned a = __float_as_uint(dev[2 * i]) + 0xffffu;
        const unsigned b = __float_as_uint(dev[2 * i + 1]) + 0xffffu;
        kp[i] = __builtin_amdgcn_perm(b, a, 0x07060302u) & 0x7fff7fffu;  // bytes 2,3 of each
    }
    int zeros;
    if constexpr (R != 64) zeros = count_less16<NP>(kp, 1);
    auto stamp = [&](int i) {
        if (trace != nullptr && lane == 0) trace[i] = FUSED_DIAG_CLOCK();
    };
    stamp(8);
    if (debug_stop == 31) return (double)(kp[0] + kp[NP - 1]);
    const int total = 64 * R;
    // 2. which key bin holds the median, and how many samples lie below the bin
    unsigned K = 0;
    int below_bin = 0, in_bin;
    int rank2 = 0, rank = 0;  // zeros sort first (reference rank.mako:261-266): rank2 = total + zeros
    int zeros_keys = 0;       // samples with key 0: the zeros and the +-2^-150
    unsigned eq0 = 0, eq1 = 0;  // R == 64: which even / odd samples of the lane have key K
    if constexpr (R == 64) {
        // Bit-sliced search. The lane's 64 keys are transposed into 15 bit planes of
        // 64 bits (np[b]: even samples, np[16 + b]: odd samples; stored inverted), so
        // "how many of the keys that still match the prefix have bit b clear" is two
        // ANDs and two population counts per lane instead of a pass over 32 registers.
        unsigned np[32];
#pragma unroll
        for (int i = 0; i < 32; i++) np[i] = kp[i] ^ 0x7fff7fffu;
        transpose_bits32(np);
        stamp(9);
        // samples with key 0 (= exact zeros): all 15 inverted planes set
        {
            unsigned z0 = np[0], z1 = np[16];
#pragma unroll
            for (int b = 1; b < 15; b++) {
                z0 &= np[b];
                z1 &= np[16 + b];
            }
            zeros = ksp_wave_sum_dpp(__popc(z0) + __popc(z1));
        }
        zeros_keys = zeros;
        if (ksp_any(tiny != 0)) zeros -= ksp_wave_sum_dpp(tiny);
        if (zeros == total) return __builtin_nan("");  // numpy: median of nothing
        rank2 = total + zeros;
        rank = rank2 / 2;
        // (the median among the +-2^-150: both middle values of an even count are, too)
        if (rank < zeros_keys) return 0x1p-150 * FUSED_MAD_NORMAL;
        eq0 = eq1 = 0xffffffffu;
        // two bits per step: the keys still matching the prefix split four ways by
        // (bit1, bit0); three counts decide both bits after ONE round of reductions
        auto step2 = [&](int hi, int lo) {
            const unsigned a0 = eq0 & np[hi], a1 = eq1 & np[16 + hi];      // bit1 clear
            const unsigned z00_0 = a0 & np[lo], z00_1 = a1 & np[16 + lo];  // 00
            const unsigned z01_0 = a0 ^ z00_0, z01_1 = a1 ^ z00_1;         // 01
            const unsigned b0 = eq0 ^ a0, b1 = eq1 ^ a1;                   // bit1 set
            const unsigned z10_0 = b0 & np[lo], z10_1 = b1 & np[16 + lo];  // 10
            const unsigned z11_0 = b0 ^ z10_0, z11_1 = b1 ^ z10_1;         // 11
            int s01 = (__popc(z00_0) + __popc(z00_1)) | ((__popc(z01_0) + __popc(z01_1)) << 16);
            int c10 = __popc(z10_0) + __popc(z10_1);
            ksp_wave_sum2_dpp(s01, c10);  // each field <= 4096
            const int n1 = below_bin + (s01 & 0xffff), n2 = n1 + (s01 >> 16), n3 = n2 + c10;
            // the two bits are the number of thresholds with count(< threshold) <= rank
            // (n1 <= n2 <= n3); selected without branches
            const bool g1 = n1 <= rank, g2 = n2 <= rank, g3 = n3 <= rank;
            K |= (unsigned)((int)g1 + (int)g2 + (int)g3) << lo;
            below_bin = g3 ? n3 : g2 ? n2 : g1 ? n1 : below_bin;
            eq0 = g3 ? z11_0 : g2 ? z10_0 : g1 ? z01_0 : z00_0;
            eq1 = g3 ? z11_1 : g2 ? z10_1 : g1 ? z01_1 : z00_1;
        };
        // one bit: the keys that match the prefix and have this bit clear are counted
        auto step1 = [&](int bit) {
            const unsigned z0 = eq0 & np[bit], z1 = eq1 & np[16 + bit];
            const int c = below_bin + ksp_wave_sum_dpp(__popc(z0) + __popc(z1));
            const bool take = c <= rank;
            K |= take ? (1u << bit) : 0u;
            below_bin = take ? c : below_bin;
            eq0 = take ? (eq0 ^ z0) : z0;
            eq1 = take ? (eq1 ^ z1) : z1;
        };
#if FUSED_MAD_BITS == 1
        bool hinted = false;
        if (top_hint != nullptr && *top_hint >= 0) {
            // follow the hinted top bits down the planes, counting per lane
            const unsigned h = (unsigned)*top_hint;  // (wave-uniform)
            unsigned e0 = 0xffffffffu, e1 = 0xffffffffu;
            int lb = 0;
#pragma unroll
            for (int bit = 14; bit >= 7; bit--) {
                const unsigned z0 = e0 & np[bit], z1 = e1 & np[16 + bit];
                if ((h >> (bit - 7)) & 1) {
                    lb += __popc(z0) + __popc(z1);
                    e0 ^= z0;
                    e1 ^= z1;
                } else {
                    e0 = z0;
                    e1 = z1;
                }
            }
            int cnt = __popc(e0) + __popc(e1);
            ksp_wave_sum2_dpp(lb, cnt);
            if (lb <= rank && rank < lb + cnt) {
                hinted = true;
                K = h << 7;
                below_bin = lb;
                eq0 = e0;
                eq1 = e1;
            }
        }
        if (hinted) {
            // (two bits per step measure 1-2 % slower here as well)
#pragma unroll
            for (int bit = 6; bit >= 0; bit--) step1(bit);
        } else {
#pragma unroll
            for (int bit = 14; bit >= 0; bit--) step1(bit);
        }
        if (top_hint != nullptr) *top_hint = (int)(K >> 7);
#else
#pragma unroll
        for (int bit = 14; bit >= 2; bit -= 2) step2(bit, bit - 1);
        step1(0);
#endif
        in_bin = ksp_wave_sum_dpp(__popc(eq0) + __popc(eq1));
    } else {
        zeros_keys = zeros;
        if (ksp_any(tiny != 0)) zeros -= ksp_wave_sum(tiny);
        if (zeros == total) return __builtin_nan("");
        rank2 = total + zeros;
        rank = rank2 / 2;
        if (rank < zeros_keys) return 0x1p-150 * FUSED_MAD_NORMAL;
        for (int bit = 14; bit >= 0; bit--) {
            const unsigned test = K | (1u << bit);
            const int c = count_less16<NP>(kp, test);
            if (c <= rank) {
                K = test;
                below_bin = c;
            }
        }
        in_bin = count_less16<NP>(kp, K + 1) - below_bin;
    }
    const bool even = !(rank2 & 1);
    stamp(10);
    if (debug_stop == 32) return (double)(K + below_bin);
    // (keys are re-derived from the deviations from here on: kp may die)
    auto key_of = [&](int j) -> unsigned {
        return min((__float_as_uint(dv(j)) & 0x7fffffffu) + 0xffffu, 0x7fffffffu) >> 16;
    };
    auto bin_mask = [&](unsigned key) -> unsigned long long {
        unsigned long long m = 0;
#pragma unroll
        for (int j = 0; j < R; j++)
            if (key_of(j) == key) m |= 1ull << j;
        return m;
    };
    int r = rank - below_bin;  // 0-based rank inside the bin
    unsigned long long cand = 0;
    const bool narrowed = in_bin > LIST_CAP;
    if (narrowed) {
        // Degenerate data (hundreds of samples in one key bin, e.g. quantised input):
        // narrow the bin with an exact search on the full float32 patterns, which
        // leaves only samples whose float32 deviations are identical.
        // (key K >= 1 covers the patterns ((K - 1) << 16) + 1 .. K << 16)
        const unsigned first = ((K - 1) << 16) + 1u;
        unsigned offs = 0;
        int below = below_bin;
        for (int bit = 15; bit >= 0; bit--) {
            const unsigned test = first + (offs | (1u << bit));
            int c = 0;
#pragma unroll
            for (int j = 0; j < R; j++) c += (__float_as_uint(dv(j)) & 0x7fffffffu) < test;
            c = ksp_wave_sum(c);
            if (c <= rank) {
                offs |= 1u << bit;
                below = c;
            }
        }
        const unsigned cur = first + offs;
        cand = 0;
#pragma unroll
        for (int j = 0; j < R; j++)
            if ((__float_as_uint(dv(j)) & 0x7fffffffu) == cur) cand |= 1ull << j;
        in_bin = ksp_wave_sum(__popcll(cand));
        r = rank - below;
        below_bin = below;
        if (in_bin > LIST_CAP) {
            // Still too many: they share ONE float32 value, so their exact float64
            // values all lie within half a float32 ulp of it. Select among them
            // exactly by bisection on the float64 bit patterns (positive doubles order
            // like their patterns; the span is at most ~2^29 patterns), recomputing the
            // exact values on every pass -- a path only heavily quantised data takes.
            auto each = [&](auto &&f) {
                unsigned long long todo = cand;
                while (__any(todo != 0)) {
                    const bool has = todo != 0;
                    const int j = has ? __ffsll((long long)todo) - 1 : 0;
                    todo &= todo - 1;
                    const double x = fabs(exact_dev<WIDTH>(c0 + j, fetch));
                    if (has) f(x);
                }
            };
            double lo = __builtin_inf(), hi = 0.0;
            each([&](double x) {
                lo = fmin(lo, x);
                hi = fmax(hi, x);
            });
            lo = ksp_wave_min(lo);
            hi = ksp_wave_max(hi);
            double xk = lo, prev = lo;
            if (lo != hi) {
                const unsigned long long base = (unsigned long long)__double_as_longlong(lo);
                const unsigned long long span = (unsigned long long)__double_as_longlong(hi) - base;
                unsigned long long offs = 0;
                int below_k = 0;
                for (int bit = 63 - __clzll((long long)span); bit >= 0; bit--) {
                    const unsigned long long test = base + (offs | (1ull << bit));
                    int c = 0;
                    each([&](double x) {
                        c += (unsigned long long)__double_as_longlong(x) < test;
                    });
                    c = ksp_wave_sum(c);
                    if (c <= r) {
                        offs |= 1ull << bit;
                        below_k = c;
                    }
                }
                xk = __longlong_as_double((long long)(base + offs));
                prev = xk;  // rank r - 1 is a duplicate of rank r ...
                if (even && r >= 1 && below_k == r) {
                    // ... unless exactly r values lie below: then it is the largest of them
                    double m = 0.0;
                    each([&](double x) { m = (x < xk) ? fmax(m, x) : m; });
                    prev = ksp_wave_max(m);
                }
            }
            if (even && r == 0) {
                // lower median lies below this value: largest float32 value below it
                float b32 = 0.0f;
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const float a = fabsf(dv(j));
                    b32 = (__float_as_uint(a) < cur) ? fmaxf(b32, a) : b32;
                }
                b32 = ksp_wave_max(b32);
                unsigned long long bm = 0;
#pragma unroll
                for (int j = 0; j < R; j++)
                    if (fabsf(dv(j)) == b32) bm |= 1ull << j;
                const int n2 = gather_exact<WIDTH>(bm, c0, list, 0, LIST_CAP, fetch);
                double below_max = 0.0;
                for (int i = lane; i < min(n2, LIST_CAP); i += 64) below_max = fmax(below_max, list[i]);
                below_max = ksp_wave_max(below_max);
                __builtin_amdgcn_wave_barrier();
                // (more than LIST_CAP samples there too: they share the float32 value
                // b32 and the largest exact one is wanted)
                if (n2 > LIST_CAP) {
                    unsigned long long todo = bm;
                    double m = 0.0;
                    while (__any(todo != 0)) {
                        const bool has = todo != 0;
                        const int j = has ? __ffsll((long long)todo) - 1 : 0;
                        todo &= todo - 1;
                        const double x = fabs(exact_dev<WIDTH>(c0 + j, fetch));
                        m = has ? fmax(m, x) : m;
                    }
                    below_max = ksp_wave_max(m);
                }
                prev = below_max;
            }
            if (even) xk = (xk + prev) / 2.0;
            return xk * FUSED_MAD_NORMAL;
        }
    }
    double xk, prev;
    bool have_prev;
    if (!narrowed && in_bin <= 64) {
        // 3a. usual case: hand the bin's samples out one per lane (float32 value and
        //     channel go through the list), rank on float32, recompute only the ties
        int *lc = (int *)list;
        int n = 0;
        auto place = [&](bool hit, int channel) {
            const unsigned long long m = ksp_ballot(hit);
            const int pos = n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (hit) lc[pos] = channel;
            n += __popcll(m);
        };
        double x;
        bool ranked = false;
        unsigned *lv = (unsigned *)lc + 128;  // HAVE_EXACT: |float32 deviations| of the candidates
        static_assert(LIST_CAP * 2 >= 128 + 64 + 8, "candidate values behind the channel numbers");
        if constexpr (R == 64 && HAVE_EXACT) {
            // every lane knows its own hits (eq0 / eq1): it hands out channel, exactness and
            // |float32 deviation| of each (the deviation through ksp_select64: as many rounds
            // as the busiest lane has hits)
            const int mine = __popc(eq0) + __popc(eq1);
            const int incl = ksp_wave_scan_dpp(mine);
            n = __builtin_amdgcn_readlane(incl, 63);
            int pos = incl - mine;
            unsigned h0 = eq0, h1 = eq1;
            while (ksp_any((h0 | h1) != 0)) {
                const bool has = (h0 | h1) != 0;
                int j = 0;
                if (h0) {
                    j = 2 * (__ffs((int)h0) - 1);
                    h0 &= h0 - 1;
                } else if (h1) {
                    j = 2 * (__ffs((int)h1) - 1) + 1;
                    h1 &= h1 - 1;
                }
                const float v = ksp_select_dev<R>(dev, j);
                if (has) {
                    lc[pos] = (c0 + j) | (((exact >> j) & 1) ? (int)0x80000000 : 0);
                    lv[pos] = __float_as_uint(v) & 0x7fffffffu;
                    pos++;
                }
            }
            // (padding for the 8 values at a time that the ranking reads)
            if (lane < 8) lv[n + lane] = 0xffffffffu;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int slot = lane < n ? lane : 0;
            const int e0 = lc[slot];
            const unsigned a32 = lv[slot];
            x = (double)__uint_as_float(a32);
            const bool inexact = lane < n && e0 >= 0;
            stamp(11);
            if (debug_stop == 33) return x;
            if (ksp_any(inexact)) {
                const double xe = fabs(exact_dev<WIDTH>(e0 & 0x7fffffff, fetch));
                x = inexact ? xe : x;
            } else {
                // every candidate's float32 value is the exact one: rank those
                unsigned k32, p32;
                rank_list_u32(lv, a32, n, r, lane, k32, p32, have_prev);
                xk = (double)__uint_as_float(k32);
                prev = (double)__uint_as_float(p32);
                ranked = true;
            }
        } else {
            if constexpr (R == 64) {
                // every lane knows its own hits (eq0 / eq1); a prefix sum over the lanes
                // gives each lane its first slot in the list and the lanes fill in their
                // (zero to a few) channels on their own -- no per-sample wave operations
                const int mine = __popc(eq0) + __popc(eq1);
                const int incl = ksp_wave_scan_dpp(mine);
                n = __builtin_amdgcn_readlane(incl, 63);
                int pos = incl - mine;
                unsigned h0 = eq0, h1 = eq1;
                while (h0 | h1) {  // divergent: as many rounds as the busiest lane has hits
                    if (h0) {
                        lc[pos++] = c0 + 2 * (__ffs((int)h0) - 1);
                        h0 &= h0 - 1;
                    } else {
                        lc[pos++] = c0 + 2 * (__ffs((int)h1) - 1) + 1;
                        h1 &= h1 - 1;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < R; j++) place(key_of(j) == K, c0 + j);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int c = lc[lane < n ? lane : 0];
            stamp(11);
            if (debug_stop == 33) return (double)c;
            // one exact recomputation serves every candidate (one per lane)
            x = fabs(exact_dev<WIDTH>(c, fetch));
        }
        stamp(12);
        if (!ranked) rank_lanes64(x, n, r, lane, xk, prev, have_prev);
        stamp(13);
        __builtin_amdgcn_wave_barrier();
    } else {
        // 3b. recompute all of the bin's samples exactly and rank them in float64
        gather_exact<WIDTH>(narrowed ? cand : bin_mask(K), c0, list, 0, LIST_CAP, fetch);
        rank_in_list(list, in_bin, r, lane, xk, prev, have_prev);
        __builtin_amdgcn_wave_barrier();
    }
    if (even && !have_prev) {
        // r == 0: the lower median is the largest value below the bin, i.e. the
        // largest exact value of the highest non-empty bin below K
        int k2 = -1;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const int kj = (int)key_of(j);
            k2 = (kj < (int)K) ? max(k2, kj) : k2;
        }
        k2 = wave_max_int(k2);
        // (nothing with a non-zero key below the bin: the value below it is a +-2^-150 --
        // were there none, an even count could not have its lower median down here)
        const bool below_is_tiny = k2 <= 0;
        unsigned long long bm = bin_mask((unsigned)(below_is_tiny ? 0 : k2));
        // within that bin the largest float32 values are enough
        float b32 = 0.0f;
#pragma unroll
        for (int j = 0; j < R; j++)
            if ((bm >> j) & 1) b32 = fmaxf(b32, fabsf(dv(j)));
        b32 = ksp_wave_max(b32);
        unsigned long long top = 0;
#pragma unroll
        for (int j = 0; j < R; j++)
            if (((bm >> j) & 1) && fabsf(dv(j)) == b32) top |= 1ull << j;
        bool recompute = true;
        if (below_is_tiny) {
            prev = 0x1p-150;
            recompute = false;
        }
        if constexpr (HAVE_EXACT) {
            // every sample at that value exact as it stands: the value itself
            if (recompute && !ksp_any((top & ~exact) != 0)) {
                prev = (double)b32;
                recompute = false;
            }
        }
        if (recompute) {
            const int n2 = gather_exact<WIDTH>(top, c0, list, 0, LIST_CAP, fetch);
            double below_max = 0.0;
            for (int i = lane; i < min(n2, LIST_CAP); i += 64) below_max = fmax(below_max, list[i]);
            prev = ksp_wave_max(below_max);
            __builtin_amdgcn_wave_barrier();
            if (n2 > LIST_CAP) {
                // more ties at b32 than the list holds: largest exact value, no list
                double m = 0.0;
                while (__any(top != 0)) {
                    const bool has = top != 0;
                    const int j = has ? __ffsll((long long)top) - 1 : 0;
                    top &= top - 1;
                    const double x = fabs(exact_dev<WIDTH>(c0 + j, fetch));
                    m = has ? fmax(m, x) : m;
                }
                prev = ksp_wave_max(m);
            }
        }
    }
    stamp(14);
    if (even) xk = (xk + prev) / 2.0;  // float64 mean, as numpy.median
    return xk * FUSED_MAD_NORMAL;
}

// exact_dev as a function that is CALLED (threshold_wide's rarely taken path must not add
// its registers to those of the loop around it)
template <int WIDTH, class Fetch>
__device__ __noinline__ double exact_dev_call(int c, Fetch fetch)
{
    return exact_dev<WIDTH>(c, fetch);
}

// ---------------------------------------------------------------------------------
// SumThreshold with up to 8 windows (1 .. 128 channels) for lanes of 64 channels: what
// threshold_flags<.., MAXK = 8> runs instead of its loop over the windows up to 8 (which
// sums every window directly, from a second copy of the deviations). Returns the flags.
//
// The sums of w consecutive values are built by doubling, S_2s[j] = S_s[j] + S_s[j + s],
// in float32 over the UNFLAGGED deviations (U, flagged samples count as 0) and, as one
// byte per position, over the flag bits (N): the host's float64 sum is U* + N thr with
// U* the exact sum, and it exceeds w thr iff U* > (w - N) thr =: T. U differs from U* by
// at most (k + 1) 2^-24 (|U| + 2 w nmax), nmax = the baseline's largest downward deviation
// (k roundings of partial sums and the rounding of the deviations themselves; the sum of
// the absolute values is at most |U| + 2 w nmax). U > T (1 + 2^-18) + 2^-18 w nmax is
// therefore a hit, U <= T (1 - 2^-18) - 2^-18 w nmax is none, and the sliver in between --
// and every window when a threshold is not positive -- is summed again as the host does
// it, sequentially in float64 from exact deviations. U and N depend on the flags only:
// they are kept from one window size to the next (one more doubling) unless flags were
// added in between.
template <int WIDTH, class Fetch>
__device__ __forceinline__ unsigned long long threshold_wide(const FusedParams &p,
                                                             const float (&dev)[64],
                                                             unsigned long long fl,
                                                             unsigned long long ge, double t1,
                                                             bool positive, int lane, int C,
                                                             Fetch &&fetch)
{
    const int c0 = lane * 64;
    float nmax = 0.0f;
#pragma unroll
    for (int j = 0; j < 64; j++) nmax = fmaxf(nmax, -dev[j]);
    nmax = ksp_wave_max(nmax);
    // U1: the unflagged deviations (flags only grow: maintained from its own previous state,
    // so that `dev` is not needed again in here -- three arrays of 64 do not fit the
    // registers, and a spilled one must not be what every rebuild reads)
    float U[64];
    unsigned Nb[16];  // byte j % 4 of Nb[j / 4]: flagged samples in the window at j
#pragma unroll
    for (int j = 0; j < 64; j++) U[j] = dev[j];
#pragma unroll
    for (int q = 0; q < 16; q++) Nb[q] = 0;
    int lvl = -1;  // U, Nb hold the sums over 2^lvl samples (-1: to be rebuilt from the flags)
    float umax = 0.0f;  // the baseline's largest unflagged deviation (as of the last rebuild)
    bool counted = false;  // Nb is kept (some lane has a flag)
    // one doubling, window s -> 2 s, in place: position j takes j + s, by residue class
    // r = j mod s in ascending order, so that the addend is still the old value; the last
    // position of a class takes the next lane's (old) value at r, fetched first
    auto level = [&](auto s_) {
        constexpr int s = decltype(s_)::value;
        ksp_static_for<s>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const float e = __shfl_down(U[r], 1, 64);
#pragma unroll
            for (int j = r; j + s < 64; j += s) U[j] += U[j + s];
            U[r + 64 - s] += e;
        });
        if (counted) {
            if constexpr (s >= 4) {
                constexpr int sq = s / 4;
                ksp_static_for<sq>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    const unsigned e = (unsigned)__shfl_down((int)Nb[r], 1, 64);
#pragma unroll
                    for (int q = r; q + sq < 16; q += sq) Nb[q] += Nb[q + sq];
                    Nb[r + 16 - sq] += e;
                });
            } else {
                const unsigned e = (unsigned)__shfl_down((int)Nb[0], 1, 64);
#pragma unroll
                for (int q = 0; q < 16; q++)
                    Nb[q] += __builtin_amdgcn_alignbyte(q + 1 < 16 ? Nb[(q + 1) % 16] : e, Nb[q], s);
            }
        }
    };
#pragma unroll 1
    for (int k = 0; k < p.n_windows; k++) {
        if (positive && !ksp_any((ge & ~fl) != 0)) break;  // nothing left that could fire
        const int w = 1 << k;
        const double scale = k == 0 ? p.scales[0] : k == 1 ? p.scales[1] : k == 2 ? p.scales[2]
                           : k == 3 ? p.scales[3] : k == 4 ? p.scales[4] : k == 5 ? p.scales[5]
                           : k == 6 ? p.scales[6] : p.scales[7];
        const float thrf = (float)(t1 * scale);  // host.py:235
        const float limf = __fmul_rn(thrf, (float)w);
        const double limit = (double)limf;  // host.py:242
        // a window that fires holds an unflagged sample above thr (one whose float32
        // deviation is at least thr): none in the baseline, nothing to do for this size
        if (positive && lvl >= 0 && umax < thrf) continue;
        if (lvl < 0) {
            counted = ksp_any(fl != 0);
            {
                const unsigned f_lo = (unsigned)fl, f_hi = (unsigned)(fl >> 32);
#pragma unroll
                for (int j = 0; j < 64; j++)
                    U[j] = (((j < 32 ? f_lo : f_hi) >> (j & 31)) & 1u) ? 0.0f : dev[j];
            }
            float m = U[0];
#pragma unroll
            for (int j = 1; j < 64; j++) m = fmaxf(m, U[j]);
            umax = ksp_wave_max(m);
            if (counted) {
#pragma unroll
                for (int q = 0; q < 16; q++)
                    Nb[q] = ((((unsigned)(fl >> (4 * q))) & 0xfu) * 0x00204081u) & 0x01010101u;
            }
            lvl = 0;
            if (positive && umax < thrf) continue;
        }
        while (lvl < k) {
            switch (lvl) {
            case 0: level(std::integral_constant<int, 1>{}); break;
            case 1: level(std::integral_constant<int, 2>{}); break;
            case 2: level(std::integral_constant<int, 4>{}); break;
            case 3: level(std::integral_constant<int, 8>{}); break;
            case 4: level(std::integral_constant<int, 16>{}); break;
            case 5: level(std::integral_constant<int, 32>{}); break;
            default: level(std::integral_constant<int, 64>{}); break;
            }
            lvl++;
        }
        // windows that lie inside the band start at channels <= C - w
        const int nvalid = C - w + 1 - c0;
        const unsigned long long valid = nvalid >= 64 ? ~0ull : nvalid <= 0 ? 0ull : ((1ull << nvalid) - 1);
        unsigned long long hits = 0, unsure = valid;
        if (positive) {
            unsigned h_lo = 0, h_hi = 0, m_lo = 0, m_hi = 0;
            // (+ w 2^-149: a subnormal deviation is rounded by up to 2^-150, not by 2^-24 of itself)
            const float c = (float)w * nmax * 0x1p-18f + (float)w * 0x1p-149f;
            if (counted) {
                const unsigned wq = (unsigned)w * 0x01010101u;
                auto one = [&](auto j_, unsigned &h, unsigned &m) {
                    constexpr int j = decltype(j_)::value;
                    const unsigned left = wq - Nb[j / 4];  // unflagged samples, per byte
                    const float rem = (float)((left >> (8 * (j % 4))) & 0xffu);
                    const float T = rem * thrf;
                    const float hi = __builtin_fmaf(T, 1.0f + 0x1p-18f, c);
                    float lo = __builtin_fmaf(T, 1.0f - 0x1p-18f, -c);
                    lo = (rem == 0.0f) ? __builtin_inff() : lo;  // all flagged: the sum IS the limit
                    h = 2 * h + (U[j] > hi);
                    m = 2 * m + (U[j] > lo);
                };
                ksp_static_for<32>([&](auto i_) { one(std::integral_constant<int, 31 - decltype(i_)::value>{}, h_lo, m_lo); });
                ksp_static_for<32>([&](auto i_) { one(std::integral_constant<int, 63 - decltype(i_)::value>{}, h_hi, m_hi); });
            } else {
                const float hi = __builtin_fmaf(limf, 1.0f + 0x1p-18f, c);
                const float lo = __builtin_fmaf(limf, 1.0f - 0x1p-18f, -c);
#pragma unroll
                for (int j = 31; j >= 0; j--) {
                    h_lo = 2 * h_lo + (U[j] > hi);
                    m_lo = 2 * m_lo + (U[j] > lo);
                }
#pragma unroll
                for (int j = 63; j >= 32; j--) {
                    h_hi = 2 * h_hi + (U[j] > hi);
                    m_hi = 2 * m_hi + (U[j] > lo);
                }
            }
            hits = (((unsigned long long)h_hi << 32) | h_lo) & valid;
            unsure = (((unsigned long long)m_hi << 32) | m_lo) & valid & ~hits;
        }
        // the undecided windows, as the host sums them
        if (ksp_any(unsure != 0)) {
            const unsigned long long f1 = __shfl_down(fl, 1, 64), f2 = __shfl_down(fl, 2, 64);
            while (ksp_any(unsure != 0)) {
                const bool has = unsure != 0;
                const int j = has ? __ffsll((long long)unsure) - 1 : 0;
                unsure &= unsure - 1;
                double s = 0.0;
                for (int m = 0; m < w; m++) {
                    const int jj = j + m;
                    const unsigned long long fw = jj < 64 ? fl : jj < 128 ? f1 : f2;
                    const bool sub = (fw >> (jj & 63)) & 1;
                    const double x = exact_dev_call<WIDTH>(c0 + jj, fetch);
                    s += sub ? (double)thrf : x;
                }
                if (has && s > limit) hits |= 1ull << j;
            }
            lvl = -1;  // (U and N are not kept across this rarely taken path: registers)
        }
        if (!ksp_any(hits != 0)) continue;
        // dilation: a hit at j flags j .. j + w - 1, up to two lanes on
        auto suffix_or = [](unsigned long long t) {
            t |= t >> 1; t |= t >> 2; t |= t >> 4; t |= t >> 8; t |= t >> 16; t |= t >> 32;
            return t;
        };
        unsigned long long add = hits;
        for (int s = 1; s < w && s < 64; s <<= 1) add |= add << s;
        unsigned long long h1 = __shfl_up(hits, 1, 64), h2 = __shfl_up(hits, 2, 64);
        if (lane < 1) h1 = 0;
        if (lane < 2) h2 = 0;
        if (w > 64) {
            add |= h1 ? ~0ull : 0ull;
            add |= suffix_or(h2 >> 1);
        } else if (w > 1) {
            add |= suffix_or(h1 >> (65 - w));
        }
        if (ksp_any((add & ~fl) != 0)) {
            fl |= add;
            lvl = -1;
        }
    }
    return fl;
}


// ---------------------------------------------------------------------------------
// Thresholds. Returns the flag mask of the lane's run (bit j: channel c0 + j).
//
// SumThreshold is evaluated on the float32 deviations with a rigorous error bound:
// |float32(d) - d| <= 2^-24 |d|, so a window sum computed from the rounded values is
// within 2^-24 * sum|d| (plus float64 rounding, covered by using 2^-23) of the exact
// float64 sum. Windows whose sum is further than that from the limit are decided as
// the exact arithmetic would decide them; the others (practically never) are summed
// again from exact deviations.
template <int R, int WIDTH, int MAXK = 4, class Fetch>
__device__ __forceinline__ unsigned long long threshold_flags(const FusedParams &p,
                                                              const float (&dev)[R], float dmax,
                                                              double noise64, int lane, int C,
                                                              Fetch &&fetch)
{
    static_assert(R <= 64, "flag mask is 64 bits");
    const int c0 = lane * R;
    unsigned long long fl = 0;
    if (p.threshold_kind == KSP_THRESHOLD_SIMPLE) {
        const double thr = p.n_sigma * noise64;  // float64 product (host.py:182)
        // dmax is the largest ROUNDED deviation: exact d <= dmax (1 + 2^-24)
        if (__any((double)dmax * (1.0 + 0x1p-23) > thr)) {
            // float32(d) > thr decides d > thr except when float32(d) is within one
            // rounding of thr; those samples are recomputed exactly
            unsigned long long unsure = 0;
#pragma unroll
            for (int j = 0; j < R; j++) {
                const double d = (double)dev[j];
                const double slack = fabs(d) * 0x1p-23 + 0x1p-149;  // (subnormals round by up to 2^-150)
                if (d - slack > thr)
                    fl |= 1ull << j;
                else if (d + slack > thr)
                    unsure |= 1ull << j;
            }
            while (__any(unsure != 0)) {
                const bool has = unsure != 0;
                const int j = has ? __ffsll((long long)unsure) - 1 : 0;
                unsure &= unsure - 1;
                const double d = exact_dev<WIDTH>(c0 + j, fetch);
                if (has && d > thr) fl |= 1ull << j;
            }
        }
        return fl;
    }
    const double t1 = p.n_sigma * noise64;  // host.py:252
    // (static indices only: a runtime-indexed array would live in scratch memory)
    static_assert(MAXK == 4 || (MAXK == 8 && R == 64), "windows above 8: lanes of 64 channels");
    constexpr int MAXW = 4;  // windows 1, 2, 4, 8 here; 16 .. 128 in threshold_wide
    float thr[MAXW];
    float thr_min = __builtin_inff();
    bool thr_nan = false;
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
        const float t = (float)(t1 * p.scales[k < KSP_MAX_WINDOWS ? k : 0]);  // host.py:235
        if (k < MAXW) thr[k < MAXW ? k : 0] = t;
        if (k < p.n_windows) {
            thr_min = fminf(thr_min, t);
            thr_nan |= (t != t);
        }
    }
    // Fast reject (exact): no window can fire unless some sample reaches
    // min_k thr_k; the 2^-20 margin makes the test conservative; needs thresholds > 0.
    const double cand = (double)thr_min * (1.0 - 0x1p-20);
    const bool positive = thr_min > 0.0f;
    const bool any = !positive || ((double)dmax >= cand);
    if (thr_nan || !__any(any)) return 0;

    // Which samples can make a window fire? The host compares the sequential float64 sum
    // of w values with w * thr_k (exact: w is a power of two), flagged samples standing
    // in as exactly thr_k. Rounding is monotone, so a window whose values are all
    // <= thr_k sums to <= w * thr_k and cannot fire: every firing window holds an
    // UNFLAGGED sample whose exact deviation exceeds thr_k >= thr_min, and (rounding to
    // float32 being monotone too) whose float32 deviation is >= thr_min. Two bit masks
    // per lane -- `gt0`: float32 deviation > thr_0, which decides window 1 outright
    // (d32 > thr_0 implies d > thr_0 for a float32 thr_0, d32 < thr_0 implies d < thr_0)
    // -- and `ge`: deviation >= thr_min. Strong interference is all in gt0; when no lane
    // has a sample in ge that is not in gt0, windows 2, 4, 8 have nothing to look at.
    unsigned long long gt0, ge;
    {
        unsigned g_lo = 0, g_hi = 0, e_lo = 0, e_hi = 0;
        const float t0 = thr[0], tm = thr_min;
        // mask = 2 * mask + (compare): v_cmp into vcc, v_addc folds it in
#pragma unroll
        for (int j = (R < 32 ? R : 32) - 1; j >= 0; j--)
            asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(e_lo) : "v"(dev[j]), "v"(tm) : "vcc");
#pragma unroll
        for (int j = R - 1; j >= 32; j--)
            asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(e_hi) : "v"(dev[j < R ? j : 0]), "v"(tm) : "vcc");
        // (dmax is the lane's largest deviation: no lane above thr_0 -- data without strong
        // interference -- means no sample above it, and gt0 stays empty)
        if (ksp_any(dmax > t0)) {
#pragma unroll
            for (int j = (R < 32 ? R : 32) - 1; j >= 0; j--)
                asm("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                    : "+v"(g_lo) : "v"(dev[j]), "v"(t0) : "vcc");
#pragma unroll
            for (int j = R - 1; j >= 32; j--)
                asm("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                    : "+v"(g_hi) : "v"(dev[j < R ? j : 0]), "v"(t0) : "vcc");
        }
        gt0 = ((unsigned long long)g_hi << 32) | g_lo;
        ge = ((unsigned long long)e_hi << 32) | e_lo;
    }
    // channels beyond the band hold deviation 0; with thresholds <= 0 everything is hot
    const unsigned long long inband =
        (c0 + R <= C) ? (R == 64 ? ~0ull : ((1ull << R) - 1))
                      : (c0 >= C ? 0ull : ((1ull << (C - c0)) - 1));
    if (positive && !__any(((ge & ~gt0) & inband) != 0)) return gt0 & inband;
    if constexpr (MAXK > 4) return threshold_wide<WIDTH>(p, dev, 0ull, ge, t1, positive, lane, C, fetch);

    // General case: some window may hold a weak sample. `hot` = unflagged samples that
    // can still make a window fire (all of them when a threshold is not positive).
    float d[R];  // working copy: deviations with flagged samples replaced by thr
#pragma unroll
    for (int j = 0; j < R; j++) d[j] = dev[j];
#pragma unroll
    for (int k = 0; k < MAXW; k++) {
        if (k >= p.n_windows) break;
        const int w = 1 << k;
        const float thrf = thr[k];
        const double limit = (double)__fmul_rn(thrf, (float)w);  // host.py:242
        if (ksp_any(fl != 0)) {  // (nothing flagged so far in this wavefront: nothing to replace)
#pragma unroll
            for (int j = 0; j < R; j++)
                if ((fl >> j) & 1) d[j] = thrf;  // host.py:237 (exact: thr is a float32)
        }
        // the next lanes' first 7 values and flag bits (w - 1 <= 7 are used)
        float ext[7];
#pragma unroll
        for (int m = 0; m < 7; m++) ext[m] = __shfl_down(d[m % R], 1 + m / R, 64);
        unsigned nfl = 0;  // flag bits of the 7 channels after this run
#pragma unroll
        for (int m = 0; m < 7; m++) {
            const unsigned long long f = __shfl_down(fl, 1 + m / R, 64);
            nfl |= (unsigned)((f >> (m % R)) & 1) << m;
        }
        // positions whose window [j, j + w) holds a hot sample, in any lane: only those
        // are summed (a wave-uniform mask, so the skipping is done by the scalar unit)
        unsigned long long need = ~0ull;
        if (positive && R == 64) {
            const unsigned long long hot = ge & ~fl;
            const unsigned long long hot_next = __shfl_down(hot, 1, 64);
            unsigned long long reach = hot;
#pragma unroll
            for (int m = 1; m < w; m++) reach |= (hot >> m) | (lane < 63 ? hot_next << (64 - m) : 0ull);
            need = ((unsigned long long)ksp_wave_or_dpp((unsigned)(reach >> 32)) << 32) |
                   ksp_wave_or_dpp((unsigned)reach);
        }
        unsigned long long hits = 0, unsure = 0;
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (!((need >> j) & 1)) continue;  // wave-uniform
            double s = 0.0, mag = 0.0;
#pragma unroll
            for (int m = 0; m < w; m++) {
                {
                    const int jj = j + m;
                    const float v = (jj < R) ? d[jj % R] : ext[(jj >= R) ? (jj - R) % 7 : 0];
                    const bool sub = (jj < R) ? ((fl >> (jj % 64)) & 1) : ((nfl >> ((jj >= R) ? (jj - R) % 7 : 0)) & 1);
                    s += (double)v;
                    mag += sub ? 0.0 : fabs((double)v);  // substituted values are exact
                }
            }
            const bool valid = (c0 + j + w <= C);
            const double slack = mag * 0x1p-23 + w * 0x1p-149;  // (subnormals round by up to 2^-150 each)
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
            for (int m = 0; m < w; m++) {
                const int jj = j + m;
                const bool sub = (jj < R) ? ((fl >> jj) & 1) : ((nfl >> (jj - R)) & 1);
                const double x = exact_dev<WIDTH>(c0 + jj, fetch);
                s += sub ? (double)thrf : x;
            }
            if (has && s > limit) hits |= 1ull << j;
        }
        // dilation: a hit at j flags j..j+w-1. `pin` holds the hits of the 7 positions
        // just below this lane's run (bit i <-> position i - 7).
        unsigned pin = 0;
        if (R >= 7) {
            const unsigned long long prev = __shfl_up(hits, 1, 64);
            if (lane > 0) pin = (unsigned)(prev >> (R >= 7 ? R - 7 : 0)) & 0x7fu;
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
