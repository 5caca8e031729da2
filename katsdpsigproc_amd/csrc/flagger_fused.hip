// Fused single-pass RFI flagger for MI355X (gfx950).
//
// The reference runs five kernels (background -> transpose -> madnz_t ->
// threshold_sum -> transpose; reference rfi/device.py:1152-1164) and moves 31 bytes
// per sample through device memory. Here each visibility is read once (8 B) and each
// flag written once (1 B); everything in between stays on chip.
//
// Work decomposition ("strip" = FUSED_STRIP (4) adjacent baselines x all channels):
//   * one 256-thread workgroup (4 wavefronts) works on one strip; its LDS image is
//     ~76 KiB, so two workgroups share a CU and one's loads overlap the other's
//     arithmetic. strip_of() hands neighbouring strips to the same XCD at the same
//     time so that the 32-byte row segments of a 128-byte line meet in that XCD's L2;
//   * the strip is read as row segments (8 rows in flight per lane, double buffered),
//     turned into numpy's |z| and parked as float32 in LDS, transposed to
//     [baseline][channel];
//   * from then on wavefront w owns baseline w and lane l a run of R consecutive
//     channels: the sliding median (median_window.h), the MAD selection and
//     SumThreshold are wave-local, cross-lane traffic goes through shuffles/ballots;
//   * deviations are kept as float32 in registers (rounding is monotone, so ordering
//     decisions can be filtered on them); every value that decides a result - the
//     MAD's median candidates, window sums within their error bound of a threshold -
//     is recomputed exactly in float64 from the LDS amplitudes, because the host path
//     is float64 after the amplitude (reference rfi/host.py:148-163, 235-245) and the
//     flags must match it bit for bit.
//   * flags are zero-filled by a memset node ahead of the kernel; the kernel only
//     writes the (sparse) non-zero bytes.
//
// Roofline: HBM, 9 algorithmic bytes per sample (8 read + 1 written).
#include "fused_common.h"
#include <hip/hip_ext.h>

#include <atomic>

#ifndef FUSED_DYN_SHIFT
#define FUSED_DYN_SHIFT 4  // the last 1 / 2^n of the strips are scheduled dynamically
#endif
#ifndef FUSED_STAGGER
#define FUSED_STAGGER 2  // x 8128 cycles: how long the second workgroup of a CU waits once
#endif

// Events armed by ksp_flagger_fused_profile for the NEXT fused launch of this thread.
static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

// =================================================================================
template <int R, int WIDTH>
__global__ __launch_bounds__(FUSED_THREADS, 2) void flagger_fused_kernel(const FusedParams p)
{
    using LY = FusedLayout<R>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int C = p.channels;
    // Workgroups are dealt to the 8 XCDs round robin, blockIdx % 8, an equal share each --
    // but the XCDs do not run equally fast (10 % between the fastest and the slowest in
    // a traced launch), so with a static strip per workgroup the fast ones idle at the
    // end. The last 1/16 of the strips are therefore handed out from a counter to twice
    // as many workgroups as there are strips: an XCD that gets through its static share
    // early takes more of them, a late one finds the counter exhausted and its surplus
    // workgroups leave at once.
    int strip;
    const bool dynamic = (int)blockIdx.x >= p.n_static;
    if (!dynamic) {
        strip = strip_of(blockIdx.x, p.n_static);
    } else {
        int *slot = (int *)lds;
        if (tid == 0) *slot = (int)atomicAdd(&p.work[0], 1u);
        __syncthreads();
        const int t = *slot;
        __syncthreads();  // everybody has read the slot before the loader reuses it
        strip = p.n_static + t;
        if (t >= p.n_dyn) {
            // nothing left: leave (the last dynamic workgroup to leave resets the counters)
            if (tid == 0 && atomicAdd(&p.work[1], 1u) == (unsigned)p.dyn_blocks - 1u) {
                p.work[0] = 0;
                p.work[1] = 0;
            }
            return;
        }
    }
    const int b0 = strip * FUSED_STRIP;

    // diagnostic time stamps (shader clock) of this wavefront's phases
    unsigned long long *trace = FUSED_DIAG_TRACE(p) ? FUSED_DIAG_TRACE(p) + ((size_t)blockIdx.x * FUSED_STRIP + wave) * 16 : nullptr;
    auto stamp = [&](int i) {
        if (trace != nullptr && lane == 0) trace[i] = FUSED_DIAG_CLOCK();
    };
    if (trace != nullptr && lane == 0)
        trace[7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    stamp(0);
#if FUSED_STAGGER > 0
    // Two workgroups share a CU and start together, and they then STAY in step -- both
    // loading (sharing the memory path), then both computing (sharing the vector ALUs)
    // -- because each slows the other equally. Holding one of the pair back once, by
    // about one LOAD phase, puts them in anti-phase for the rest of the launch (every
    // later workgroup inherits the slot, and with it the phase, of the one it
    // replaces): one's loads then run under the other's arithmetic. Which of the two
    // waits is told by the hardware's workgroup slot number on the CU. Speed only.
    if ((int)blockIdx.x < p.first_round &&
        ((__builtin_amdgcn_s_getreg(63492 /* HW_REG_HW_ID */) >> 16) & 1)) {
#pragma unroll
        for (int i = 0; i < FUSED_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    bool masked = true;  // may the strip hold samples that take no part (NaN in LDS)?
    if (!p.is_amplitude && b0 + FUSED_STRIP <= p.baselines) {
        if (p.flags_mode == KSP_FLAGS_NONE)
            masked = load_strip_fast<R, KSP_FLAGS_NONE>(p, lds, b0, tid);
        else if (p.flags_mode == KSP_FLAGS_CHANNEL)
            masked = load_strip_fast<R, KSP_FLAGS_CHANNEL>(p, lds, b0, tid);
        else
            masked = load_strip_fast<R, KSP_FLAGS_FULL>(p, lds, b0, tid);
    } else {
        load_strip<R>(p, lds, b0, tid);
    }
    stamp(1);
    const bool any_masked = __syncthreads_or(masked);
    stamp(2);
    if (FUSED_DIAG_STOP(p) == 1 || FUSED_DIAG_STOP(p) == 11) return;

    const int bl = b0 + wave;
    float *myrow = lds + wave * LY::ROW;
    double *list = (double *)(lds + LY::LDS_FLOATS) + wave * LY::LIST_DOUBLES;
    // amplitude of any channel of this baseline, for exact recomputation (LDS copy)
    auto fetch = [&](int c) -> float {
        return (c >= 0 && c < C) ? myrow[LY::index(c)] : __builtin_nanf("");
    };
    float dev[R];
    float dmax;
    bool merged = false;
    if constexpr (R == 64) {
        // clean strip over the whole band: the merging median (median_merge.h)
        if (!any_masked && C == 64 * R && FUSED_DIAG_STOP(p) != 21) {
            MergeMedian<R, WIDTH> mm;
            mm.template run_lane<LY::RUN - R>(myrow + lane * LY::RUN, lane, dev, dmax);
            merged = true;
        }
    }
    if (!merged) median_phase<R, WIDTH>(myrow, lane, dev, dmax);
    stamp(3);
    const FusedParams &pa = p;
    if (FUSED_DIAG_STOP(pa) == 2) {
        float acc = dmax;
#pragma unroll
        for (int j = 0; j < R; j++) acc += dev[j];
        if (acc == 12345.678f && pa.noise) pa.noise[0] = acc;  // keep the work alive
        return;
    }

    const double noise64 = mad_noise<R, WIDTH, LY::LIST_DOUBLES>(dev, lane, list, fetch, FUSED_DIAG_STOP(pa), trace);
    if (lane == 0 && pa.noise != nullptr && bl < pa.baselines) pa.noise[bl] = (float)noise64;
    stamp(4);
    if (FUSED_DIAG_STOP(pa) == 3 || FUSED_DIAG_STOP(pa) > 30) return;

    const unsigned long long fl =
        threshold_flags<R, WIDTH>(pa, dev, dmax, noise64, lane, C, fetch);
    stamp(5);
    if (FUSED_DIAG_STOP(pa) == 4) {
        if (fl == 0x123456789abcull && pa.noise) pa.noise[0] = 1.0f;
        return;
    }

    if (pa.deviations != nullptr) {
        // stage float32 deviations in this wavefront's LDS row (the amplitudes are no
        // longer needed), then write them as [channel][8 baselines]
#pragma unroll
        for (int j = 0; j < R; j++) myrow[lane * LY::RUN + j] = dev[j];
        __syncthreads();
        constexpr int LPR = FUSED_STRIP / 2;
        const int q = tid % LPR, r0 = tid / LPR;
        const int blq = b0 + 2 * q;
        for (int row = r0; row < C; row += FUSED_THREADS / LPR) {
            const int idx = LY::index(row);
            const float v0 = lds[(2 * q) * LY::ROW + idx];
            const float v1 = lds[(2 * q + 1) * LY::ROW + idx];
            float *dst = pa.deviations + (size_t)row * pa.dev_stride + blq;
            if (blq + 1 < pa.baselines && (pa.dev_stride & 1) == 0)
                *(float2 *)dst = make_float2(v0, v1);
            else {
                if (blq < pa.baselines) dst[0] = v0;
                if (blq + 1 < pa.baselines) dst[1] = v1;
            }
        }
    }
    write_flags(pa, fl, lane * R, bl, C);
    stamp(6);
    if (dynamic && tid == 0 && atomicAdd(&pa.work[1], 1u) == (unsigned)pa.dyn_blocks - 1u) {
        pa.work[0] = 0;
        pa.work[1] = 0;
    }
}

// =================================================================================
template <int R, int WIDTH>
static int launch_fused(int device, hipStream_t s, const FusedParams &p)
{
    using LY = FusedLayout<R>;
    const size_t lds_bytes = LY::LDS_BYTES;
    // all flags start at zero; the kernels only write the (rare) non-zero ones
    KSP_CHECK(hipMemsetAsync(p.flags, 0, (size_t)(p.channels - 1) * p.flags_stride + p.baselines,
                             s));
    auto kern = flagger_fused_kernel<R, WIDTH>;
    // the opt-in to more than 64 KiB of dynamic LDS is per device (one context per
    // device in one process is a supported arrangement, reference doc/user/init.rst:4-6)
    static std::atomic<bool> attr_set[64];
    if (device < 0 || device >= 64 || !attr_set[device].load(std::memory_order_acquire)) {
        KSP_CHECK(hipFuncSetAttribute((const void *)kern,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        if (device >= 0 && device < 64) attr_set[device].store(true, std::memory_order_release);
#ifdef KSP_DIAG
        if (getenv("KSP_FUSED_DEBUG_OCC")) {
            int nb = -1;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, FUSED_THREADS, lds_bytes);
            fprintf(stderr, "flagger_fused_kernel<%d>: %d workgroups/CU, LDS %zu B\n", R, nb, lds_bytes);
        }
#endif
    }
#ifdef KSP_DIAG
    const char *trace_path = getenv("KSP_FUSED_DEBUG_TRACE");
    if (trace_path != nullptr) {
        // diagnostic run: collect per-wavefront phase time stamps and dump them
        g_prof_start = g_prof_stop = nullptr;
        FusedParams pt = p;
        const size_t n = (size_t)(p.n_static + p.dyn_blocks) * FUSED_STRIP * 16;
        KSP_CHECK(hipMalloc(&pt.trace, n * 8));
        KSP_CHECK(hipMemsetAsync(pt.trace, 0, n * 8, s));
        hipLaunchKernelGGL(kern, dim3(p.n_static + p.dyn_blocks), dim3(FUSED_THREADS), lds_bytes, s, pt);
        KSP_LAUNCH_CHECK();
        KSP_CHECK(hipStreamSynchronize(s));
        unsigned long long *host = (unsigned long long *)malloc(n * 8);
        KSP_CHECK(hipMemcpy(host, pt.trace, n * 8, hipMemcpyDeviceToHost));
        KSP_CHECK(hipFree(pt.trace));
        FILE *f = fopen(trace_path, "wb");
        if (f != nullptr) {
            fwrite(host, 8, n, f);
            fclose(f);
        }
        free(host);
        return 0;
    }
#endif
    // events armed by ksp_flagger_fused_profile time exactly this kernel (not the
    // zero-fill before it); they are consumed by this launch whatever its outcome
    const hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;
    g_prof_start = g_prof_stop = nullptr;
    if (ev0 != nullptr)
        hipExtLaunchKernelGGL(kern, dim3(p.n_static + p.dyn_blocks), dim3(FUSED_THREADS),
                              lds_bytes, s, ev0, ev1, 0, p);
    else
        hipLaunchKernelGGL(kern, dim3(p.n_static + p.dyn_blocks), dim3(FUSED_THREADS), lds_bytes,
                           s, p);
    KSP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ksp_flagger_fused_profile(void *start_event, void *stop_event)
{
    KSP_REQUIRE((start_event == nullptr) == (stop_event == nullptr), "need both events or none");
    g_prof_start = (hipEvent_t)start_event;
    g_prof_stop = (hipEvent_t)stop_event;
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
                                 int flag_value, void *workspace)
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
    p.work = (unsigned *)workspace;
    p.n_dyn = 0;
    if (workspace != nullptr && p.n_strips >= 2048) p.n_dyn = (p.n_strips >> FUSED_DYN_SHIFT) & ~63;
    p.n_static = p.n_strips - p.n_dyn;
    p.dyn_blocks = 2 * p.n_dyn;
    {
        static std::atomic<int> cus[64];
        int n = (device >= 0 && device < 64) ? cus[device].load(std::memory_order_relaxed) : 0;
        if (n == 0) {
            KSP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
            if (device >= 0 && device < 64) cus[device].store(n, std::memory_order_relaxed);
        }
        p.first_round = 2 * n;
    }
#ifdef KSP_DIAG
    p.trace = nullptr;
    {
        const char *dbg = getenv("KSP_FUSED_DEBUG_STOP");
        p.debug_stop = dbg ? atoi(dbg) : 0;
    }
#endif
    p.n_sigma = n_sigma;
    for (int k = 0; k < KSP_MAX_WINDOWS; k++)
        p.scales[k] = (scales64 != nullptr && k < n_windows) ? scales64[k] : 0.0;

    hipStream_t s = (hipStream_t)stream;
    if (channels <= 64 * 4) return launch_fused<4, 13>(device, s, p);
    if (channels <= 64 * 16) return launch_fused<16, 13>(device, s, p);
    return launch_fused<64, 13>(device, s, p);
}
