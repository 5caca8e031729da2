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
#include "flagger_fused_kernel.h"

// Events armed by ksp_flagger_fused_profile for the NEXT fused launch of this thread.
static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
// Kernels launched by this thread's last ksp_flagger_fused call (ksp_flagger_fused_last_path).
static thread_local int g_last_path = 0;

// widths other than 13 (flagger_fused_w*.hip): lanes always own 64 channels there
int ksp_fused_launch_w3_7(int width, int device, hipStream_t s, const FusedParams &p,
                          hipEvent_t ev0, hipEvent_t ev1);
int ksp_fused_launch_w9_11(int width, int device, hipStream_t s, const FusedParams &p,
                           hipEvent_t ev0, hipEvent_t ev1);
int ksp_fused_launch_w15_17(int width, int device, hipStream_t s, const FusedParams &p,
                            hipEvent_t ev0, hipEvent_t ev1);
int ksp_fused_launch_w19_21(int width, int device, hipStream_t s, const FusedParams &p,
                            hipEvent_t ev0, hipEvent_t ev1);
int ksp_fused_launch_w23_27(int width, int device, hipStream_t s, const FusedParams &p,
                            hipEvent_t ev0, hipEvent_t ev1);
int ksp_fused_launch_w29_31(int width, int device, hipStream_t s, const FusedParams &p,
                            hipEvent_t ev0, hipEvent_t ev1);

// more than 4096 channels (flagger_fused_long.hip)
int ksp_fused_long_supported(int channels, int width);
int ksp_fused_launch_long(int device, hipStream_t s, const FusedParams &p, hipEvent_t ev0,
                          hipEvent_t ev1);

// 4096 channels, whole strips of 8 baselines (flagger_ring.hip)
bool ksp_ring_supported(const FusedParams &p, int width);
int ksp_ring_launch(int width, int device, hipStream_t s, const FusedParams &p, int n_cu,
                    hipEvent_t ev0, hipEvent_t ev1);

static int ksp_fused_launch_other_width(int width, int device, hipStream_t s,
                                        const FusedParams &p, hipEvent_t ev0, hipEvent_t ev1)
{
    if (width <= 7) return ksp_fused_launch_w3_7(width, device, s, p, ev0, ev1);
    if (width <= 11) return ksp_fused_launch_w9_11(width, device, s, p, ev0, ev1);
    if (width <= 17) return ksp_fused_launch_w15_17(width, device, s, p, ev0, ev1);
    if (width <= 21) return ksp_fused_launch_w19_21(width, device, s, p, ev0, ev1);
    if (width <= 27) return ksp_fused_launch_w23_27(width, device, s, p, ev0, ev1);
    return ksp_fused_launch_w29_31(width, device, s, p, ev0, ev1);
}

extern "C" int ksp_flagger_fused_profile(void *start_event, void *stop_event)
{
    KSP_REQUIRE((start_event == nullptr) == (stop_event == nullptr), "need both events or none");
    g_prof_start = (hipEvent_t)start_event;
    g_prof_stop = (hipEvent_t)stop_event;
    return 0;
}

extern "C" int ksp_flagger_fused_last_path(void) { return g_last_path; }

// ring kernel: 0 = by size, 1 = whenever it applies, -1 = never; per thread, as the launch is
static thread_local int g_ring_mode = 2;  // (2: not yet taken from the environment)
static int ring_mode()
{
    if (g_ring_mode == 2) {
        const char *e = getenv("KSP_FUSED_RING");
        g_ring_mode = e == nullptr ? 0 : (e[0] == '1' ? 1 : -1);
    }
    return g_ring_mode;
}
extern "C" int ksp_flagger_fused_ring_mode(int mode)
{
    const int before = ring_mode();
    if (mode >= -1 && mode <= 1) g_ring_mode = mode;
    return before;
}

extern "C" int ksp_flagger_fused_supported(int channels, int width, int n_windows)
{
    if (n_windows < 1 || n_windows > KSP_MAX_WINDOWS) return 0;
    if (channels > 4096) return n_windows <= 4 && ksp_fused_long_supported(channels, width);
    return channels >= 1 && width >= 3 && width <= 31 && (width & 1);
}

extern "C" int ksp_flagger_fused(int device, void *stream, const void *vis,
                                 const uint8_t *in_flags, uint8_t *flags, float *deviations,
                                 float *noise, int channels, int baselines, int vis_stride,
                                 int in_flags_stride, int flags_stride, int dev_stride, int width,
                                 int is_amplitude, int flags_mode, int threshold_kind,
                                 double n_sigma, const double *scales64, int n_windows,
                                 int flag_value, void *workspace)
{
    // profiling events are consumed by this call whatever its outcome
    const hipEvent_t ev0 = g_prof_start, ev1 = g_prof_stop;
    g_prof_start = g_prof_stop = nullptr;
    g_last_path = 0;
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
    int n_cu = 0;
    p.n_dyn = 0;
    if (workspace != nullptr && p.n_strips >= 2048) p.n_dyn = (p.n_strips >> FUSED_DYN_SHIFT) & ~63;
    p.n_static = p.n_strips - p.n_dyn;
    p.dyn_blocks = p.n_dyn * FUSED_DYN_OVER / 4;
    {
        static std::atomic<int> cus[64];
        int n = (device >= 0 && device < 64) ? cus[device].load(std::memory_order_relaxed) : 0;
        if (n == 0) {
            KSP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
            if (device >= 0 && device < 64) cus[device].store(n, std::memory_order_relaxed);
        }
        p.first_round = 2 * n;
        n_cu = n;
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
    // The persistent ring kernel takes the whole strips of 8 baselines of a 4096-channel
    // launch without input flags; a ragged remainder (< 8 baselines) goes to the
    // 4-baseline kernel.
    // It pays from about 4 strips per workgroup on (measured, tools/time_ring_sizes.py: 0.089 ms
    // against 0.065 at 4096 baselines, 0.133 = 0.132 at 8192, 0.218 against 0.236 at 16384,
    // 0.384 against 0.431 at 32768); ksp_flagger_fused_ring_mode (tests, diagnostics; initial
    // value from KSP_FUSED_RING=1 / 0 in the environment) forces the choice.
    const int mode = ring_mode();
    const bool want_ring = mode != 0 ? mode > 0 : baselines / 8 >= 4 * n_cu;
    if (want_ring && width == 13 && ksp_ring_supported(p, width)) {
        // (Letting the ring kernel zero-fill `flags` itself -- write-through stores beside the
        // first strip's loads, a completion counter before the first flag byte -- was built
        // and measured: step time unchanged, 0.376 against 0.377 ms clean and 0.518 against
        // 0.519 with interference; the memset stays.)
        KSP_CHECK(hipMemsetAsync(p.flags, 0, (size_t)(p.channels - 1) * p.flags_stride + p.baselines, s));
        const int whole = baselines - baselines % 8;
        if (whole < baselines) {
            FusedParams t = p;
            t.vis = (const float2 *)p.vis + whole;
            t.flags = p.flags + whole;
            if (p.noise != nullptr) t.noise = p.noise + whole;
            t.baselines = baselines - whole;
            t.n_strips = ksp_divup(t.baselines, FUSED_STRIP);
            t.n_dyn = 0;
            t.n_static = t.n_strips;
            t.dyn_blocks = 0;
            const int rc = launch_fused<64, 13>(device, s, t, nullptr, nullptr, false);
            if (rc != 0) return rc;
            g_last_path |= 1;
        }
        g_last_path |= 4;
        return ksp_ring_launch(width, device, s, p, n_cu, ev0, ev1);
    }
    g_last_path = channels > 4096 ? 2 : 1;
    if (channels > 4096) return ksp_fused_launch_long(device, s, p, ev0, ev1);
    if (width != 13) return ksp_fused_launch_other_width(width, device, s, p, ev0, ev1);
    const bool wide = threshold_kind == KSP_THRESHOLD_SUM && n_windows > 4;  // (lanes of 64 channels)
    if (channels <= 64 * 4 && !wide) return launch_fused<4, 13>(device, s, p, ev0, ev1);
    if (channels <= 64 * 16 && !wide) return launch_fused<16, 13>(device, s, p, ev0, ev1);
    return launch_fused<64, 13>(device, s, p, ev0, ev1);
}
