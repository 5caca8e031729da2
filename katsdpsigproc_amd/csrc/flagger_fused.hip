// Fused single-pass RFI flagger for MI355X (gfx950).
//
// The reference runs five kernels (background -> transpose -> madnz_t ->
// threshold_sum -> transpose; reference rfi/device.py:1152-1164) and moves 31 bytes
// per sample through device memory. Here each visibility is read once (8 B) and each
// flag written once (1 B); everything in between stays on chip.
//
// Work decomposition ("strip" = 8 adjacent baselines x all channels):
//   * persistent 512-thread workgroups (8 wavefronts), one per CU, each walking over
//     strips; a strip is read as 64-byte row segments, turned into numpy's |z| and
//     parked as float32 in LDS, transposed to [baseline][channel] (146 KiB of 160 KiB);
//   * wavefront w then owns baseline w and lane l a run of R consecutive channels: the
//     sliding median (median_window.h), the MAD selection and SumThreshold are
//     wave-local (no workgroup barrier between them); cross-lane traffic goes through
//     shuffles, ballots and the wavefront's own LDS row;
//   * deviations are computed in float64 like the host path (reference
//     rfi/host.py:148-163, 235-245), stored rounded to float32 over the dead amplitudes
//     in LDS, and every decision that could depend on the rounding is re-taken from an
//     exact float64 recomputation (fused_common.h), so flags are bit-identical;
//   * software pipeline: the 16-byte loads of strip k+1 are issued into registers
//     before strip k is processed and consumed after it, so HBM latency is hidden
//     behind a whole strip of arithmetic although only one workgroup fits a CU.
//
// Roofline: HBM, 9 algorithmic bytes per sample (8 read + 1 written).
#include "fused_common.h"

template <int R, int WIDTH>
__global__ __launch_bounds__(FUSED_THREADS, 2) void flagger_fused_kernel(const FusedParams p)
{
    using LY = FusedLayout<R>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int C = p.channels;
    float *myrow = lds + wave * LY::ROW;
    double *list = (double *)(lds + LY::LDS_FLOATS) + wave * LY::LIST_STRIDE;

    int id = blockIdx.x;
    if (id >= p.n_strips) return;
    StripLoader<R> loader;
    loader.request(p, strip_of(id, p.n_strips) * FUSED_STRIP, tid);

    for (; id < p.n_strips; id += gridDim.x) {
        const int b0 = strip_of(id, p.n_strips) * FUSED_STRIP;
        const int bl = b0 + wave;
        // strip k: requests (issued one iteration ago) -> amplitudes in LDS
        loader.finish(p, lds, b0, tid);
        __syncthreads();
        // strip k+1: its requests stay in flight during everything below
        const int next = id + gridDim.x;
        if (next < p.n_strips) loader.request(p, strip_of(next, p.n_strips) * FUSED_STRIP, tid);
        if (p.debug_stop == 1) return;

        const double dmax = median_phase<R, WIDTH>(p, bl, myrow, lane, C);
        if (p.debug_stop == 2) {
            if (dmax == 12345.678 && p.noise) p.noise[0] = 1.f;  // keep the work alive
            return;
        }
        const double noise64 = mad_noise<R, WIDTH, LY::LIST_DOUBLES>(p, bl, myrow, lane, list);
        if (lane == 0 && p.noise != nullptr && bl < p.baselines) p.noise[bl] = (float)noise64;
        if (p.debug_stop == 3) return;

        if (p.deviations != nullptr) {
            // the rows hold the float32 deviations: write them as [channel][8 baselines]
            // before the threshold stage may use them as scratch
            __syncthreads();
            const int q = tid & 3, r0 = tid >> 2;
            const int blq = b0 + 2 * q;
            for (int row = r0; row < C; row += FUSED_THREADS / 4) {
                const int slot = LY::dev_slot(row);
                const float v0 = lds[(2 * q) * LY::ROW + slot];
                const float v1 = lds[(2 * q + 1) * LY::ROW + slot];
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

        const unsigned long long fl =
            threshold_flags<R, WIDTH>(p, bl, myrow, dmax, noise64, lane, C);
        if (p.debug_stop == 4) {
            if (fl == 0x123456789abcull && p.noise) p.noise[0] = 1.0f;
            return;
        }
        write_flags(p, fl, lane * R, bl, C);
        __syncthreads();  // every wavefront is done with strip k's LDS image
    }
}

// =================================================================================
template <int R, int WIDTH>
static int launch_fused(hipStream_t s, const FusedParams &p, int cus)
{
    using LY = FusedLayout<R>;
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
    // persistent grid: as many workgroups as fit the chip at once (LDS allows one per CU
    // for R = 64, more for the small-band variants)
    int per_cu = (int)((160 * 1024) / LY::LDS_BYTES);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    int grid = (cus > 0 ? cus : 256) * per_cu;
    const char *g = getenv("KSP_FUSED_GRID");  // diagnostic override
    if (g && atoi(g) > 0) grid = atoi(g);
    if (grid > p.n_strips) grid = p.n_strips;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FUSED_THREADS), LY::LDS_BYTES, s, p);
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

    static int cus[64] = {0};
    if (device >= 0 && device < 64 && cus[device] == 0) {
        hipDeviceProp_t prop;
        KSP_CHECK(hipGetDeviceProperties(&prop, device));
        cus[device] = prop.multiProcessorCount;
    }
    const int n_cu = (device >= 0 && device < 64) ? cus[device] : 256;

    hipStream_t s = (hipStream_t)stream;
    if (channels <= 64 * 4) return launch_fused<4, 13>(s, p, n_cu);
    if (channels <= 64 * 16) return launch_fused<16, 13>(s, p, n_cu);
    return launch_fused<64, 13>(s, p, n_cu);
}
