// Fused single-pass RFI flagger for MI355X (gfx950).
//
// The reference runs five kernels (background -> transpose -> madnz_t ->
// threshold_sum -> transpose; reference rfi/device.py:1152-1164) and moves 31 bytes
// per sample through device memory. Here each visibility is read once (8 B) and each
// flag written once (1 B); everything in between stays on chip.
//
// Work decomposition ("strip" = 8 adjacent baselines x all channels):
//   * one 512-thread workgroup (8 wavefronts) per CU works on one strip at a time;
//   * the strip is read as 64-byte row segments, turned into numpy's |z| and parked
//     as float32 in LDS, transposed to [baseline][channel] (136 KiB of 160 KiB);
//   * from then on wavefront w owns baseline w and lane l a run of R consecutive
//     channels: the sliding median (median_window.h), the MAD selection and
//     SumThreshold are wave-local, cross-lane traffic goes through shuffles/ballots;
//   * deviations live in float64 registers: the host path is float64 after the
//     amplitude (reference rfi/host.py:148-163, 235-245) and flags must match it.
//
// Two kernels share those phases (fused_common.h):
//   flagger_fused_kernel  one strip per workgroup; every option (input flags, ragged
//                         shapes, amplitude input, deviations output).
//   flagger_pipe_kernel   persistent and software-pipelined: while a wavefront slides
//                         its median over strip k, every step also requests one sample
//                         of strip k+1 per lane, and turns the sample requested 16 steps
//                         earlier into an amplitude that overwrites an LDS slot strip k no
//                         longer needs. HBM latency, the amplitude arithmetic and the
//                         median therefore overlap instead of queueing behind a barrier.
//
// Roofline: HBM, 9 algorithmic bytes per sample (8 read + 1 written).
#include "fused_common.h"

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
    const int b0 = strip_of(blockIdx.x, p.n_strips) * FUSED_STRIP;

    load_strip<R>(p, lds, b0, tid);
    __syncthreads();
    if (p.debug_stop == 1) return;

    const int bl = b0 + wave;
    float *myrow = lds + wave * LY::ROW;
    double *list = (double *)(lds + LY::LDS_FLOATS) + wave * LY::LIST_DOUBLES;
    // amplitude of any channel of this baseline, for exact recomputation (LDS copy)
    auto fetch = [&](int c) -> float {
        return (c >= 0 && c < C) ? myrow[LY::index(c)] : __builtin_nanf("");
    };
    float dev[R];
    double dmax;
    median_phase<R, WIDTH>(myrow, lane, C, dev, dmax, [](int) {});
    if (p.debug_stop == 2) {
        float acc = (float)dmax;
#pragma unroll
        for (int j = 0; j < R; j++) acc += dev[j];
        if (acc == 12345.678f && p.noise) p.noise[0] = acc;  // keep the work alive
        return;
    }

    const double noise64 = mad_noise<R, WIDTH, LY::LIST_DOUBLES>(dev, lane, list, fetch);
    if (lane == 0 && p.noise != nullptr && bl < p.baselines) p.noise[bl] = (float)noise64;
    if (p.debug_stop == 3) return;

    const unsigned long long fl =
        threshold_flags<R, WIDTH>(p, dev, dmax, noise64, lane, C, fetch);
    if (p.debug_stop == 4) {
        if (fl == 0x123456789abcull && p.noise) p.noise[0] = 1.0f;
        return;
    }

    if (p.deviations != nullptr) {
        // stage float32 deviations in this wavefront's LDS row (the amplitudes are no
        // longer needed), then write them as [channel][8 baselines]
#pragma unroll
        for (int j = 0; j < R; j++) myrow[lane * LY::RUN + j] = dev[j];
        __syncthreads();
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
    }
    write_flags(p, fl, lane * R, bl, C);
}

// =================================================================================
// Persistent, software-pipelined variant. Requirements (checked by the launcher):
// complex64 input, channels == 64 * R, baselines a multiple of 8, no deviations output.
//
// Prefetch schedule for the NEXT strip, per lane and per median step j (0..R-1):
//   "op" i (0..R-1) handles channel offset q_i = (i + 6) % R inside every lane run,
//   i.e. row l' * R + q_i for l' = tid / 8 and baseline tid % 8 (8 lanes = one 64-byte
//   row segment). Op i is issued (global load) at step max(0, i + 9 - D) and consumed
//   (amplitude -> LDS) at step i + 9; ops that would be consumed after the last step
//   run in a tail behind a barrier. Slot q = i + 6 was last read (as the sample
//   entering a window) at step q - 6 = i, so with a workgroup barrier every 4 steps no
//   wavefront can still need it at step i + 9; offsets 0..5 are read late (by the lane
//   below, steps R-6..R-1) and are therefore the last ops, in the tail.
template <int R, int WIDTH>
__global__ __launch_bounds__(FUSED_THREADS, 2) void flagger_pipe_kernel(const FusedParams p)
{
    using LY = FusedLayout<R>;
    constexpr int D = 16;     // loads in flight per lane
    constexpr int LAG = 9;    // steps between the last use of a slot and its overwrite
    constexpr int LOOP_OPS = R - LAG;  // ops consumed inside the median loop
    static_assert(R >= 32 && R % 4 == 0, "schedule assumes runs of at least 32 channels");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int C = p.channels;

    float *myrow = lds + wave * LY::ROW;
    double *list = (double *)(lds + LY::LDS_FLOATS) + wave * LY::LIST_DOUBLES;
    // prefetch role of this lane: row l' * R + q of baseline pb within the strip
    const int pl = tid >> 3;  // l' (0..63)
    const int pb = tid & 7;
    float *pslot = lds + pb * LY::ROW + pl * LY::RUN;

    int id = blockIdx.x;
    if (id >= p.n_strips) return;
    load_strip<R>(p, lds, strip_of(id, p.n_strips) * FUSED_STRIP, tid);
    __syncthreads();

    for (; id < p.n_strips; id += gridDim.x) {
        const int b0 = strip_of(id, p.n_strips) * FUSED_STRIP;
        const int bl = b0 + wave;
        const int next = id + gridDim.x;
        const bool has_next = next < p.n_strips;  // workgroup-uniform
        const int nb = (has_next ? strip_of(next, p.n_strips) : 0) * FUSED_STRIP + pb;
        const float2 *nsrc = (const float2 *)p.vis + (size_t)pl * R * p.vis_stride + nb;

        float2 pf[D];
        auto issue = [&](int i) {  // request op i
            const int q = (i + 6) % R;
            pf[i % D] = nsrc[(size_t)q * p.vis_stride];
        };
        auto consume = [&](int i) {  // op i: amplitude -> LDS slot (dead in strip k)
            const int q = (i + 6) % R;
            const float2 v = pf[i % D];
            pslot[q] = amp_with_flags(p, v.x, v.y, pl * R + q, nb);
        };

        // amplitude of any channel of this baseline, for exact recomputation: the LDS
        // copy is overwritten by the next strip, so go back to memory (a few dozen
        // 8-byte reads per baseline, served by L2 / Infinity Cache)
        auto fetch = [&](int c) -> float {
            if (c < 0 || c >= C) return __builtin_nanf("");
            const float2 v = ((const float2 *)p.vis)[(size_t)c * p.vis_stride + bl];
            return amp_with_flags(p, v.x, v.y, c, bl);
        };
        float dev[R];
        double dmax;
        median_phase<R, WIDTH>(myrow, lane, C, dev, dmax, [&](int j) {
            if (has_next) {
                if (j >= LAG && j - LAG < LOOP_OPS) consume(j - LAG);
                if (j == 0) {
#pragma unroll
                    for (int i = 0; i <= D - LAG; i++) issue(i);
                } else if (j + D - LAG < R)
                    issue(j + D - LAG);
            }
            if ((j & 3) == 3) __syncthreads();  // keeps the wavefronts within 4 steps
        });
        if (has_next) {
            __syncthreads();  // every wavefront has finished reading strip k
#pragma unroll
            for (int i = LOOP_OPS; i < R; i++) consume(i);
        }

        const double noise64 = mad_noise<R, WIDTH, LY::LIST_DOUBLES>(dev, lane, list, fetch);
        if (lane == 0 && p.noise != nullptr) p.noise[bl] = (float)noise64;
        const unsigned long long fl =
            threshold_flags<R, WIDTH>(p, dev, dmax, noise64, lane, C, fetch);
        write_flags(p, fl, lane * R, bl, C);
        __syncthreads();  // strip k+1 is complete in LDS
    }
}

// =================================================================================
template <int R, int WIDTH>
static int launch_fused(hipStream_t s, const FusedParams &p, int cus)
{
    using LY = FusedLayout<R>;
    const size_t lds_bytes = LY::LDS_BYTES;
    // all flags start at zero; the kernels only write the (rare) non-zero ones
    KSP_CHECK(hipMemsetAsync(p.flags, 0, (size_t)(p.channels - 1) * p.flags_stride + p.baselines,
                             s));
    // the pipelined kernel is opt-in (KSP_FUSED_PIPELINE=1) until it beats the plain one
    const char *pipe = getenv("KSP_FUSED_PIPELINE");
    const bool pipelined = R >= 32 && p.channels == 64 * R && !p.is_amplitude &&
                           p.deviations == nullptr && (p.baselines % FUSED_STRIP) == 0 &&
                           p.debug_stop == 0 && (pipe && pipe[0] == '1');
    if (pipelined) {
        if constexpr (R >= 32) {
            auto kern = flagger_pipe_kernel<R, WIDTH>;
            static bool attr_set = false;
            if (!attr_set) {
                KSP_CHECK(hipFuncSetAttribute((const void *)kern,
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              160 * 1024));
                attr_set = true;
            }
            // one workgroup per CU (LDS-limited)
            int grid = cus > 0 ? cus : 256;
            if (grid > p.n_strips) grid = p.n_strips;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(FUSED_THREADS), lds_bytes, s, p);
        }
    } else {
        auto kern = flagger_fused_kernel<R, WIDTH>;
        static bool attr_set = false;
        if (!attr_set) {
            KSP_CHECK(hipFuncSetAttribute((const void *)kern,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3(p.n_strips), dim3(FUSED_THREADS), lds_bytes, s, p);
    }
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
