// Fused flagger for 4097 .. 12288 channels: see fused_long.h.
#include <hip/hip_ext.h>

#include <atomic>

#include "fused_long.h"

template <int NR, int S, int WIDTH>
__global__ __launch_bounds__(64 * S, 1) void flagger_long_kernel(const FusedParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int C = p.channels;
    const int runs = (C + 63) >> 6;
    const int row_floats = runs * LONG_RUN + 8;
    // strips that share a 256-byte stretch of every row go to one XCD (as strip_of does
    // for the 4096-channel kernel; 8 strips of 4 baselines, or of 3: then 96-byte groups)
    const int b0 = strip_of(blockIdx.x, gridDim.x) * S;

    bool masked;
    if constexpr (S == 4) {
      if (!p.is_amplitude && b0 + S <= p.baselines) {
        if (p.flags_mode == KSP_FLAGS_NONE)
            masked = load_strip_long_pairs<KSP_FLAGS_NONE>(p, lds, row_floats, runs, b0, tid);
        else if (p.flags_mode == KSP_FLAGS_CHANNEL)
            masked = load_strip_long_pairs<KSP_FLAGS_CHANNEL>(p, lds, row_floats, runs, b0, tid);
        else
            masked = load_strip_long_pairs<KSP_FLAGS_FULL>(p, lds, row_floats, runs, b0, tid);
      } else {
        masked = load_strip_long<S>(p, lds, row_floats, runs, b0, tid);
      }
    } else {
        masked = load_strip_long<S>(p, lds, row_floats, runs, b0, tid);
    }
    const bool any_masked = __syncthreads_or(masked);

    const int bl = b0 + wave;
    if (bl >= p.baselines) return;  // ragged last strip: whole wavefronts leave together
    const float *myrow = lds + wave * row_floats;
    double *list = (double *)(lds + S * row_floats) + wave * 256;
    auto fetch = [&](int c) -> float {
        return (c >= 0 && c < C) ? myrow[long_index(c)] : __builtin_nanf("");
    };
    float dev[NR][64];
    float dmax = -__builtin_inff();
    int tiny = 0;  // deviations of +-2^-150 (SortedWindow::tiny)
    const bool merged = !any_masked && (C & 63) == 0 && WIDTH <= 13;
#pragma unroll
    for (int g = 0; g < NR; g++) {
        const int grun = g * 64 + lane;
        float dm = -__builtin_inff();
        bool done = false;
        if constexpr (WIDTH <= 13) {
            if (merged) {
                // clean strip, whole runs: the merging median on the lanes that own a run
                if (grun < runs) {
                    const float *run = myrow + grun * LONG_RUN;
                    MergeMedian<64, WIDTH> mm;
                    mm.run_src([&](int i) { return run[i]; },
                               [&](int i) { return run[i - (LONG_RUN - 64)]; },
                               [&](int i) { return run[i + (LONG_RUN - 64)]; }, grun == 0,
                               grun == runs - 1, dev[g], dm);
                    tiny += mm.tiny;
                } else {
#pragma unroll
                    for (int j = 0; j < 64; j++) dev[g][j] = 0.0f;
                }
                done = true;
            }
        }
        if (!done) {
            const int c0 = grun * 64;
            auto amp_rel = [&](int i) -> float {
                const int c = c0 + i;
                return (c >= 0 && c < 64 * runs) ? myrow[long_index(c)] : __builtin_nanf("");
            };
            median_phase_src<64, WIDTH>(amp_rel, dev[g], dm, &tiny);
        }
        dmax = fmaxf(dmax, dm);
    }
    const double noise64 = mad_noise_long<NR, WIDTH, 256>(dev, lane, list, fetch, tiny);
    if (lane == 0 && p.noise != nullptr) p.noise[bl] = (float)noise64;
    if (p.deviations != nullptr) {
#pragma unroll
        for (int g = 0; g < NR; g++) {
            const int c0 = (g * 64 + lane) << 6;
#pragma unroll
            for (int j = 0; j < 64; j++)
                if (c0 + j < C) p.deviations[(size_t)(c0 + j) * p.dev_stride + bl] = dev[g][j];
        }
    }
    unsigned long long fl[NR];
    threshold_flags_long<NR, WIDTH>(p, dev, dmax, noise64, lane, C, fetch, fl);  // clobbers dev
    const uint8_t fv = (uint8_t)p.flag_value;
#pragma unroll
    for (int g = 0; g < NR; g++) {
        const int c0 = (g * 64 + lane) << 6;
        unsigned long long f = fl[g];
        while (f) {
            const int j = __ffsll((long long)f) - 1;
            f &= f - 1;
            p.flags[(size_t)(c0 + j) * p.flags_stride + bl] = fv;
        }
    }
}

template <int NR, int S, int WIDTH>
static int launch_long(int device, hipStream_t s, const FusedParams &p, hipEvent_t ev0,
                       hipEvent_t ev1)
{
    const int runs = (p.channels + 63) >> 6;
    const size_t lds_bytes = sizeof(float) * S * (runs * LONG_RUN + 8) + sizeof(double) * 256 * S;
    KSP_CHECK(hipMemsetAsync(p.flags, 0, (size_t)(p.channels - 1) * p.flags_stride + p.baselines, s));
    auto kern = flagger_long_kernel<NR, S, WIDTH>;
    // opt in once per device to the whole 160 KiB (the size in use depends on the channel count)
    static std::atomic<bool> attr_set[64];
    if (device < 0 || device >= 64 || !attr_set[device].load(std::memory_order_acquire)) {
        hipFuncAttributes fa;
        KSP_CHECK(hipFuncGetAttributes(&fa, (const void *)kern));
        KSP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024 - (int)fa.sharedSizeBytes));
        if (device >= 0 && device < 64) attr_set[device].store(true, std::memory_order_release);
    }
    if (ev0 != nullptr)  // events of ksp_flagger_fused_profile: around the kernel itself
        hipExtLaunchKernelGGL(kern, dim3(ksp_divup(p.baselines, S)), dim3(64 * S), lds_bytes, s,
                              ev0, ev1, 0, p);
    else
        hipLaunchKernelGGL(kern, dim3(ksp_divup(p.baselines, S)), dim3(64 * S), lds_bytes, s, p);
    KSP_LAUNCH_CHECK();
    return 0;
}

// Largest channel count each strip height can hold in 160 KiB of LDS.
static int long_fits(int channels, int strip)
{
    const int runs = (channels + 63) >> 6;
    return sizeof(float) * strip * (runs * LONG_RUN + 8) + sizeof(double) * 256 * strip <= 160 * 1024;
}

#ifdef KSP_LONG_THREE_GROUPS
int ksp_fused_launch_long3(int device, hipStream_t s, const FusedParams &p, hipEvent_t ev0,
                           hipEvent_t ev1)
{
    if (long_fits(p.channels, 4)) return launch_long<3, 4, 13>(device, s, p, ev0, ev1);
    return launch_long<3, 3, 13>(device, s, p, ev0, ev1);
}
#else
int ksp_fused_launch_long3(int device, hipStream_t s, const FusedParams &p, hipEvent_t ev0,
                           hipEvent_t ev1);

int ksp_fused_long_supported(int channels, int width)
{
    return width == 13 && channels > 4096 && channels <= 12288 && long_fits(channels, 3);
}

int ksp_fused_launch_long(int device, hipStream_t s, const FusedParams &p, hipEvent_t ev0,
                          hipEvent_t ev1)
{
    if (p.channels <= 8192) return launch_long<2, 4, 13>(device, s, p, ev0, ev1);  // 4 x 8192 always fit
    return ksp_fused_launch_long3(device, s, p, ev0, ev1);
}
#endif
