// Noise estimate: 1.4826 * median(|x| : x != 0) per baseline.
//
// madnz_t (baseline-major input, reference rfi/madnz_t.mako:72-87): one 256-thread
// workgroup owns one baseline; the row is read once, fully coalesced, into
// registers and the median is found by the bit-wise rank search of rank.h.
//
// madnz (channel-major input, reference rfi/madnz.mako:105-123): lane <-> baseline
// keeps the reads coalesced; like the reference kernel it re-reads the column for
// every search pass (the reference itself recommends the transposed variant,
// rfi/device.py:366), but the 16 channel phases of a workgroup make each pass a
// set of contiguous 256-byte row reads that hit in L2 after the first pass.
//
// Numerics (reference rfi/host.py:157-163 on float32 input): numpy.median stays in
// float32 -- even count -> float32(a + b) * 0.5 -- and the 1.4826 scale is applied
// in float64; the float32 output is that float64 product rounded once.
#include "rank.h"

#define KSP_MAD_NORMAL 1.4826

template <int VT>
__global__ __launch_bounds__(KSP_RANK_THREADS) void madnz_t_kernel(const float *__restrict__ in,
                                                                    float *__restrict__ noise,
                                                                    int channels, int stride)
{
    __shared__ RankScratch scratch;
    const int bl = blockIdx.x;
    const int t = threadIdx.x;
    const float *row = in + (size_t)bl * stride;
    float v[VT];
    int zeros = 0;
#pragma unroll
    for (int i = 0; i < VT; i++) {
        const int c = i * KSP_RANK_THREADS + t;
        float a = __builtin_nanf("");
        if (c < channels) {
            a = fabsf(row[c]);
            zeros += (a == 0.0f);
        }
        v[i] = a;
    }
    zeros = block_sum(zeros, &scratch);
    // zeros sort first, so the median of the non-zero values has rank
    // (channels + zeros) / 2 in the whole row (reference rank.mako:261-266)
    const int rank2 = channels + zeros;
    float med = block_select(v, rank2 / 2, !(rank2 & 1), &scratch);
    if (zeros == channels) med = __builtin_nanf("");  // numpy: median of nothing
    if (t == 0) noise[bl] = (float)((double)med * KSP_MAD_NORMAL);
}

// Channel-major variant: workgroup = 64 baselines x 16 channel phases.
#define MADNZ_PHASES 16
__global__ __launch_bounds__(64 * MADNZ_PHASES) void madnz_kernel(const float *__restrict__ in,
                                                                  float *__restrict__ noise,
                                                                  int channels, int baselines,
                                                                  int stride)
{
    __shared__ int isum[MADNZ_PHASES][64];
    __shared__ float fmx[MADNZ_PHASES][64];
    const int lane = threadIdx.x & 63;
    const int phase = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane;
    const bool active = b < baselines;
    const float *col = in + (active ? b : 0);

    // counts of keys below `pivot` (and, optionally, the largest such key)
    auto count_below = [&](unsigned pivot, float *below) -> int {
        int c = 0;
        float m = 0.0f;
        for (int ch = phase; ch < channels; ch += MADNZ_PHASES) {
            const float a = fabsf(col[(size_t)ch * stride]);
            const bool lt = __float_as_uint(a) < pivot;
            c += lt;
            if (below) m = lt ? fmaxf(m, a) : m;
        }
        __syncthreads();
        isum[phase][lane] = c;
        if (below) fmx[phase][lane] = m;
        __syncthreads();
        c = 0;
        m = 0.0f;
#pragma unroll
        for (int p = 0; p < MADNZ_PHASES; p++) {
            c += isum[p][lane];
            if (below) m = fmaxf(m, fmx[p][lane]);
        }
        if (below) *below = m;
        return c;
    };

    // number of zeros = keys below the pattern of the smallest positive float
    const int zeros = count_below(1u, nullptr);
    const int rank2 = channels + zeros;
    const int rank = rank2 / 2;
    unsigned cur = 0;
    for (int bit = 30; bit >= 0; bit--) {
        const unsigned test = cur | (1u << bit);
        const int c = count_below(test, nullptr);
        if (c <= rank) cur = test;
    }
    float result = __uint_as_float(cur);
    float below;
    const int c = count_below(cur, &below);  // barriers: every thread takes part
    if (!(rank2 & 1)) {
        const float prev = (c == rank) ? below : result;
        result = __fmul_rn(__fadd_rn(result, prev), 0.5f);
    }
    if (zeros == channels) result = __builtin_nanf("");
    if (phase == 0 && active) noise[b] = (float)((double)result * KSP_MAD_NORMAL);
}

extern "C" int ksp_madnz_t(int device, void *stream, const float *in, float *noise, int channels,
                           int baselines, int stride)
{
    KSP_REQUIRE(in != nullptr && noise != nullptr, "NULL buffer");
    KSP_REQUIRE(channels > 0 && baselines >= 0 && stride >= channels, "bad shape");
    if (baselines == 0) return 0;
    KSP_CHECK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    const int vt = ksp_divup(channels, KSP_RANK_THREADS);
#define KSP_MT(VT)                                                                             \
    hipLaunchKernelGGL(madnz_t_kernel<VT>, dim3(baselines), dim3(KSP_RANK_THREADS), 0, s, in, \
                       noise, channels, stride)
    if (vt <= 1)
        KSP_MT(1);
    else if (vt <= 2)
        KSP_MT(2);
    else if (vt <= 4)
        KSP_MT(4);
    else if (vt <= 8)
        KSP_MT(8);
    else if (vt <= 16)
        KSP_MT(16);
    else if (vt <= 24)
        KSP_MT(24);
    else if (vt <= 32)
        KSP_MT(32);
    else if (vt <= 40)
        KSP_MT(40);
    else if (vt <= 64)
        KSP_MT(64);
    else {
        ksp_set_error("ksp_madnz_t: %d channels exceed the supported maximum of %d", channels,
                      64 * KSP_RANK_THREADS);
        return (int)hipErrorInvalidValue;
    }
#undef KSP_MT
    KSP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ksp_madnz(int device, void *stream, const float *in, float *noise, int channels,
                         int baselines, int stride)
{
    KSP_REQUIRE(in != nullptr && noise != nullptr, "NULL buffer");
    KSP_REQUIRE(channels > 0 && baselines >= 0 && stride >= baselines, "bad shape");
    if (baselines == 0) return 0;
    KSP_CHECK(hipSetDevice(device));
    hipLaunchKernelGGL(madnz_kernel, dim3(ksp_divup(baselines, 64)), dim3(64 * MADNZ_PHASES), 0,
                       (hipStream_t)stream, in, noise, channels, baselines, stride);
    KSP_LAUNCH_CHECK();
    return 0;
}
