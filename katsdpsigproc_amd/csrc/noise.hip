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
#include <atomic>

#include "bitplane.h"
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
#pragma unroll
    for (int i = 0; i < VT; i++) {
        const int c = i * KSP_RANK_THREADS + t;
        v[i] = (c < channels) ? fabsf(row[c]) : __builtin_nanf("");
    }
    const float med = block_median_non_zero(v, channels, &scratch);
    if (t == 0) noise[bl] = (float)((double)med * KSP_MAD_NORMAL);
}

// ----------------------------------------------------------------------------
// madnz_t for rows of up to 4096 channels: ONE WAVEFRONT per baseline, lane l holds
// channels [64 l, 64 l + 64) (256 contiguous bytes per lane, 16 KiB per wavefront).
// The 31-bit patterns of |x| are transposed into 31 bit planes of 64 bits per lane
// (two in-register 32 x 32 bit transposes), after which one step of the bit-wise
// rank search is a few ANDs and population counts per lane plus DPP wave reductions
// -- two bits per step, no LDS, no barriers -- instead of a compare per value and a
// workgroup reduction per bit.
// The search itself: u[64] are the lane's 64 |x| patterns (0x7fffffff for channels that
// do not exist); returns the float32 median of the non-zero values of the row (NaN if
// there are none), identical in every lane.
__device__ __forceinline__ float wave_median_nonzero(const unsigned (&u)[64], int channels)
{
    // inverted bit planes: hi[b] / hi[16 + b] = bit 16 + b of the even / odd values
    // clear; lo[...] likewise for bits 0..15
    unsigned hi[32], lo[32];
#pragma unroll
    for (int i = 0; i < 32; i++) {
        hi[i] = ~__builtin_amdgcn_perm(u[2 * i + 1], u[2 * i], 0x07060302u);
        lo[i] = ~__builtin_amdgcn_perm(u[2 * i + 1], u[2 * i], 0x05040100u);
    }
    transpose_bits32(hi);
    transpose_bits32(lo);
    auto plane = [&](int bit, int half) -> unsigned {
        return bit >= 16 ? hi[16 * half + bit - 16] : lo[16 * half + bit];
    };
    // zeros: every one of the 31 bits clear
    unsigned z0 = lo[0], z1 = lo[16];
#pragma unroll
    for (int b = 1; b < 16; b++) {
        z0 &= lo[b];
        z1 &= lo[16 + b];
    }
#pragma unroll
    for (int b = 0; b < 15; b++) {
        z0 &= hi[b];
        z1 &= hi[16 + b];
    }
    const int zeros = ksp_wave_sum_dpp(__popc(z0) + __popc(z1));
    // zeros sort first, so the median of the non-zero values has rank
    // (channels + zeros) / 2 in the whole row (reference rank.mako:261-266)
    const int rank2 = channels + zeros;
    const int rank = rank2 / 2;
    PlaneSearch<decltype(plane)> search(rank, plane);
    // (one bit per step: measured faster than two-bit steps here as in the fused kernel)
    search.template step1<30>();
    search.template step1<29>();
    search.template step1<28>();
    search.template step1<27>();
    search.template step1<26>();
    search.template step1<25>();
    search.template step1<24>();
    search.template step1<23>();
    search.template step1<22>();
    search.template step1<21>();
    search.template step1<20>();
    search.template step1<19>();
    search.template step1<18>();
    search.template step1<17>();
    search.template step1<16>();
    search.template step1<15>();
    search.template step1<14>();
    search.template step1<13>();
    search.template step1<12>();
    search.template step1<11>();
    search.template step1<10>();
    search.template step1<9>();
    search.template step1<8>();
    search.template step1<7>();
    search.template step1<6>();
    search.template step1<5>();
    search.template step1<4>();
    search.template step1<3>();
    search.template step1<2>();
    search.template step1<1>();
    search.template step1<0>();
    const unsigned pat = search.prefix;  // pattern of the value of rank `rank`
    float result = __uint_as_float(pat);
    if (!(rank2 & 1)) {
        // even count: also the value of rank - 1 -- the same value if it occurs below
        // rank as well, else the largest value below it
        float prev = result;
        if (search.below == rank) {
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < 64; i++) m = max(m, u[i] < pat ? u[i] : 0u);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
            prev = __uint_as_float(m);
        }
        result = __fmul_rn(__fadd_rn(result, prev), 0.5f);
    }
    if (zeros == channels) result = __builtin_nanf("");  // numpy: median of nothing
    return result;
}

// Self-test of the rank library (tests/test_gpu_ops.py::TestRankLibrary, the
// counterpart of reference test/test_rank.py:161-213): median of the non-zero values of
// `n` <= 16384 non-negative floats by the workgroup search (out[0]) and, for n <= 4096,
// by the wavefront bit-plane search (out[1]; else the workgroup search again).
template <int VT>
__global__ __launch_bounds__(KSP_RANK_THREADS) void selftest_median_non_zero_kernel(
    const float *__restrict__ data, int n, float *__restrict__ out)
{
    __shared__ RankScratch scratch;
    const int t = threadIdx.x;
    float v[VT];
#pragma unroll
    for (int i = 0; i < VT; i++) {
        const int c = i * KSP_RANK_THREADS + t;
        v[i] = (c < n) ? data[c] : __builtin_nanf("");
    }
    const float med = block_median_non_zero(v, n, &scratch);
    if (t == 0) out[0] = out[1] = med;
    if (n <= 4096 && t < 64) {
        unsigned u[64];
#pragma unroll
        for (int i = 0; i < 64; i++)
            u[i] = (t * 64 + i < n) ? (__float_as_uint(data[t * 64 + i]) & 0x7fffffffu) : 0x7fffffffu;
        const float w = wave_median_nonzero(u, n);
        if (t == 0) out[1] = w;
    }
}

extern "C" int ksp_selftest_median_non_zero(int device, void *stream, const float *data,
                                            float *out, int n)
{
    KSP_REQUIRE(data != nullptr && out != nullptr && n >= 1 && n <= 64 * KSP_RANK_THREADS,
                "bad arguments");
    KSP_CHECK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    if (n <= 4 * KSP_RANK_THREADS)
        hipLaunchKernelGGL(selftest_median_non_zero_kernel<4>, dim3(1), dim3(KSP_RANK_THREADS), 0,
                           s, data, n, out);
    else if (n <= 40 * KSP_RANK_THREADS)
        hipLaunchKernelGGL(selftest_median_non_zero_kernel<40>, dim3(1), dim3(KSP_RANK_THREADS), 0,
                           s, data, n, out);
    else
        hipLaunchKernelGGL(selftest_median_non_zero_kernel<64>, dim3(1), dim3(KSP_RANK_THREADS), 0,
                           s, data, n, out);
    KSP_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void madnz_t_wave_kernel(const float *__restrict__ in,
                                                           float *__restrict__ noise, int channels,
                                                           int baselines, int stride, int vec_ok)
{
    const int lane = threadIdx.x & 63;
    const int bl = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bl >= baselines) return;  // whole wavefronts leave together
    const float *row = in + (size_t)bl * stride + lane * 64;
    // |x| patterns; channels that do not exist get the largest pattern and are never
    // reached because ranks are taken among `channels` values
    unsigned u[64];
    if (vec_ok && lane * 64 + 64 <= channels) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint4 q = *(const uint4 *)(row + 4 * i);
            u[4 * i] = q.x & 0x7fffffffu;
            u[4 * i + 1] = q.y & 0x7fffffffu;
            u[4 * i + 2] = q.z & 0x7fffffffu;
            u[4 * i + 3] = q.w & 0x7fffffffu;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 64; i++)
            u[i] = (lane * 64 + i < channels) ? (__float_as_uint(row[i]) & 0x7fffffffu) : 0x7fffffffu;
    }
    const float result = wave_median_nonzero(u, channels);
    if (lane == 0) noise[bl] = (float)((double)result * KSP_MAD_NORMAL);
}

// madnz (channel-major input) for up to 4096 channels: a 512-thread workgroup brings a
// strip of 8 adjacent baselines x all channels into LDS (32-byte row segments, the
// layout of the fused flagger: [baseline][channel], lane runs padded by 4 words), then
// each of its 8 wavefronts runs the bit-plane search on its baseline. One pass over the
// data instead of the 33 of the per-bit re-reading kernel below.
#define MADNZ_STRIP 8
#define MADNZ_RUN 68                        // 64 channels + 4 words of padding per lane
#define MADNZ_ROW (64 * MADNZ_RUN + 8)      // words per baseline
__global__ __launch_bounds__(64 * MADNZ_STRIP) void madnz_strip_kernel(
    const float *__restrict__ in, float *__restrict__ noise, int channels, int baselines, int stride)
{
    extern __shared__ __attribute__((aligned(16))) float img[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * MADNZ_STRIP;
    // cooperative load: 2 lanes x 16 B per row, 256 rows per pass
    const int q = tid & 1, r0 = tid >> 1;
    const int bq = b0 + 4 * q;
    const bool vec = (stride % 4 == 0) && ((size_t)in % 16 == 0) && (bq + 4 <= baselines);
    for (int row = r0; row < 4096; row += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < channels) {
            const float *src = in + (size_t)row * stride + bq;
            if (vec) {
                v = *(const float4 *)src;
            } else {
                if (bq < baselines) v.x = src[0];
                if (bq + 1 < baselines) v.y = src[1];
                if (bq + 2 < baselines) v.z = src[2];
                if (bq + 3 < baselines) v.w = src[3];
            }
        }
        const int idx = (row >> 6) * MADNZ_RUN + (row & 63);
        img[(4 * q + 0) * MADNZ_ROW + idx] = v.x;
        img[(4 * q + 1) * MADNZ_ROW + idx] = v.y;
        img[(4 * q + 2) * MADNZ_ROW + idx] = v.z;
        img[(4 * q + 3) * MADNZ_ROW + idx] = v.w;
    }
    __syncthreads();
    const int bl = b0 + wave;
    if (bl >= baselines) return;  // whole wavefronts leave together
    const float *run = img + wave * MADNZ_ROW + lane * MADNZ_RUN;
    unsigned u[64];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint4 w = *(const uint4 *)(run + 4 * i);
        u[4 * i] = w.x & 0x7fffffffu;
        u[4 * i + 1] = w.y & 0x7fffffffu;
        u[4 * i + 2] = w.z & 0x7fffffffu;
        u[4 * i + 3] = w.w & 0x7fffffffu;
    }
#pragma unroll
    for (int i = 0; i < 64; i++)
        if (lane * 64 + i >= channels) u[i] = 0x7fffffffu;
    const float result = wave_median_nonzero(u, channels);
    if (lane == 0) noise[bl] = (float)((double)result * KSP_MAD_NORMAL);
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
    if (channels > 1024 && channels <= 4096) {
        const int vec_ok = (stride % 4 == 0) && ((uintptr_t)in % 16 == 0);
        hipLaunchKernelGGL(madnz_t_wave_kernel, dim3(ksp_divup(baselines, 4)), dim3(256), 0, s, in,
                           noise, channels, baselines, stride, vec_ok);
        KSP_LAUNCH_CHECK();
        return 0;
    }
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
    if (channels <= 4096) {
        const size_t lds = sizeof(float) * MADNZ_STRIP * MADNZ_ROW;
        static std::atomic<bool> attr_set[64];
        if (device < 0 || device >= 64 || !attr_set[device].load(std::memory_order_acquire)) {
            KSP_CHECK(hipFuncSetAttribute((const void *)madnz_strip_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (device >= 0 && device < 64) attr_set[device].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(madnz_strip_kernel, dim3(ksp_divup(baselines, MADNZ_STRIP)),
                           dim3(64 * MADNZ_STRIP), lds, (hipStream_t)stream, in, noise, channels,
                           baselines, stride);
        KSP_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(madnz_kernel, dim3(ksp_divup(baselines, 64)), dim3(64 * MADNZ_PHASES), 0,
                       (hipStream_t)stream, in, noise, channels, baselines, stride);
    KSP_LAUNCH_CHECK();
    return 0;
}
