// Thresholding kernels.
//
// threshold_simple(_t): flags = dev > n_sigma * noise[baseline] (stands in for
// reference rfi/threshold_simple.mako:27-40 and rfi/threshold_simple_t.mako:28-42).
// Pure streaming: 4 elements per lane (16-byte loads, 4-byte stores) when the rows
// are 16-byte aligned. HBM-bound at 5 bytes per sample.
//
// threshold_sum: Offringa SumThreshold along channels on baseline-major data
// (stands in for reference rfi/threshold_sum.mako:49-132). One 256-thread
// workgroup owns a chunk of one baseline; each thread owns VT consecutive channels.
// The chunk lives in LDS as float32 (every value is either a float32 deviation or
// a float32 threshold, so this loses nothing); per window the threads
//   1. overwrite already-flagged samples with the window's threshold (LDS),
//   2. read their VT + w - 1 values and form the w-term sums sequentially in
//      float64 -- the order and precision of numpy.convolve in the host class
//      (reference rfi/host.py:239-242), NOT the float32 Kogge-Stone tree of the
//      reference kernel -- comparing against float32(threshold * w); for windows up
//      to 8 only at the positions whose window holds a sample above the window's
//      threshold (no other window can fire),
//   3. exchange hit bit-masks with the left neighbour and dilate them with shifts.
// Only full windows inside the band count (host `mode="valid"`); the reference
// kernel's zero padding at the band edges is deliberately not reproduced.
// When a baseline does not fit one chunk, chunks overlap by the reference's
// EDGE = 2^n - n - 1 halo (rfi/device.py:848-850) and each writes only its core.
#include "ksp_common.h"

// ----------------------------------------------------------------------------
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void threshold_simple_kernel(
    const float *__restrict__ dev, const float *__restrict__ noise, uint8_t *__restrict__ flags,
    int rows, int cols, int stride, float n_sigma, uint8_t flag_value, int vec_ok)
{
    const int row = blockIdx.y;
    const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c0 >= cols) return;
    const size_t base = (size_t)row * stride + c0;
    float thr_row = 0.0f;
    if (TRANSPOSED) thr_row = __fmul_rn(n_sigma, noise[row]);
    if (vec_ok && c0 + 4 <= cols) {
        const float4 d = *(const float4 *)(dev + base);
        float t0 = thr_row, t1 = thr_row, t2 = thr_row, t3 = thr_row;
        if (!TRANSPOSED) {
            t0 = __fmul_rn(n_sigma, noise[c0]);
            t1 = __fmul_rn(n_sigma, noise[c0 + 1]);
            t2 = __fmul_rn(n_sigma, noise[c0 + 2]);
            t3 = __fmul_rn(n_sigma, noise[c0 + 3]);
        }
        uchar4 f;
        f.x = d.x > t0 ? flag_value : 0;
        f.y = d.y > t1 ? flag_value : 0;
        f.z = d.z > t2 ? flag_value : 0;
        f.w = d.w > t3 ? flag_value : 0;
        *(uchar4 *)(flags + base) = f;
    } else {
        for (int i = 0; i < 4 && c0 + i < cols; i++) {
            const float t = TRANSPOSED ? thr_row : __fmul_rn(n_sigma, noise[c0 + i]);
            flags[base + i] = dev[base + i] > t ? flag_value : 0;
        }
    }
}

extern "C" int ksp_threshold_simple(int device, void *stream, const float *deviations,
                                    const float *noise, uint8_t *flags, int rows, int cols,
                                    int stride, float n_sigma, int flag_value, int transposed)
{
    KSP_REQUIRE(deviations != nullptr && noise != nullptr && flags != nullptr, "NULL buffer");
    KSP_REQUIRE(rows >= 0 && cols >= 0 && stride >= cols, "bad shape");
    if (rows == 0 || cols == 0) return 0;
    KSP_CHECK(hipSetDevice(device));
    const int vec_ok = (stride % 4 == 0) && ((uintptr_t)deviations % 16 == 0) &&
                       ((uintptr_t)flags % 4 == 0);
    // grid.y holds at most 65535 rows: larger arrays go in slices
    for (int r0 = 0; r0 < rows; r0 += 65535) {
        const int nr = rows - r0 < 65535 ? rows - r0 : 65535;
        dim3 grid(ksp_divup(cols, 1024), nr);
        const float *d = deviations + (size_t)r0 * stride;
        uint8_t *f = flags + (size_t)r0 * stride;
        if (transposed)
            hipLaunchKernelGGL(threshold_simple_kernel<true>, grid, dim3(256), 0,
                               (hipStream_t)stream, d, noise + r0, f, nr, cols, stride, n_sigma,
                               (uint8_t)flag_value, vec_ok);
        else
            hipLaunchKernelGGL(threshold_simple_kernel<false>, grid, dim3(256), 0,
                               (hipStream_t)stream, d, noise, f, nr, cols, stride, n_sigma,
                               (uint8_t)flag_value, vec_ok);
    }
    KSP_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------
struct SumParams {
    float scales[KSP_MAX_WINDOWS];
};

template <int VT>
__global__ __launch_bounds__(256) void threshold_sum_kernel(
    const float *__restrict__ dev, const float *__restrict__ noise, uint8_t *__restrict__ flags,
    int channels, int stride, float n_sigma, SumParams params, int n_windows, uint8_t flag_value,
    int core, int edge)
{
    constexpr int TOT = 256 * VT;
    constexpr int MAXW = 1 << (KSP_MAX_WINDOWS - 1);  // largest window
    static_assert(VT >= 8 && VT < 57, "neighbour exchange assumes 8 <= VT < 57");
    __shared__ float vals[TOT + MAXW];
    __shared__ unsigned long long hitmask[256];
    __shared__ unsigned long long hotmask[256];
    __shared__ int lasthit[256];
    __shared__ uint8_t fbytes[TOT];

    const int t = threadIdx.x;
    const int bl = blockIdx.y;
    const int chunk = blockIdx.x;
    const int base = chunk * core - edge;  // global channel of local position 0
    const float *row = dev + (size_t)bl * stride;

    // coalesced load of the chunk; positions outside the band hold 0 and are
    // excluded from every sum by the validity test below
    for (int j = t; j < TOT + MAXW; j += 256) {
        const int g = base + j;
        vals[j] = (j < TOT && g >= 0 && g < channels) ? row[g] : 0.0f;
    }
    const float t1 = __fmul_rn(n_sigma, noise[bl]);
    unsigned long long fl = 0;  // bit i = channel t*VT + i is flagged
    const int j0 = t * VT;
    __syncthreads();

    // Exact fast reject: thresholds are positive and every window's limit is w * thr_k,
    // so if no sample of the chunk reaches min_k thr_k (with a 2^-20 margin that covers
    // the rounding of the float64 sums) no window can fire and nothing is flagged. Data
    // without interference takes this exit; the result is the same either way.
    {
        float thr_min = __builtin_inff();
        bool thr_nan = false;
        for (int k = 0; k < n_windows; k++) {
            const float thr = __fmul_rn(t1, params.scales[k]);
            thr_min = fminf(thr_min, thr);
            thr_nan |= (thr != thr);
        }
        float m = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < VT; i++) m = fmaxf(m, vals[j0 + i]);  // NaN samples never fire
        const bool candidate = !(thr_min > 0.0f) || thr_nan || ((double)m >= (double)thr_min * (1.0 - 0x1p-20));
        if (!__syncthreads_or(candidate && !thr_nan)) {
            uint8_t *frow0 = flags + (size_t)bl * stride;
            for (int j = t; j < TOT; j += 256) {
                const int g = base + j;
                if (j >= edge && j < edge + core && g < channels) frow0[g] = 0;
            }
            return;
        }
    }

    for (int k = 0; k < n_windows; k++) {
        const int w = 1 << k;
        const float thr = __fmul_rn(t1, params.scales[k]);
        const double limit = (double)__fmul_rn(thr, (float)w);
        // 1. already-flagged samples contribute exactly thr (host.py:237)
        if (k > 0) {
#pragma unroll
            for (int i = 0; i < VT; i++)
                if ((fl >> i) & 1) vals[j0 + i] = thr;
            __syncthreads();
        }
        // 2. sums of w consecutive values, sequential float64. Only where they can matter:
        //    w values that are all <= thr sum to <= w * thr (the float64 sum of w copies of a
        //    float32 is exact for these w, and rounding is monotone), so a window that fires
        //    holds a sample > thr -- an unflagged one, flagged ones stand in as thr itself.
        //    Each thread marks such samples among its VT, sees its right neighbour's marks
        //    (windows of up to 8 reach 7 positions into them) and sums only the windows
        //    that hold one: a handful per chunk once strong interference has been flagged
        //    by window 1, instead of VT * (1 + 2 + 4 + 8) additions.
        unsigned long long hits = 0;
        if (thr > 0.0f && w <= 8) {
            unsigned long long hot = 0;
#pragma unroll
            for (int i = 0; i < VT; i++) hot |= (unsigned long long)(vals[j0 + i] > thr) << i;
            hotmask[t] = hot;
            __syncthreads();
            const unsigned long long next = t < 255 ? hotmask[t + 1] : 0ull;
            const unsigned long long span = hot | (next << VT);  // VT + 7 <= 39 < 64 bits used
            unsigned long long need = span;
            if (w >= 2) need |= need >> 1;
            if (w >= 4) need |= need >> 2;
            if (w >= 8) need |= need >> 4;
            need &= (1ull << VT) - 1;
            while (need) {
                const int i = __ffsll((long long)need) - 1;
                need &= need - 1;
                double s = 0.0;
                for (int m = 0; m < w; m++) s += (double)vals[j0 + i + m];
                const int g = base + j0 + i;
                const bool valid = (g >= 0) && (g + w <= channels) && (j0 + i + w <= TOT);
                if (valid && s > limit) hits |= 1ull << i;
            }
        } else {
#pragma unroll
            for (int i = 0; i < VT; i++) {
                double s = 0.0;
                for (int m = 0; m < w; m++) s += (double)vals[j0 + i + m];
                const int g = base + j0 + i;
                const bool valid = (g >= 0) && (g + w <= channels) && (j0 + i + w <= TOT);
                if (valid && s > limit) hits |= 1ull << i;
            }
        }
        // 3. dilate: a hit at j flags j .. j+w-1, possibly into the next thread(s)
        if (w <= 8) {
            hitmask[t] = hits;
            __syncthreads();
            const unsigned long long prev = t > 0 ? hitmask[t - 1] : 0ull;
            // bits 0..6 = previous thread's last 7 positions (VT >= 8 > w - 1),
            // bits 7.. = own positions
            unsigned long long comb = (hits << 7) | ((prev >> (VT - 7)) & 0x7full);
            if (w >= 2) comb |= comb << 1;
            if (w >= 4) comb |= comb << 2;
            if (w >= 8) comb |= comb << 4;
            fl |= (comb >> 7) & ((1ull << VT) - 1);
        } else {
            // wide windows reach back over several threads: a position is flagged when
            // the nearest hit at or before it is fewer than w positions away
            constexpr int NONE = -(1 << 30);
            lasthit[t] = hits ? j0 + 63 - __clzll((long long)hits) : NONE;
            __syncthreads();
            int carry = NONE;
            const int reach = (w - 1 + VT - 1) / VT;
            for (int q = 1; q <= reach && q <= t; q++) carry = max(carry, lasthit[t - q]);
#pragma unroll
            for (int i = 0; i < VT; i++) {
                if ((hits >> i) & 1) carry = j0 + i;
                if (j0 + i - carry < w) fl |= 1ull << i;
            }
        }
        // hitmask / lasthit are rewritten only after the next window's first barrier
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VT; i++) fbytes[j0 + i] = ((fl >> i) & 1) ? flag_value : 0;
    __syncthreads();
    // each chunk writes its core [chunk*core, (chunk+1)*core) only
    uint8_t *frow = flags + (size_t)bl * stride;
    for (int j = t; j < TOT; j += 256) {
        const int g = base + j;
        if (j >= edge && j < edge + core && g < channels) frow[g] = fbytes[j];
    }
}

extern "C" int ksp_threshold_sum(int device, void *stream, const float *deviations,
                                 const float *noise, uint8_t *flags, int channels, int baselines,
                                 int stride, float n_sigma, const float *scales, int n_windows,
                                 int flag_value, int vt)
{
    KSP_REQUIRE(deviations != nullptr && noise != nullptr && flags != nullptr, "NULL buffer");
    KSP_REQUIRE(scales != nullptr, "scales is NULL");
    KSP_REQUIRE(channels >= 0 && baselines >= 0 && stride >= channels, "bad shape");
    KSP_REQUIRE(n_windows >= 1 && n_windows <= KSP_MAX_WINDOWS,
                "n_windows must be 1..8 (windows up to 128)");
    if (channels == 0 || baselines == 0) return 0;
    KSP_CHECK(hipSetDevice(device));
    SumParams p;
    for (int k = 0; k < KSP_MAX_WINDOWS; k++) p.scales[k] = k < n_windows ? scales[k] : 0.0f;
    const int edge = (1 << n_windows) - n_windows - 1;
    hipStream_t s = (hipStream_t)stream;
#define KSP_TS(VT)                                                                              \
    do {                                                                                        \
        const int tot = 256 * VT;                                                               \
        const int core = (channels <= tot) ? tot : tot - 2 * edge;                        \
        const int chunks = ksp_divup(channels, core);                                            \
        const int e = (chunks == 1) ? 0 : edge;                                                  \
        /* grid.y holds at most 65535 baselines: larger arrays go in slices */                  \
        for (int b0 = 0; b0 < baselines; b0 += 65535) {                                          \
            const int nb = baselines - b0 < 65535 ? baselines - b0 : 65535;                      \
            hipLaunchKernelGGL(threshold_sum_kernel<VT>, dim3(chunks, nb), dim3(256), 0, s,      \
                               deviations + (size_t)b0 * stride, noise + b0,                     \
                               flags + (size_t)b0 * stride, channels, stride, n_sigma, p,        \
                               n_windows, (uint8_t)flag_value, core, e);                         \
        }                                                                                        \
    } while (0)
    // vt = channels per thread (8, 16 or 32: chunks of 2048, 4096 or 8192 channels); 0: the
    // smallest that holds the baseline in one chunk (no halo), 32 beyond that
    KSP_REQUIRE(vt == 0 || vt == 8 || vt == 16 || vt == 32, "vt must be 0, 8, 16 or 32");
    if (vt == 0) vt = channels <= 256 * 8 ? 8 : channels <= 256 * 16 ? 16 : 32;
    const int edge_all = (1 << n_windows) - n_windows - 1;
    KSP_REQUIRE(channels <= 256 * vt || 256 * vt > 2 * edge_all, "chunk shorter than its halo");
    if (vt == 8)
        KSP_TS(8);
    else if (vt == 16)
        KSP_TS(16);
    else
        KSP_TS(32);
#undef KSP_TS
    KSP_LAUNCH_CHECK();
    return 0;
}
