// background_median_filter: deviations = amplitude - sliding median along channels
// (stands in for reference rfi/background_median_filter.mako:200-220).
//
// Data are [C][B] with baselines contiguous, so lane <-> baseline makes every
// global access a fully coalesced 512-byte (complex64) / 256-byte (float32) row
// segment per wavefront. Each wavefront owns 64 baselines x one channel segment
// and walks the segment serially, keeping the window as a sorted register array
// (median_window.h). The next WIDTH rows are requested before the current WIDTH
// are consumed, so HBM latency overlaps the median arithmetic. The segment length
// is chosen at launch so that the grid has several thousand wavefronts.
//
// Numerics follow the host class (reference rfi/host.py:133-151): amplitude is
// numpy's complex64 abs in float32; median and subtraction are float64; the
// float32 output is the rounded float64 deviation (SortedWindow::deviation: one
// float32 subtraction when the window holds an odd number of valid samples, which
// gives the same bits); masked samples give 0.
// HBM-bound: 8 B read + 4 B written per sample (+ WIDTH-1 halo rows per segment).
//
// Segments that lie wholly inside the band, without input flags, take the MERGING
// median (median_merge.h) instead: the walk already handles the channels in blocks of
// WIDTH, and the window of an output is a sorted suffix of one block plus a sorted
// prefix of the next -- about 21 min/med3/max per sample instead of the sorted window's
// 12 compare/select pairs + 13 med3. A NaN visibility (which the host path skips, so
// the window shrinks) sends the segment back through the sorted-window walk.
#include "median_merge.h"
#include "median_window.h"

template <int WIDTH>
__global__ __launch_bounds__(256) void background_kernel(
    const void *__restrict__ in, float *__restrict__ out, const uint8_t *__restrict__ flags,
    int channels, int baselines, int stride, int flags_stride, int seg_len, int is_amplitude,
    int flags_mode)
{
    constexpr int H = WIDTH / 2;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + lane;
    const int seg = blockIdx.y * 4 + wave;
    const int c_begin = seg * seg_len;  // first output channel of this segment
    if (c_begin >= channels) return;    // whole wave exits together
    const int c_end = min(channels, c_begin + seg_len);
    const bool active = b < baselines;
    const int bb = active ? b : 0;

    // Amplitude of sample (c, bb), NaN when it is outside the band or masked.
    auto fetch = [&](int c) -> float {
        float a = __builtin_nanf("");
        if (c >= 0 && c < channels) {
            const size_t idx = (size_t)c * stride + bb;
            if (is_amplitude)
                a = ((const float *)in)[idx];
            else {
                const float2 z = ((const float2 *)in)[idx];
                a = ksp_abs_c64(z.x, z.y);
            }
            if (flags_mode == KSP_FLAGS_CHANNEL) {
                if (flags[c]) a = __builtin_nanf("");
            } else if (flags_mode == KSP_FLAGS_FULL) {
                if (flags[(size_t)c * flags_stride + bb]) a = __builtin_nanf("");
            }
        }
        return a;
    };

    // Sample c enters at step c; the output for channel c - H is ready after it.
    const int first = c_begin - H;
    const int last = c_end + H;  // exclusive
    if constexpr (WIDTH <= 13) {
        if (flags_mode == KSP_FLAGS_NONE && first >= 0 && last <= channels) {  // wave-uniform
            MergeMedian<64, WIDTH> mm;
            mm.pinf = __builtin_inff();
            mm.ninf = -__builtin_inff();
            asm volatile("" : "+v"(mm.pinf), "+v"(mm.ninf));
            bool bad = false;
            float nx[WIDTH];
            // WIDTH rows of the lane's baseline at once (all inside the band here): complex
            // input goes through the batched |z| (short division in packed pairs)
            auto fetch_rows = [&](int c0, bool clamp) {
                if (is_amplitude) {
#pragma unroll
                    for (int k = 0; k < WIDTH; k++)
                        nx[k] = ((const float *)in)[(size_t)(clamp ? min(c0 + k, channels - 1) : c0 + k) * stride + bb];
                } else {
                    float2 z[WIDTH];
#pragma unroll
                    for (int k = 0; k < WIDTH; k++)
                        z[k] = ((const float2 *)in)[(size_t)(clamp ? min(c0 + k, channels - 1) : c0 + k) * stride + bb];
                    ksp_abs_c64_rows<WIDTH>(z, nx);
                }
            };
            fetch_rows(first, false);
            // block [base, base + WIDTH) gives the outputs base + H .. base + H + WIDTH - 1
            for (int base = first; base + H < c_end; base += WIDTH) {
                float cu[WIDTH];
#pragma unroll
                for (int k = 0; k < WIDTH; k++) {
                    cu[k] = nx[k];
                    bad |= cu[k] != cu[k];
                }
                fetch_rows(base + WIDTH, true);
                float S[MergeMedian<64, WIDTH>::S_SIZE];
                S[mm.off(WIDTH - 1)] = cu[WIDTH - 1];
                ksp_static_for<WIDTH - 1>([&](auto u_) {
                    constexpr int t = WIDTH - 2 - decltype(u_)::value;
                    mm.template insert<WIDTH - 1 - t>(&S[mm.off(t + 1)], cu[t], &S[mm.off(t)]);
                });
                float P[WIDTH];
                ksp_static_for<WIDTH>([&](auto t_) {
                    constexpr int t = decltype(t_)::value;
                    if constexpr (t >= 1) {
                        bad |= nx[t - 1] != nx[t - 1];
                        if constexpr (t == 1)
                            P[0] = nx[0];
                        else
                            mm.template insert<t - 1>(P, nx[t - 1], P);
                    }
                    const float med = mm.template rank<WIDTH - t, t, H>(&S[mm.off(t)], P);
                    const float xc = (t + H < WIDTH) ? cu[t + H < WIDTH ? t + H : 0]
                                                     : nx[t + H >= WIDTH ? t + H - WIDTH : 0];
                    const int oc = base + H + t;
                    if (oc < c_end && active) out[(size_t)oc * stride + b] = xc - med;
                });
            }
            // (samples up to c_end + H - 1 were all looked at: cu of every block, nx of the last)
            if (!__any(bad)) return;
        }
    }

    SortedWindow<WIDTH> win;
    win.reset();
    float ring[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; i++) ring[i] = __builtin_nanf("");

    // WIDTH rows starting at channel c0 (any of them may lie outside the band): loads of
    // values and flag bytes first, then the batched |z|, then the masks
    auto fetch_block = [&](int c0, float (&dst)[WIDTH]) {
        if (is_amplitude) {
#pragma unroll
            for (int k = 0; k < WIDTH; k++) dst[k] = fetch(c0 + k);
            return;
        }
        float2 z[WIDTH];
        uint8_t f[WIDTH];
#pragma unroll
        for (int k = 0; k < WIDTH; k++) {
            const int cc = min(max(c0 + k, 0), channels - 1);
            z[k] = ((const float2 *)in)[(size_t)cc * stride + bb];
            f[k] = 0;
            if (flags_mode == KSP_FLAGS_CHANNEL)
                f[k] = flags[cc];
            else if (flags_mode == KSP_FLAGS_FULL)
                f[k] = flags[(size_t)cc * flags_stride + bb];
        }
        ksp_abs_c64_rows<WIDTH>(z, dst);
#pragma unroll
        for (int k = 0; k < WIDTH; k++) {
            const int c = c0 + k;
            if (c < 0 || c >= channels || f[k]) dst[k] = __builtin_nanf("");
        }
    };
    float nxt[WIDTH];
    fetch_block(first, nxt);
    for (int base = first; base < last; base += WIDTH) {
        float cur[WIDTH];
#pragma unroll
        for (int k = 0; k < WIDTH; k++) cur[k] = nxt[k];
        if (base + WIDTH < last) fetch_block(base + WIDTH, nxt);
#pragma unroll
        for (int k = 0; k < WIDTH; k++) {
            const int c = base + k;  // entering sample
            if (c < last) {          // wave-uniform
                float leaving = ring[k];
                asm("" : "+v"(leaving));  // opaque: do not carry the entry-time mask along
                win.step(leaving, leaving == leaving, cur[k], cur[k] == cur[k]);
                ring[k] = cur[k];
                const int oc = c - H;  // output channel
                if (oc >= c_begin) {   // wave-uniform; oc < c_end holds since c < last
                    float xc = ring[(k + WIDTH - H) % WIDTH];
                    asm("" : "+v"(xc));
                    float d = win.deviation(xc);
                    d = (xc == xc) ? d : 0.0f;
                    if (active) out[(size_t)oc * stride + b] = d;
                }
            }
        }
    }
}

template <int WIDTH>
static int launch_background(hipStream_t s, const void *in, float *out, const uint8_t *flags,
                             int channels, int baselines, int stride, int flags_stride,
                             int is_amplitude, int flags_mode, int csplit)
{
    const int wave_cols = ksp_divup(baselines, 64);
    // csplit = number of channel segments a baseline is cut into (each segment re-reads
    // WIDTH - 1 channels of halo); 0: aim for >= 8192 wavefronts. Segments are at least
    // 4 * WIDTH channels long either way.
    int want_segs = csplit > 0 ? csplit : ksp_divup(8192, wave_cols);
    int seg_len = ksp_divup(channels, want_segs);
    if (seg_len < 4 * WIDTH) seg_len = 4 * WIDTH;
    if (seg_len > channels) seg_len = channels;
    const int segs = ksp_divup(channels, seg_len);
    dim3 grid(wave_cols, ksp_divup(segs, 4));
    hipLaunchKernelGGL(background_kernel<WIDTH>, grid, dim3(256), 0, s, in, out, flags, channels,
                       baselines, stride, flags_stride, seg_len, is_amplitude, flags_mode);
    KSP_LAUNCH_CHECK();
    return 0;
}

extern "C" int ksp_background_median_filter(int device, void *stream, const void *in, float *out,
                                            const uint8_t *flags, int channels, int baselines,
                                            int stride, int flags_stride, int width,
                                            int is_amplitude, int flags_mode, int csplit)
{
    KSP_REQUIRE(in != nullptr && out != nullptr, "NULL buffer");
    KSP_REQUIRE(channels >= 0 && baselines >= 0 && stride >= baselines, "bad shape");
    KSP_REQUIRE(width >= 3 && (width & 1), "width must be odd and >= 3");
    KSP_REQUIRE(flags_mode >= KSP_FLAGS_NONE && flags_mode <= KSP_FLAGS_FULL, "bad flags_mode");
    KSP_REQUIRE(flags_mode == KSP_FLAGS_NONE || flags != nullptr, "flags buffer is NULL");
    KSP_REQUIRE(flags_mode != KSP_FLAGS_FULL || flags_stride >= baselines, "bad flags_stride");
    KSP_REQUIRE(csplit >= 0, "csplit is negative");
    if (channels == 0 || baselines == 0) return 0;
    KSP_CHECK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
#define KSP_BG(W)                                                                               \
    case W:                                                                                     \
        return launch_background<W>(s, in, out, flags, channels, baselines, stride, flags_stride, \
                                    is_amplitude, flags_mode, csplit)
    switch (width) {
        KSP_BG(3);
        KSP_BG(5);
        KSP_BG(7);
        KSP_BG(9);
        KSP_BG(11);
        KSP_BG(13);
        KSP_BG(15);
        KSP_BG(17);
        KSP_BG(19);
        KSP_BG(21);
        KSP_BG(23);
        KSP_BG(25);
        KSP_BG(27);
        KSP_BG(29);
        KSP_BG(31);
    default:
        ksp_set_error("ksp_background_median_filter: width %d has no compiled kernel "
                      "(available: odd 3..31)", width);
        return (int)hipErrorInvalidValue;
    }
#undef KSP_BG
}
