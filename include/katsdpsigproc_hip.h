/*
 * katsdpsigproc_hip.h -- C-ABI of the MI355X (gfx950) RFI-flagging library.
 *
 * The reference (ska-sa/katsdpsigproc) is pure Python: its device kernels are
 * Mako-templated C compiled at run time and launched through PyCUDA/PyOpenCL, so
 * it has no FFI of its own for this path. This header defines the boundary that
 * replaces that run-time compile-and-launch layer: every entry point names the
 * reference interface it stands in for (file:line below). All functions are
 * `extern "C"`, take plain pointers and sizes, return 0 on success or a non-zero
 * hipError_t, and record a message retrievable with ksp_last_error().
 *
 * Conventions (reference: SURVEY.md conventions; rfi/device.py docstrings):
 *   C = channels, B = baselines. Non-transposed arrays are [C][B] (baseline
 *   contiguous); `_t` arrays are [B][C]. Strides are in ELEMENTS of the array's
 *   dtype (the reference passes buffer.padded_shape[1], e.g. rfi/device.py:319).
 *   Device pointers must come from ksp_malloc (or any hipMalloc of the same
 *   process); kernels never allocate. `stream` is a hipStream_t (NULL = default).
 *   Launches are asynchronous and in-order on their stream, like the reference's
 *   command queues (doc/user/init.rst:37-39).
 */
#ifndef KATSDPSIGPROC_HIP_H
#define KATSDPSIGPROC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSP_ABI_VERSION 5

/* BackgroundFlags (reference: rfi/device.py:40-46) */
#define KSP_FLAGS_NONE 0
#define KSP_FLAGS_CHANNEL 1
#define KSP_FLAGS_FULL 2

/* threshold kinds for the fused flagger */
#define KSP_THRESHOLD_SIMPLE 0
#define KSP_THRESHOLD_SUM 1

#define KSP_MAX_WINDOWS 8

typedef struct ksp_device_props {
    char name[256];          /* AbstractDevice.name            (abc.py:105-108) */
    char arch[64];           /* gcnArchName, e.g. "gfx950:sramecc+:xnack-"      */
    int32_t compute_units;
    int32_t wavefront_size;  /* AbstractDevice.simd_group_size (abc.py:140-147) */
    int32_t max_threads_per_block;
    int32_t lds_bytes_per_block;
    int32_t clock_khz;
    int32_t driver_version;  /* AbstractDevice.driver_version  (abc.py:115-118) */
    int32_t runtime_version;
    int64_t total_memory;
} ksp_device_props;

/* ---- library / errors ---------------------------------------------------- */
int ksp_abi_version(void);
const char *ksp_last_error(void);

/* ---- devices (reference: cuda.py Device 87-160; abc.py:98-157) ------------ */
int ksp_device_count(int *count);
int ksp_device_get_props(int device, ksp_device_props *props);

/* ---- memory (reference: cuda.py Context.allocate_raw / allocate_pinned,
 *      cuda.py:189-222; abc.py:180-208) ------------------------------------- */
int ksp_malloc(int device, size_t bytes, void **ptr);
int ksp_free(int device, void *ptr);
int ksp_host_alloc(size_t bytes, void **ptr);
int ksp_host_free(void *ptr);

/* ---- streams and events (reference: cuda.py CommandQueue 249-479, Event 54-84;
 *      abc.py:71-95, 434-448) ------------------------------------------------ */
int ksp_stream_create(int device, void **stream);
int ksp_stream_destroy(int device, void *stream);
int ksp_stream_synchronize(int device, void *stream);
int ksp_event_create(int device, void **event);
/* An event that only orders work between streams of this device: no time stamp and no
 * system-scope fence (cache write-back + invalidate) when it is recorded. Not for
 * ksp_event_elapsed_ms, and not for handing results to the host. */
int ksp_event_create_ordering(int device, void **event);
int ksp_event_destroy(int device, void *event);
int ksp_event_record(int device, void *event, void *stream);
int ksp_event_synchronize(int device, void *event);
int ksp_event_elapsed_ms(int device, void *start, void *end, float *ms);
int ksp_stream_wait_event(int device, void *stream, void *event);

/* ---- copies (reference: abc.py:253-404; cuda.py:263-440). kind: 0 = host to
 *      device, 1 = device to host, 2 = device to device. The rect form copies
 *      shape[0] bytes x shape[1] x shape[2] with byte strides (strides[0] == 1),
 *      exactly the contract of enqueue_*_buffer_rect (abc.py:291-322). --------- */
int ksp_memcpy_async(int device, void *dst, const void *src, size_t bytes, int kind,
                     void *stream);
int ksp_memcpy_rect_async(int device, void *dst, size_t dst_origin, const size_t dst_strides[3],
                          const void *src, size_t src_origin, const size_t src_strides[3],
                          const size_t shape[3], int ndim, int kind, void *stream);
int ksp_memset_async(int device, void *ptr, int value, size_t bytes, void *stream);

/* ========================================================================== *
 *  Kernels. One launcher per reference kernel.
 * ========================================================================== */

/* transpose (reference: transpose.mako:44-73; launch transpose.py:146-167).
 * dst[c][r] = src[r][c]; elem_size in {1,2,4,8,16} bytes. */
int ksp_transpose(int device, void *stream, void *dst, const void *src, int in_rows, int in_cols,
                  int out_stride, int in_stride, int elem_size);

/* percentile5_float (reference: percentile.mako:115-140; percentile.py:193-209).
 * Per row, over columns [first_col, first_col + n_cols): out[0..4][row] =
 * min, max, sorted[(n-1)/4], sorted[3(n-1)/4], sorted[(n-1)/2] of |in|.
 * is_amplitude: in is float32 (positive); else complex64 and |.| is numpy's abs. */
int ksp_percentile5_float(int device, void *stream, const void *in, float *out, int rows,
                          int in_stride, int out_stride, int first_col, int n_cols,
                          int is_amplitude);

/* maskedsum_float (reference: maskedsum.mako:38-68; maskedsum.py:141-156).
 * out[col] = sum_row mask[row] * in[row][col]  (complex64 -> complex64), or
 * sum_row mask[row] * |in[row][col]| (-> float32) if use_amplitudes. */
int ksp_maskedsum_float(int device, void *stream, const void *in, const float *mask, void *out,
                        int in_stride, int n_rows, int n_cols, int use_amplitudes);

/* background_median_filter (reference: rfi/background_median_filter.mako:200-220;
 * launch rfi/device.py:311-325). in: [C][stride] complex64 or float32 amplitudes;
 * out: [C][stride] float32 deviations; flags: [C] (CHANNEL) or [C][flags_stride]
 * (FULL) uint8, any non-zero value masks the sample. width must be odd, <= 63.
 * csplit: number of channel segments per baseline (the reference's tunable of the same
 * name, rfi/device.py:212-252), 0 = let the launcher choose. */
int ksp_background_median_filter(int device, void *stream, const void *in, float *out,
                                 const uint8_t *flags, int channels, int baselines, int stride,
                                 int flags_stride, int width, int is_amplitude, int flags_mode,
                                 int csplit);

/* madnz_t (reference: rfi/madnz_t.mako:72-87; launch rfi/device.py:594-607).
 * in: [B][stride] float32; noise[b] = float32(1.4826 * median(|x| : x != 0)). */
int ksp_madnz_t(int device, void *stream, const float *in, float *noise, int channels,
                int baselines, int stride);

/* madnz (reference: rfi/madnz.mako:105-123; launch rfi/device.py:453-469).
 * Same statistic on channel-major data in: [C][stride]. */
int ksp_madnz(int device, void *stream, const float *in, float *noise, int channels,
              int baselines, int stride);

/* threshold_simple / threshold_simple_t (reference: rfi/threshold_simple.mako:27-40,
 * rfi/threshold_simple_t.mako:28-42; launch rfi/device.py:781-800).
 * flags = dev > n_sigma * noise[bl] ? flag_value : 0. rows x cols is the array
 * shape as stored: (C, B) if !transposed, (B, C) if transposed. */
int ksp_threshold_simple(int device, void *stream, const float *deviations, const float *noise,
                         uint8_t *flags, int rows, int cols, int stride, float n_sigma,
                         int flag_value, int transposed);

/* threshold_sum (reference: rfi/threshold_sum.mako:49-132; launch
 * rfi/device.py:968-987). deviations/flags: [B][stride]. Window k has size 2^k and
 * threshold float32(float32(n_sigma * noise[b]) * scales[k]) -- the float32 chain
 * numpy's host class follows when noise is float32 (rfi/host.py:235,252). Sums
 * are float64 over full windows only (rfi/host.py:239-242). scales is a HOST
 * pointer to n_windows floats (n_windows <= 8). vt: channels per thread (the
 * reference's tunable, rfi/device.py:868-887): 8, 16 or 32, 0 = let the launcher choose. */
int ksp_threshold_sum(int device, void *stream, const float *deviations, const float *noise,
                      uint8_t *flags, int channels, int baselines, int stride, float n_sigma,
                      const float *scales, int n_windows, int flag_value, int vt);

/* Fused single-pass flagger: the MI355X-native form of FlaggerDevice
 * (reference: rfi/device.py:1062-1166 composes background -> [transpose] ->
 * noise_est -> threshold -> [transpose]). One launch reads vis [C][vis_stride]
 * once and writes flags [C][flags_stride]; everything after the float32
 * amplitude is float64, so flags are bit-identical to rfi.host.FlaggerHost
 * (rfi/host.py:270-273). deviations (float32, [C][dev_stride]) and noise
 * (float32 [B]) are optional outputs (NULL to skip). scales64 is a HOST pointer
 * to n_windows doubles (falloff^-k, rfi/host.py:215). workspace: NULL, or 64 bytes of
 * device memory owned by the caller, zeroed once when allocated and used by one
 * launch at a time (scheduling counters: with it the last strips of a large array are
 * handed to whichever XCD is free; the kernel leaves it zeroed again). */
int ksp_flagger_fused(int device, void *stream, const void *vis, const uint8_t *in_flags,
                      uint8_t *flags, float *deviations, float *noise, int channels,
                      int baselines, int vis_stride, int in_flags_stride, int flags_stride,
                      int dev_stride, int width, int is_amplitude, int flags_mode,
                      int threshold_kind, double n_sigma, const double *scales64, int n_windows,
                      int flag_value, void *workspace);

/* Arms two events (from ksp_event_create) for the calling thread's NEXT
 * ksp_flagger_fused call: they are recorded immediately before and after the flagger
 * kernel itself, excluding the zero-fill of flags that precedes it (the counterpart of
 * the reference's per-kernel profiling, abc.py:405-432 / TuningCommandQueue). Pass
 * NULL, NULL to disarm. */
int ksp_flagger_fused_profile(void *start_event, void *stop_event);

/* Returns 1 if ksp_flagger_fused supports this configuration (else callers fall
 * back to the kernel-per-stage sequence): up to 4096 channels with any odd width 3 .. 31
 * and 1 .. 8 SumThreshold windows (rfi/device.py:840-852 takes any n_windows); 4097 ..
 * 12288 channels with width 13 and at most 4 windows. */
int ksp_flagger_fused_supported(int channels, int width, int n_windows);

/* Which kernels the calling thread's LAST ksp_flagger_fused call launched (no reference
 * counterpart: the tests use it to prove which path they exercised): 0 none, or a sum of
 * 1 = flagger_fused_kernel (strips of 4 baselines, up to 4096 channels),
 * 2 = flagger_long_kernel (4097-12288 channels),
 * 4 = flagger_ring_kernel (persistent, strips of 8 baselines, 4096 channels, complex input
 *     without input flags, no deviations output, at most 4 windows; chosen from about 4
 *     strips per compute unit on, see ksp_flagger_fused_ring_mode); 5 = ring kernel plus the
 *     4-baseline kernel for a remainder of fewer than 8 baselines. */
int ksp_flagger_fused_last_path(void);

/* Which launches of the calling thread take the persistent ring kernel where it applies
 * (no reference counterpart; tests and diagnostics): 0 = those with at least 4 strips of 8
 * baselines per compute unit (the default), 1 = all, -1 = none. The initial value comes from
 * KSP_FUSED_RING=1 / 0 in the environment if set. Returns the previous mode; any other
 * argument only queries. */
int ksp_flagger_fused_ring_mode(int mode);

/* Self-tests of the arithmetic building blocks (no reference counterpart; they exist
 * so that the test-suite can pin device arithmetic against IEEE / numpy results).
 * ksp_selftest_sqrt12: out[i] = the kernels' square root of the float32 with bit
 *   pattern 0x3f800000 + i (i.e. every float32 from 1.0 upwards; n <= 2^23 + 1 covers
 *   [1, 2]), to be compared with a correctly rounded sqrt.
 * ksp_selftest_abs: out[i] = the kernels' |re[i] + j im[i]| (numpy's complex64 abs,
 *   rfi/host.py:137). All pointers are device pointers. */
int ksp_selftest_sqrt12(int device, void *stream, float *out, int n);
int ksp_selftest_abs(int device, void *stream, const float *re, const float *im, float *out,
                     int n);

/* Self-tests of the rank / reduction library (csrc/rank.h, bitplane.h), the counterpart
 * of the reference's test kernels test/test_rank.mako:37-113 driven by
 * test/test_rank.py:67-213. One 256-thread workgroup each; device pointers.
 * ksp_selftest_rank: out[q] = number of the n <= 2048 non-negative floats in data that
 *   are strictly below q, for 0 <= q < m (rank.mako:31-105 `rank`; NaN never counts).
 * ksp_selftest_minmax: out[0] / out[1] = smallest / largest non-NaN value of the
 *   n <= 2048 floats, NaN if all are NaN (rank.mako:57-84).
 * ksp_selftest_median_non_zero: median of the non-zero values of n <= 16384
 *   non-negative floats (float32 mean of the middle two for an even count,
 *   rank.mako:253-267): out[0] by the workgroup search, out[1] by the wavefront
 *   bit-plane search when n <= 4096 (else the workgroup search again). */
int ksp_selftest_rank(int device, void *stream, const float *data, int *out, int n, int m);
int ksp_selftest_minmax(int device, void *stream, const float *data, float *out, int n);
int ksp_selftest_median_non_zero(int device, void *stream, const float *data, float *out,
                                 int n);

/* ---- run-time compilation and generic launch (reference abc.py:160-245 `compile`,
 * 406-432 `enqueue_kernel`; cuda.py:182-187, 442-459) ----
 * ksp_rtc_compile: compile HIP `source` with hiprtc for the device's architecture
 *   (options: n_options strings such as "-DNAME=1", "-I/dir") and load it; *module_out
 *   receives a module handle. The compiler's log (warnings, or the errors on failure)
 *   is copied to `log` (NUL-terminated, at most log_capacity bytes; may be NULL).
 * ksp_module_get_function: kernel `name` (an extern "C" __global__ function) of a module.
 * ksp_launch_function: launch with grid/block given as 3 unsigned each (in workgroups /
 *   threads) and kernel_params = array of pointers to the argument values, in order.
 * ksp_module_unload: free the module. */
int ksp_rtc_compile(int device, const char *source, const char *const *options, int n_options,
                    void **module_out, char *log, size_t log_capacity);
int ksp_module_get_function(int device, void *module, const char *name, void **function_out);
int ksp_module_unload(int device, void *module);
int ksp_launch_function(int device, void *stream, void *function, const unsigned *grid,
                        const unsigned *block, unsigned shared_bytes, void **kernel_params);

/* ---- FFT over hipFFT (reference fft.py:64-202 binds cuFFT the same way) ----
 * Transform types (the values hipFFT and cuFFT share). */
#define KSP_FFT_R2C 0x2a
#define KSP_FFT_C2R 0x2c
#define KSP_FFT_C2C 0x29
#define KSP_FFT_D2Z 0x6a
#define KSP_FFT_Z2D 0x6c
#define KSP_FFT_Z2Z 0x69
/* ksp_fft_plan_create: batched plan of `rank` (1..3) dimensions n[], unit strides, with
 *   padded (embedding) shapes and batch distances in elements on both sides
 *   (hipfftMakePlanMany64; reference fft.py:304-323). Automatic work-area allocation is
 *   off: *work_size bytes must be supplied to ksp_fft_exec.
 * ksp_fft_exec: run the plan on `stream`; inverse selects the direction of C2C / Z2Z
 *   (real transforms have only one). Unnormalised, like the reference. */
int ksp_fft_plan_create(int device, int rank, const long long *n, const long long *inembed,
                        long long idist, const long long *onembed, long long odist, int type,
                        long long batch, void **plan_out, size_t *work_size);
int ksp_fft_plan_destroy(int device, void *plan);
int ksp_fft_exec(int device, void *stream, void *plan, int type, void *src, void *dest,
                 void *work_area, int inverse);

#ifdef __cplusplus
}
#endif
#endif /* KATSDPSIGPROC_HIP_H */
