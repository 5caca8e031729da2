/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the RFI-flagging hot path.
 *
 * This file is a plain-C restatement of the algorithms in the reference's
 * NumPy/pandas host implementation (reference: src/katsdpsigproc/rfi/host.py).
 * It exists so that the HIP kernels can be checked bit-for-bit on the GPU box,
 * where the reference itself cannot travel. It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it. Parity is PINNED: tests/test_oracle_golden.py checks every function
 * here against golden vectors produced by importing the real reference
 * (tests/golden/make_golden.py) and against the reference's own known-answer
 * vectors (test/rfi/test_background.py:52-60, test/rfi/test_noise_est.py:35-50,
 * test/rfi/test_threshold.py:44-57, test/rfi/test_flagger.py:55-71).
 *
 * Numerics follow the reference exactly, including dtypes:
 *   - amplitude: numpy's complex64 abs, which on numpy 2.x is
 *     mx * sqrtf(fmaf(r, r, 1)), r = mn / mx (pinned by the golden "abs probe").
 *   - background: float32 amplitudes, float64 rolling median and deviations
 *     (host.py:133-151; pandas rolling(center=True, min_periods=1).median()).
 *   - noise: float64 median of non-zero |deviation| times 1.4826 (host.py:157-163).
 *   - thresholds: see comments at each function (host.py:181-183, 218-254).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAD_NORMAL 1.4826 /* reference: src/katsdpsigproc/rfi/__init__.py:31 */
#define MAX_WIDTH 255

/* numpy's np.abs for complex64 (host.py:137 calls np.abs(vis)). */
static inline float abs_c64(float re, float im)
{
    float ar = fabsf(re), ai = fabsf(im);
    if (isinf(ar) || isinf(ai))
        return INFINITY;
    float mx = ar > ai ? ar : ai;
    float mn = ar > ai ? ai : ar;
    if (isnan(ar) || isnan(ai))
        return NAN;
    if (mx == 0.0f)
        return 0.0f;
    float r = mn / mx;
    return mx * sqrtf(fmaf(r, r, 1.0f));
}

void oracle_abs_c64(const float *vis_ri, float *amp, int64_t n)
{
    for (int64_t i = 0; i < n; i++)
        amp[i] = abs_c64(vis_ri[2 * i], vis_ri[2 * i + 1]);
}

/* Median of n doubles by insertion sort; even n -> mean of the middle two
 * (pandas roll_median_c / numpy.median convention). n >= 1. */
static double median_small(double *v, int n)
{
    for (int i = 1; i < n; i++) {
        double x = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > x) {
            v[j + 1] = v[j];
            j--;
        }
        v[j + 1] = x;
    }
    if (n & 1)
        return v[n / 2];
    return (v[n / 2] + v[n / 2 - 1]) / 2.0;
}

/*
 * BackgroundMedianFilterHost.__call__ (host.py:133-151).
 *   vis: [C][B] complex64 (is_amplitude=0) or float32 (is_amplitude=1)
 *   flags_mode: 0 none, 1 per-channel [C], 2 full [C][B]; any non-zero byte
 *               masks the sample (host.py:143 astype(bool)).
 *   out: [C][B] float64 deviations, 0 where the sample is masked or NaN.
 * Window of output c is [c-H, c+H] clipped to the band; masked / NaN samples
 * are skipped (pandas min_periods=1).
 */
void oracle_background_median_filter(const void *vis, int is_amplitude,
                                     const uint8_t *flags, int flags_mode,
                                     int width, int channels, int baselines,
                                     double *out)
{
    const int H = width / 2;
#pragma omp parallel
    {
        float *amp = (float *)malloc(sizeof(float) * (size_t)channels);
        uint8_t *valid = (uint8_t *)malloc((size_t)channels);
#pragma omp for schedule(static)
        for (int b = 0; b < baselines; b++) {
            for (int c = 0; c < channels; c++) {
                size_t idx = (size_t)c * baselines + b;
                float a;
                if (is_amplitude)
                    a = ((const float *)vis)[idx];
                else
                    a = abs_c64(((const float *)vis)[2 * idx], ((const float *)vis)[2 * idx + 1]);
                int f = 0;
                if (flags_mode == 1)
                    f = flags[c] != 0;
                else if (flags_mode == 2)
                    f = flags[idx] != 0;
                amp[c] = a;
                valid[c] = !f && !isnan(a);
            }
            for (int c = 0; c < channels; c++) {
                size_t idx = (size_t)c * baselines + b;
                double win[MAX_WIDTH];
                int n = 0;
                int lo = c - H < 0 ? 0 : c - H;
                int hi = c + H >= channels ? channels - 1 : c + H;
                for (int k = lo; k <= hi; k++)
                    if (valid[k])
                        win[n++] = (double)amp[k];
                if (!valid[c] || n == 0)
                    out[idx] = 0.0;
                else
                    out[idx] = (double)amp[c] - median_small(win, n);
            }
        }
        free(amp);
        free(valid);
    }
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

static int cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/*
 * NoiseEstMADHost.__call__ (host.py:157-163) for float64 deviations [C][B]:
 * noise[b] = median(|d| : |d| > 0) * 1.4826, numpy.median semantics in float64
 * (even count -> (a + b) / 2). All-zero baseline -> NaN, as numpy.
 */
void oracle_noise_est_mad_f64(const double *dev, int channels, int baselines, double *noise)
{
#pragma omp parallel
    {
        double *tmp = (double *)malloc(sizeof(double) * (size_t)channels);
#pragma omp for schedule(static)
        for (int b = 0; b < baselines; b++) {
            int n = 0;
            for (int c = 0; c < channels; c++) {
                double a = fabs(dev[(size_t)c * baselines + b]);
                if (a > 0)
                    tmp[n++] = a;
            }
            if (n == 0) {
                noise[b] = NAN;
                continue;
            }
            qsort(tmp, (size_t)n, sizeof(double), cmp_double);
            double med = (n & 1) ? tmp[n / 2] : (tmp[n / 2 - 1] + tmp[n / 2]) / 2.0;
            noise[b] = med * MAD_NORMAL;
        }
        free(tmp);
    }
}

/*
 * Same, for float32 deviations: numpy.median of a float32 array stays in
 * float32 (even count -> float32(a + b) / 2), the result is stored in a
 * float64 array and scaled by 1.4826 in float64 (host.py:159-163).
 */
void oracle_noise_est_mad_f32(const float *dev, int channels, int baselines, double *noise)
{
#pragma omp parallel
    {
        float *tmp = (float *)malloc(sizeof(float) * (size_t)channels);
#pragma omp for schedule(static)
        for (int b = 0; b < baselines; b++) {
            int n = 0;
            for (int c = 0; c < channels; c++) {
                float a = fabsf(dev[(size_t)c * baselines + b]);
                if (a > 0)
                    tmp[n++] = a;
            }
            if (n == 0) {
                noise[b] = NAN;
                continue;
            }
            qsort(tmp, (size_t)n, sizeof(float), cmp_float);
            float med;
            if (n & 1)
                med = tmp[n / 2];
            else {
                volatile float s = tmp[n / 2 - 1] + tmp[n / 2];
                med = s / 2.0f;
            }
            noise[b] = (double)med * MAD_NORMAL;
        }
        free(tmp);
    }
}

/*
 * ThresholdSimpleHost.__call__ (host.py:181-183): flags = dev > n_sigma * noise.
 * noise_is_f32 selects numpy's promotion: a float32 noise array keeps
 * n_sigma * noise in float32 (python scalar is weak, NEP 50); float64 noise
 * gives a float64 product. dev is promoted to the threshold's type or wider.
 */
void oracle_threshold_simple(const void *dev, int dev_is_f32, const void *noise, int noise_is_f32,
                             int channels, int baselines, double n_sigma, int flag_value,
                             uint8_t *flags)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < channels; c++) {
        for (int b = 0; b < baselines; b++) {
            size_t idx = (size_t)c * baselines + b;
            double d = dev_is_f32 ? (double)((const float *)dev)[idx] : ((const double *)dev)[idx];
            double thr;
            if (noise_is_f32) {
                volatile float t = (float)n_sigma * ((const float *)noise)[b];
                thr = (double)t;
            } else
                thr = n_sigma * ((const double *)noise)[b];
            flags[idx] = (uint8_t)((d > thr) ? flag_value : 0);
        }
    }
}

/*
 * ThresholdSumHost (host.py:206-254), one baseline at a time
 * (apply_baseline, host.py:218-246):
 *   threshold1 = n_sigma * noise[b]   (float32 arithmetic if noise is float32,
 *                                      else float64; host.py:252)
 *   for k, window = 2**k, scale = falloff ** -k (python float, host.py:214-215):
 *     thr  = float32(threshold1 * scale)   (float32 product if threshold1 is
 *                                           float32: scale is first rounded to
 *                                           float32, NEP 50; host.py:235)
 *     d[flags] = thr                                        (host.py:237)
 *     sums = convolve(d, ones(window), 'valid'): float64, summed sequentially
 *            in ascending index order starting from 0.0      (host.py:239-240)
 *     hit = sums > float32(thr * window)                    (host.py:242)
 *     flags |= dilate(hit, window)                          (host.py:244-245)
 * dev is [C][B]; deviations keep their own dtype for the d[flags] = thr store
 * (exact either way because thr is a float32).
 */
void oracle_threshold_sum(const void *dev, int dev_is_f32, const void *noise, int noise_is_f32,
                          int channels, int baselines, double n_sigma, int n_windows,
                          double falloff, int flag_value, uint8_t *flags)
{
#pragma omp parallel
    {
        double *d = (double *)malloc(sizeof(double) * (size_t)channels);
        uint8_t *f = (uint8_t *)malloc((size_t)channels);
        uint8_t *hit = (uint8_t *)malloc((size_t)channels);
#pragma omp for schedule(static)
        for (int b = 0; b < baselines; b++) {
            for (int c = 0; c < channels; c++) {
                size_t idx = (size_t)c * baselines + b;
                d[c] = dev_is_f32 ? (double)((const float *)dev)[idx] : ((const double *)dev)[idx];
                f[c] = 0;
            }
            for (int k = 0; k < n_windows; k++) {
                int window = 1 << k;
                double scale = pow(falloff, -(double)k);
                float thr;
                if (noise_is_f32) {
                    volatile float t1 = (float)n_sigma * ((const float *)noise)[b];
                    volatile float t2 = t1 * (float)scale;
                    thr = t2;
                } else {
                    double t1 = n_sigma * ((const double *)noise)[b];
                    thr = (float)(t1 * scale);
                }
                volatile float limit_f = thr * (float)window;
                double limit = (double)limit_f;
                for (int c = 0; c < channels; c++)
                    if (f[c])
                        d[c] = (double)thr;
                int n_sums = channels - window + 1;
                for (int j = 0; j < n_sums; j++) {
                    double s = 0.0;
                    for (int i = 0; i < window; i++)
                        s += d[j + i];
                    hit[j] = s > limit;
                }
                for (int j = 0; j < n_sums; j++)
                    if (hit[j])
                        for (int i = 0; i < window; i++)
                            f[j + i] = 1;
            }
            for (int c = 0; c < channels; c++)
                flags[(size_t)c * baselines + b] = (uint8_t)(f[c] ? flag_value : 0);
        }
        free(d);
        free(f);
        free(hit);
    }
}

/*
 * FlaggerHost.__call__ (host.py:270-273) with BackgroundMedianFilterHost,
 * NoiseEstMADHost and ThresholdSumHost (threshold_kind=1) or
 * ThresholdSimpleHost (threshold_kind=0): everything after the float32
 * amplitude is float64, exactly as the host classes chain.
 * deviations_out / noise_out may be NULL.
 */
void oracle_flagger(const void *vis, int is_amplitude, const uint8_t *in_flags, int flags_mode,
                    int width, int channels, int baselines, int threshold_kind, double n_sigma,
                    int n_windows, double falloff, int flag_value, uint8_t *flags,
                    double *deviations_out, double *noise_out)
{
    size_t n = (size_t)channels * baselines;
    double *dev = deviations_out ? deviations_out : (double *)malloc(sizeof(double) * n);
    double *noise = noise_out ? noise_out : (double *)malloc(sizeof(double) * (size_t)baselines);
    oracle_background_median_filter(vis, is_amplitude, in_flags, flags_mode, width, channels,
                                    baselines, dev);
    oracle_noise_est_mad_f64(dev, channels, baselines, noise);
    if (threshold_kind == 1)
        oracle_threshold_sum(dev, 0, noise, 0, channels, baselines, n_sigma, n_windows, falloff,
                             flag_value, flags);
    else
        oracle_threshold_simple(dev, 0, noise, 0, channels, baselines, n_sigma, flag_value, flags);
    if (!deviations_out)
        free(dev);
    if (!noise_out)
        free(noise);
}

/*
 * Percentile5 oracle: the reference has no host class; its test states the
 * expected result as np.percentile(np.abs(x), [0,100,25,75,50], axis=1,
 * method="lower") (test/test_percentile.py:79-85), i.e. sorted[(n-1)*q/100]
 * with integer floor. src is [R][stride] float32 amplitudes; columns
 * [first, first+n). out is [5][R].
 */
void oracle_percentile5_f32(const float *src, int rows, int stride, int first, int n, float *out)
{
#pragma omp parallel
    {
        float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
#pragma omp for schedule(static)
        for (int r = 0; r < rows; r++) {
            memcpy(tmp, src + (size_t)r * stride + first, sizeof(float) * (size_t)n);
            qsort(tmp, (size_t)n, sizeof(float), cmp_float);
            out[0 * (size_t)rows + r] = tmp[0];
            out[1 * (size_t)rows + r] = tmp[n - 1];
            out[2 * (size_t)rows + r] = tmp[(n - 1) / 4];
            out[3 * (size_t)rows + r] = tmp[((n - 1) * 3) / 4];
            out[4 * (size_t)rows + r] = tmp[(n - 1) / 2];
        }
        free(tmp);
    }
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_threads(int n)
{
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
