// The fused flagger kernel and its launcher as templates over (R = channels per lane,
// WIDTH = median window): flagger_fused.hip instantiates width 13 (the reference
// script's, scripts/rfiflagtest.py:71) for R = 4, 16, 64, the flagger_fused_w*.hip files
// the other odd widths 3..31 for R = 64 -- one translation unit per few widths so that
// they compile in parallel.
#pragma once
#include <hip/hip_ext.h>

#include <atomic>

#include "fused_common.h"

#ifndef FUSED_DYN_SHIFT
#define FUSED_DYN_SHIFT 4  // the last 1 / 2^n of the strips are scheduled dynamically
#endif
#ifndef FUSED_DYN_OVER
#define FUSED_DYN_OVER 8  // quarters: dynamic workgroups launched per dynamic strip (8 = 2x)
#endif
#ifndef FUSED_STAGGER
#define FUSED_STAGGER 2  // x 8128 cycles: how long the second workgroup of a CU waits once
#endif

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
    // Workgroups are dealt to the 8 XCDs round robin, blockIdx % 8, an equal share each --
    // but the XCDs do not run equally fast (10 % between the fastest and the slowest in
    // a traced launch), so with a static strip per workgroup the fast ones idle at the
    // end. The last 1/16 of the strips are therefore handed out from a counter to twice
    // as many workgroups as there are strips: an XCD that gets through its static share
    // early takes more of them, a late one finds the counter exhausted and its surplus
    // workgroups leave at once.
    int strip;
    const bool dynamic = (int)blockIdx.x >= p.n_static;
    if (!dynamic) {
        strip = strip_of(blockIdx.x, p.n_static);
    } else {
        int *slot = (int *)lds;
        if (tid == 0) *slot = (int)atomicAdd(&p.work[0], 1u);
        __syncthreads();
        const int t = *slot;
        __syncthreads();  // everybody has read the slot before the loader reuses it
        strip = p.n_static + t;
        if (t >= p.n_dyn) {
            // nothing left: leave (the last dynamic workgroup to leave resets the counters)
            if (tid == 0 && atomicAdd(&p.work[1], 1u) == (unsigned)p.dyn_blocks - 1u) {
                p.work[0] = 0;
                p.work[1] = 0;
            }
            return;
        }
    }
    const int b0 = strip * FUSED_STRIP;

    // diagnostic time stamps (shader clock) of this wavefront's phases
    unsigned long long *trace = FUSED_DIAG_TRACE(p) ? FUSED_DIAG_TRACE(p) + ((size_t)blockIdx.x * FUSED_STRIP + wave) * 16 : nullptr;
    auto stamp = [&](int i) {
        if (trace != nullptr && lane == 0) trace[i] = FUSED_DIAG_CLOCK();
    };
    if (trace != nullptr && lane == 0)
        trace[7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    stamp(0);
#if FUSED_STAGGER > 0
    // Two workgroups share a CU and start together, and they then STAY in step -- both
    // loading (sharing the memory path), then both computing (sharing the vector ALUs)
    // -- because each slows the other equally. Holding one of the pair back once, by
    // about one LOAD phase, puts them in anti-phase for the rest of the launch (every
    // later workgroup inherits the slot, and with it the phase, of the one it
    // replaces): one's loads then run under the other's arithmetic. Which of the two
    // waits is told by the hardware's workgroup slot number on the CU. Speed only.
    if ((int)blockIdx.x < p.first_round &&
        ((__builtin_amdgcn_s_getreg(63492 /* HW_REG_HW_ID */) >> 16) & 1)) {
#pragma unroll
        for (int i = 0; i < FUSED_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    bool masked = true;  // may the strip hold samples that take no part (NaN in LDS)?
    if (!p.is_amplitude && b0 + FUSED_STRIP <= p.baselines) {
        if (p.flags_mode == KSP_FLAGS_NONE)
            masked = load_strip_fast<R, KSP_FLAGS_NONE>(p, lds, b0, tid);
        else if (p.flags_mode == KSP_FLAGS_CHANNEL)
            masked = load_strip_fast<R, KSP_FLAGS_CHANNEL>(p, lds, b0, tid);
        else
            masked = load_strip_fast<R, KSP_FLAGS_FULL>(p, lds, b0, tid);
    } else {
        load_strip<R>(p, lds, b0, tid);
    }
    stamp(1);
    const bool any_masked = __syncthreads_or(masked);
    stamp(2);
    if (FUSED_DIAG_STOP(p) == 1 || FUSED_DIAG_STOP(p) == 11) return;

    const int bl = b0 + wave;
    float *myrow = lds + wave * LY::ROW;
    double *list = (double *)(lds + LY::LDS_FLOATS) + wave * LY::LIST_DOUBLES;
    // amplitude of any channel of this baseline, for exact recomputation (LDS copy)
    auto fetch = [&](int c) -> float {
        return (c >= 0 && c < C) ? myrow[LY::index(c)] : __builtin_nanf("");
    };
    float dev[R];
    float dmax;
    int tiny = 0;  // deviations of +-2^-150 (SortedWindow::tiny)
    bool merged = false;
    if constexpr (R == 64 && WIDTH <= 13) {
        // clean strip over the whole band: the merging median (median_merge.h; its
        // W (W + 1) / 2 suffix-list registers rule out wider windows)
        if (!any_masked && C == 64 * R && FUSED_DIAG_STOP(p) != 21) {
            MergeMedian<R, WIDTH> mm;
            mm.template run_lane<LY::RUN - R>(myrow + lane * LY::RUN, lane, dev, dmax);
            tiny = mm.tiny;
            merged = true;
        }
    }
    if (!merged) median_phase<R, WIDTH>(myrow, lane, dev, dmax, &tiny);
    stamp(3);
    const FusedParams &pa = p;
    if (FUSED_DIAG_STOP(pa) == 2) {
        float acc = dmax;
#pragma unroll
        for (int j = 0; j < R; j++) acc += dev[j];
        if (acc == 12345.678f && pa.noise) pa.noise[0] = acc;  // keep the work alive
        return;
    }

    const double noise64 = mad_noise<R, WIDTH, LY::LIST_DOUBLES>(dev, lane, list, fetch, FUSED_DIAG_STOP(pa), trace, 0,
                                                                        nullptr, tiny);
    if (lane == 0 && pa.noise != nullptr && bl < pa.baselines) pa.noise[bl] = (float)noise64;
    stamp(4);
    if (FUSED_DIAG_STOP(pa) == 3 || FUSED_DIAG_STOP(pa) > 30) return;

    const unsigned long long fl =
        threshold_flags<R, WIDTH, (R == 64 ? 8 : 4)>(pa, dev, dmax, noise64, lane, C, fetch);
    stamp(5);
    if (FUSED_DIAG_STOP(pa) == 4) {
        if (fl == 0x123456789abcull && pa.noise) pa.noise[0] = 1.0f;
        return;
    }

    if (pa.deviations != nullptr) {
        // stage float32 deviations in this wavefront's LDS row (the amplitudes are no
        // longer needed), then write them as [channel][8 baselines]
#pragma unroll
        for (int j = 0; j < R; j++) myrow[lane * LY::RUN + j] = dev[j];
        __syncthreads();
        constexpr int LPR = FUSED_STRIP / 2;
        const int q = tid % LPR, r0 = tid / LPR;
        const int blq = b0 + 2 * q;
        for (int row = r0; row < C; row += FUSED_THREADS / LPR) {
            const int idx = LY::index(row);
            const float v0 = lds[(2 * q) * LY::ROW + idx];
            const float v1 = lds[(2 * q + 1) * LY::ROW + idx];
            float *dst = pa.deviations + (size_t)row * pa.dev_stride + blq;
            if (blq + 1 < pa.baselines && (pa.dev_stride & 1) == 0)
                *(float2 *)dst = make_float2(v0, v1);
            else {
                if (blq < pa.baselines) dst[0] = v0;
                if (blq + 1 < pa.baselines) dst[1] = v1;
            }
        }
    }
    write_flags(pa, fl, lane * R, bl, C);
    stamp(6);
    if (dynamic && tid == 0 && atomicAdd(&pa.work[1], 1u) == (unsigned)pa.dyn_blocks - 1u) {
        pa.work[0] = 0;
        pa.work[1] = 0;
    }
}

// =================================================================================
template <int R, int WIDTH>
inline int launch_fused(int device, hipStream_t s, const FusedParams &p, hipEvent_t ev0,
                        hipEvent_t ev1, bool zero_fill = true)
{
    using LY = FusedLayout<R>;
    const size_t lds_bytes = LY::LDS_BYTES;
    // all flags start at zero; the kernels only write the (rare) non-zero ones
    if (zero_fill)
        KSP_CHECK(hipMemsetAsync(p.flags, 0, (size_t)(p.channels - 1) * p.flags_stride + p.baselines,
                                 s));
    auto kern = flagger_fused_kernel<R, WIDTH>;
    // the opt-in to more than 64 KiB of dynamic LDS is per device (one context per
    // device in one process is a supported arrangement, reference doc/user/init.rst:4-6)
    static std::atomic<bool> attr_set[64];
    if (device < 0 || device >= 64 || !attr_set[device].load(std::memory_order_acquire)) {
        KSP_CHECK(hipFuncSetAttribute((const void *)kern,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        if (device >= 0 && device < 64) attr_set[device].store(true, std::memory_order_release);
#ifdef KSP_DIAG
        if (getenv("KSP_FUSED_DEBUG_OCC")) {
            int nb = -1;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kern, FUSED_THREADS, lds_bytes);
            fprintf(stderr, "flagger_fused_kernel<%d>: %d workgroups/CU, LDS %zu B\n", R, nb, lds_bytes);
        }
#endif
    }
#ifdef KSP_DIAG
    const char *trace_path = getenv("KSP_FUSED_DEBUG_TRACE");
    if (trace_path != nullptr) {
        // diagnostic run: collect per-wavefront phase time stamps and dump them
        FusedParams pt = p;
        const size_t n = (size_t)(p.n_static + p.dyn_blocks) * FUSED_STRIP * 16;
        KSP_CHECK(hipMalloc(&pt.trace, n * 8));
        KSP_CHECK(hipMemsetAsync(pt.trace, 0, n * 8, s));
        hipLaunchKernelGGL(kern, dim3(p.n_static + p.dyn_blocks), dim3(FUSED_THREADS), lds_bytes, s, pt);
        KSP_LAUNCH_CHECK();
        KSP_CHECK(hipStreamSynchronize(s));
        unsigned long long *host = (unsigned long long *)malloc(n * 8);
        KSP_CHECK(hipMemcpy(host, pt.trace, n * 8, hipMemcpyDeviceToHost));
        KSP_CHECK(hipFree(pt.trace));
        FILE *f = fopen(trace_path, "wb");
        if (f != nullptr) {
            fwrite(host, 8, n, f);
            fclose(f);
        }
        free(host);
        return 0;
    }
#endif
    // events armed by ksp_flagger_fused_profile time exactly this kernel (not the
    // zero-fill before it)
    if (ev0 != nullptr)
        hipExtLaunchKernelGGL(kern, dim3(p.n_static + p.dyn_blocks), dim3(FUSED_THREADS),
                              lds_bytes, s, ev0, ev1, 0, p);
    else
        hipLaunchKernelGGL(kern, dim3(p.n_static + p.dyn_blocks), dim3(FUSED_THREADS), lds_bytes,
                           s, p);
    KSP_LAUNCH_CHECK();
    return 0;
}

