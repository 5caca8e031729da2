// Persistent fused flagger for 4096-channel bands (flagger_ring.hip): one 512-thread
// workgroup per CU walks over strips of 8 baselines (64-byte row segments), wavefront w
// owns baseline w of the strip, lane l the run of 64 channels [64 l, 64 l + 64) -- as in
// flagger_fused_kernel.h -- but a baseline's amplitudes never form an image in LDS:
//
//   * the visibilities arrive by LDS-DMA (global_load_lds_dwordx4, one source address per
//     lane) in a RING of 32 step slots; step j = the 64 rows {64 l + j}, i.e. position j of
//     every lane's run, 64 B each: 4 KiB per slot, 128 KiB for the ring. The ring is
//     consumed in chunks of 4 steps; the slots of a chunk are refilled with the steps 32
//     further on (of this strip, then of the next one) one chunk later, so the requests for
//     the next strip are in flight while this one computes and no register holds them;
//   * a wavefront picks its baseline's sample out of a slot (ds_read_b64), turns it into
//     numpy's |z| and feeds it straight to the merging median (median_merge.h), which
//     consumes the run front to back: deviations appear in registers as the samples stream
//     in. The first H outputs of a lane need the END of the left neighbour's run: they are
//     produced last (band-edge conventions as in MergeMedian::run_src);
//   * the MAD and the thresholds then run from registers as before (fused_common.h). The few
//     exact float64 recomputations they ask for (exact_dev) read the visibilities again
//     from global memory -- 14 loads per candidate, all in flight together.
//
// LDS is only a staging ring here, so a CU holds 8 baselines in the registers of 8
// wavefronts while the next strip's loads are under way: the loads of one strip overlap
// the arithmetic of the previous one inside ONE workgroup, and a strip is 64 bytes wide
// instead of 32. Strips are handed out from per-XCD ticket counters (workspace): the 8
// strips of a 512-byte stretch of a row go to one XCD, and an XCD that runs out takes
// from the others.
//
// Speed notes (tools/ring_probe.hip): rows 64 apart must not be a multiple of ~16 KiB x
// 2^k apart in memory; the `vis` slot's row padding (FlaggerDeviceTemplate tuning
// `vis_pad`) takes care of that: 2.8 TB/s for the DMA pattern alone at stride 256 KiB,
// 5.1-5.4 with 8..48 elements of padding.
#pragma once
#include "fused_common.h"

#define RING_THREADS 512
#define RING_STRIP 8
#define RING_NSLOT 32            // step slots in the ring
#define RING_G 4                 // steps per chunk
#define RING_SLOT_BYTES 4096     // 64 rows x 64 B
#define RING_LIST_DOUBLES 256    // per wavefront (MAD candidate list)
#define RING_XCD_GROUP 8         // strips per XCD run

struct RingLayout {
    static constexpr size_t RING_BYTES = (size_t)RING_NSLOT * RING_SLOT_BYTES;
    static constexpr size_t LIST_BYTES = sizeof(double) * RING_LIST_DOUBLES * RING_STRIP;
    static constexpr size_t CTRL_BYTES = 64;
    static constexpr size_t LDS_BYTES = RING_BYTES + LIST_BYTES + CTRL_BYTES;
};

// workspace words used by this kernel (the 4-baseline kernels use [0], [1])
#define RING_WORK_LIST 2   // [2 .. 9]: next ticket of XCD list x
#define RING_WORK_DONE 10  // workgroups finished

// ticket t of list x -> strip (runs of RING_XCD_GROUP strips per XCD); monotone in t
__device__ __forceinline__ int ring_strip_of(int x, int t)
{
    return ((t / RING_XCD_GROUP) * 8 + x) * RING_XCD_GROUP + (t % RING_XCD_GROUP);
}

// Next strip for a workgroup of XCD list `x`: its own list first, then the others'.
// Returns -1 when every list is exhausted. One lane calls this.
__device__ __forceinline__ int ring_take(unsigned *work, int x, int n_strips)
{
    for (int k = 0; k < 8; k++) {
        const int xx = (x + k) & 7;
        // (a list that has been seen exhausted keeps counting up: harmless, reset at the end)
        const int t = (int)atomicAdd(&work[RING_WORK_LIST + xx], 1u);
        const int s = ring_strip_of(xx, t);
        if (s < n_strips) return s;
    }
    return -1;
}

// lane i <- lane i + 1 (lane 63 gets `edge`) / lane i <- lane i - 1 (lane 0 gets `edge`)
__device__ __forceinline__ float ring_from_next(float v, float edge, int lane)
{
    const float r = __shfl_down(v, 1, 64);
    return lane == 63 ? edge : r;
}
__device__ __forceinline__ float ring_from_prev(float v, float edge, int lane)
{
    const float r = __shfl_up(v, 1, 64);
    return lane == 0 ? edge : r;
}

// =================================================================================
template <int WIDTH>
__global__ __launch_bounds__(RING_THREADS) void flagger_ring_kernel(const FusedParams p)
{
    constexpr int R = 64, W = WIDTH, H = WIDTH / 2;
    constexpr int G = RING_G, NSLOT = RING_NSLOT, STEPS = 64, NCHUNK = STEPS / G;
    constexpr int K = NSLOT / G;       // chunks in the ring
    constexpr int PER_CHUNK = G / 2;   // DMA pieces a wavefront issues per chunk
    static_assert(W % 2 == 1 && W <= 13 && 2 * H <= G * 3, "merging median only");
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.channels;  // == 4096
    double *list = (double *)(lds + RingLayout::RING_BYTES) + wave * RING_LIST_DOUBLES;
    int *ctrl = (int *)(lds + RingLayout::RING_BYTES + RingLayout::LIST_BYTES);
    const unsigned lds_base = (unsigned)(size_t)(lds_void *)lds;

    // ---- DMA role: piece q (rows l = 16 q .. 16 q + 15) of the steps of parity `par`;
    // lane i fetches row 16 q + i / 4, 16-byte chunk (i % 4) ^ swizzle of its 64 bytes
    // (adjacent lanes stay inside one row segment; the swizzle spreads the readers' banks)
    const int q = wave & 3, par = wave >> 2;
    const int rl = lane >> 2;
    const int chunk16 = (lane & 3) ^ ((rl >> 2) & 3);
    const size_t row_bytes = (size_t)p.vis_stride * 8;
    const unsigned lane_off = (unsigned)((size_t)(16 * q + rl) * 64 * row_bytes + (size_t)chunk16 * 16);
    // ---- reader role: baseline = wave (16-byte pair pp, half hh)
    const int pp = wave >> 1, hh = wave & 1;
    const int l16 = lane & 15;
    const unsigned rd_off = (lane >> 4) * 1024 + l16 * 64 + ((pp ^ ((l16 >> 2) & 3)) * 16) + 8 * hh;
    typedef const __attribute__((address_space(3))) char lds_cchar;
    typedef const __attribute__((address_space(3))) unsigned long long lds_cu64;
    lds_cchar *lds3 = (lds_cchar *)(lds_void *)lds;

    auto issue = [&](const char *strip_base, int j) {  // this wavefront's piece of step j
        // (the row pitch behind an opaque copy: otherwise all 64 products j * pitch are
        // hoisted out of the strip loop and spilled)
        unsigned long long rb = row_bytes;
        asm volatile("" : "+s"(rb));
        const unsigned long long sa = (unsigned long long)strip_base + (unsigned long long)j * rb;  // (wave-uniform)
        const unsigned long long src =
            ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sa >> 32)) << 32) |
            (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)sa);  // (the builtin returns int)
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (j % NSLOT) * RING_SLOT_BYTES + q * 1024);
        // Inline assembly: the compiler orders every later LDS read behind an LDS-DMA it
        // knows about with s_waitcnt vmcnt(0), which would serialise the ring. M0 (the LDS
        // destination) is restored for the compiler.
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(src), "s"(dst) : "memory");
    };
    auto strip_base = [&](int strip) -> const char * {
        return (const char *)p.vis + (size_t)strip * (RING_STRIP * 8);
    };

    // ---- schedule: two strips ahead (the next strip's address is needed half way through
    // this one, and no vector-memory instruction other than the DMA may be issued while
    // the ring is being consumed: the waits below count them)
    const int xcd = blockIdx.x & 7;
    if (tid == 0) {
        ctrl[0] = ring_take(p.work, xcd, p.n_strips);
        ctrl[1] = ctrl[0] >= 0 ? ring_take(p.work, xcd, p.n_strips) : -1;
    }
    __syncthreads();
    // (read back through readfirstlane: strip numbers and everything derived from them live
    // in scalar registers)
    int cur = __builtin_amdgcn_readfirstlane(ctrl[0]), nxt = __builtin_amdgcn_readfirstlane(ctrl[1]);
    auto finish = [&]() {
        if (tid == 0 && atomicAdd(&p.work[RING_WORK_DONE], 1u) == gridDim.x - 1u) {
#pragma unroll
            for (int k = RING_WORK_LIST; k <= RING_WORK_DONE; k++) p.work[k] = 0;
        }
    };
    if (cur < 0) {
        finish();
        return;
    }
    {
        const char *b = strip_base(cur);
        // (steps NSLOT - G .. NSLOT - 1 are requested behind the first barrier, as for every
        // later strip)
        for (int j = par; j < NSLOT - G; j += 2) issue(b, j);
    }
    const float nanv = __builtin_nanf("");
    const float2 *visf = (const float2 *)p.vis;

    const int lane_id = lane;
    for (;;) {
        // (the lane number behind an opaque copy per strip: the MAD's 64 lane masks and the
        // like are otherwise computed once, ahead of the loop, and spilled)
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        const int bl = cur * RING_STRIP + wave;
        const bool more = nxt >= 0;
        const char *base_cur = strip_base(cur);
        const char *base_nxt = strip_base(more ? nxt : cur);
        // exact recomputations read the visibilities again (branch-free: the loads of a
        // window are all issued before the first is used)
        auto fetch = [&](int c) -> float {
            const bool inside = c >= 0 && c < C;
            const int cc = inside ? c : 0;
            const float2 v = visf[(size_t)cc * p.vis_stride + bl];
            float a = ksp_abs_c64(v.x, v.y);
            return inside ? a : nanv;
        };
        // (Vector-memory operations complete in issue order and s_waitcnt vmcnt(N) waits for
        // all but the N youngest, so the counted waits below only need a LOWER bound on what
        // was issued after the pieces they wait for: the stores, re-reads and tickets of the
        // phases in between make them wait longer, never shorter.)
        float amp[STEPS];
        unsigned umax = 0;
        // chunk c: make its 4 steps visible, refill the previous chunk's slots, |z|
        auto load_chunk = [&](auto c_) {
            constexpr int c = decltype(c_)::value;
            // this wavefront's pieces of chunk c have landed: it has issued those of chunks
            // c + 1 .. c + K - 2 after them (the last strip stops at its own last chunk)
            constexpr int AH = (K - 2) * PER_CHUNK;
            if (more || (c + K - 2) * G + G <= STEPS)
                __builtin_amdgcn_s_waitcnt(0x0070 | (AH & 15) | ((AH >> 4) << 14));
            else
                __builtin_amdgcn_s_waitcnt(0x0070);
            __syncthreads();
            // the slots of chunk c - 1 (every wavefront has read them) take the steps
            // NSLOT further on; chunk -1 = the previous strip's last chunk
#pragma unroll
            for (int g = par; g < G; g += 2) {
                constexpr int cm = (c + NCHUNK - 1) % NCHUNK;
                const int j = cm * G + g + NSLOT - (c == 0 ? STEPS : 0);
                if (j < STEPS)
                    issue(base_cur, j);
                else if (more)
                    issue(base_nxt, j - STEPS);
            }
            float2 z[G];
            unsigned ro = rd_off;
            asm volatile("" : "+v"(ro));  // (opaque: or 16 slot addresses are kept in registers for good)
#pragma unroll
            for (int g = 0; g < G; g++) {
                const unsigned long long w = *(lds_cu64 *)(lds3 + ro + ((c * G + g) % NSLOT) * RING_SLOT_BYTES);
                z[g] = make_float2(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
            }
            // |z| (numpy's; packed short division when every magnitude is ordinary)
            unsigned key = ~0u;
#pragma unroll
            for (int g = 0; g < G; g++) key &= ksp_abs_range_key(z[g].x, z[g].y);
            if (!ksp_any((key & KSP_ABS_RANGE_BIT) == 0)) {
#pragma unroll
                for (int g = 0; g < G; g += 2)
                    ksp_abs_c64_inrange_x2(z[g].x, z[g].y, z[g + 1].x, z[g + 1].y, amp[c * G + g],
                                           amp[c * G + g + 1]);
            } else {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    amp[c * G + g] = ksp_abs_c64(z[g].x, z[g].y);
                    umax = max(umax, __float_as_uint(amp[c * G + g]));
                }
            }
        };

        float dev[R];
        float dmax;
        {
            MergeMedian<R, W> mm;
            mm.pinf = __builtin_inff();
            mm.ninf = -__builtin_inff();
            asm volatile("" : "+v"(mm.pinf), "+v"(mm.ninf));
            const float pinf = mm.pinf, ninf = mm.ninf;
            const bool first = lane == 0, last = lane == 63;
            dmax = ninf;
            float rh[H];  // right halo: positions 64 .. 64 + H - 1
            // position i of the lane's run, 0 <= i < 64 + H, asked for in increasing order
            auto X = [&](auto i_) -> float {
                constexpr int i = decltype(i_)::value;
                if constexpr (i < STEPS) {
                    if constexpr (i % G == 0) load_chunk(std::integral_constant<int, i / G>{});
                    return amp[i];
                } else {
                    if constexpr (i == STEPS) {
#pragma unroll
                        for (int k = 0; k < H; k++)
                            rh[k] = ring_from_next(amp[k], (k & 1) ? ninf : pinf, lane);
                    }
                    return rh[i - STEPS];
                }
            };
            // ---- outputs H .. 63: MergeMedian::run_src with the origin moved to position H
            constexpr int RV = R - H;
            constexpr int STAGES = (RV + W - 1) / W;
            using MM = MergeMedian<R, W>;
            float cur_b[W];
            ksp_static_for<W>([&](auto k_) { cur_b[decltype(k_)::value] = X(k_); });
            ksp_static_for<STAGES>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                float S[MM::S_SIZE];
                S[MM::off(W - 1)] = cur_b[W - 1];
                ksp_static_for<W - 1>([&](auto u_) {
                    constexpr int t = W - 2 - decltype(u_)::value;
                    mm.template insert<W - 1 - t>(&S[MM::off(t + 1)], cur_b[t], &S[MM::off(t)]);
                });
                float P[W], nxt_b[W];
                ksp_static_for<W>([&](auto t_) {
                    constexpr int t = decltype(t_)::value;
                    constexpr int jv = m * W + t;  // output, counted from position H
                    if constexpr (jv < RV) {
                        constexpr int j = jv + H;
                        if constexpr (t >= 1) {
                            nxt_b[t - 1] = X(std::integral_constant<int, (m + 1) * W + t - 1>{});
                            if constexpr (t == 1)
                                P[0] = nxt_b[0];
                            else
                                mm.template insert<t - 1>(P, nxt_b[t - 1], P);
                        }
                        const float med = mm.template rank<W - t, t, H>(&S[MM::off(t)], P);
                        const float xc = (t + H < W) ? cur_b[t + H < W ? t + H : 0] : nxt_b[t + H >= W ? t + H - W : 0];
                        float d = xc - med;
                        // windows that reach beyond the band by an odd number of samples
                        // (last lane only) hold an even number of valid ones
                        constexpr bool right_odd = j + H >= R && ((j + H - R + 1) & 1);
                        if constexpr (right_odd) {
                            const float lo = mm.template rank<W - t, t, H - 1>(&S[MM::off(t)], P);
                            if (last) d = (float)((double)xc - ((double)lo + (double)med) * 0.5);
                        }
                        // (pinned: the optimiser otherwise sinks the whole median below the
                        // last chunk, next to the first use of the deviations)
                        asm volatile("" : "+v"(d));
                        dmax = mm.vmax(dmax, d);
                        dev[j] = d;
                    }
                });
                if constexpr (m + 1 < STAGES) {
                    ksp_static_for<W>([&](auto k_) {
                        constexpr int k = decltype(k_)::value;
                        constexpr int i = (m + 1) * W + k;
                        constexpr bool have = (m * W + k + 1 < RV) && (k + 1 < W);
                        if constexpr (!have) {
                            if constexpr (i < R + H)
                                nxt_b[k] = X(std::integral_constant<int, i>{});
                            else
                                nxt_b[k] = pinf;
                        }
                    });
#pragma unroll
                    for (int k = 0; k < W; k++) cur_b[k] = nxt_b[k];
                }
            });
            // ---- outputs 0 .. H - 1: the window is a suffix of the left neighbour's last H
            // samples plus a prefix of the lane's own first 2 H
            if constexpr (H >= 1) {
                float L[H];
#pragma unroll
                for (int k = 0; k < H; k++)
                    L[k] = ring_from_prev(amp[STEPS - H + k], ((H - k) & 1) ? pinf : ninf, lane);
                // suffix lists of L: SL[soff(j)] .. = sorted L[j .. H-1]
                constexpr auto soff = [](int j) { return j * H - j * (j - 1) / 2; };
                float SL[H * (H + 1) / 2];
                SL[soff(H - 1)] = L[H - 1];
                ksp_static_for<H - 1>([&](auto u_) {
                    constexpr int j = H - 2 - decltype(u_)::value;
                    mm.template insert<H - 1 - j>(&SL[soff(j + 1)], L[j], &SL[soff(j)]);
                });
                // prefix list of the own samples, grown to 2 H
                float PL[2 * H];
                PL[0] = amp[0];
                ksp_static_for<H>([&](auto k_) {
                    constexpr int k = decltype(k_)::value + 1;  // insert amp[k] into PL[0 .. k)
                    mm.template insert<k>(PL, amp[k], PL);
                });
                ksp_static_for<H>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    if constexpr (j >= 1) mm.template insert<j + H>(PL, amp[j + H], PL);
                    const float med = mm.template rank<H - j, j + H + 1, H>(&SL[soff(j)], PL);
                    const float xc = amp[j];
                    float d = xc - med;
                    if constexpr ((H - j) & 1) {
                        const float lo = mm.template rank<H - j, j + H + 1, H - 1>(&SL[soff(j)], PL);
                        if (first) d = (float)((double)xc - ((double)lo + (double)med) * 0.5);
                    }
                    dmax = mm.vmax(dmax, d);
                    dev[j] = d;
                });
            }
        }
        // From here on the ring is not touched; the requests for the next strip stay in
        // flight. A baseline with a NaN amplitude (NaN or infinite input) takes the general
        // sorted-window median on amplitudes read again from global memory.
        if (ksp_any(umax > 0x7f800000u)) {
            float a2[R + 2 * H];
            int c_first = lane * R - H;
            asm volatile("" : "+v"(c_first));  // (opaque: nothing of this path is hoisted out of the loop)
#pragma unroll
            for (int i = 0; i < R + 2 * H; i++) a2[i] = fetch(c_first + i);
            median_phase_src<R, W>([&](int i) { return a2[i + H]; }, dev, dmax);
        }
        // next but one strip: the ticket of the own list is in flight during the MAD
        unsigned t_own = 0;
        if (tid == 0 && more) t_own = atomicAdd(&p.work[RING_WORK_LIST + xcd], 1u);

        const double noise64 = mad_noise<R, W, RING_LIST_DOUBLES>(dev, lane, list, fetch);
        if (lane == 0 && p.noise != nullptr) p.noise[bl] = (float)noise64;
        const unsigned long long fl = threshold_flags<R, W>(p, dev, dmax, noise64, lane, C, fetch);
        write_flags(p, fl, lane * R, bl, C);

        if (tid == 0) {
            int take = -1;
            if (more) {
                take = ring_strip_of(xcd, (int)t_own);
                if (take >= p.n_strips) take = ring_take(p.work, (xcd + 1) & 7, p.n_strips);
            }
            ctrl[2] = take;
        }
        if (!more) break;
        __syncthreads();  // (also keeps a fast wavefront out of the ring of a slow one's strip)
        cur = nxt;
        nxt = __builtin_amdgcn_readfirstlane(ctrl[2]);
        __syncthreads();  // ctrl[2] is read before thread 0 can overwrite it
    }
    finish();
}

// =================================================================================
// Can this launch take the ring kernel? (whole band of 4096 channels, complex input, no
// input flags, at least one whole strip, lane offsets within 32 bits)
inline bool ring_supported(const FusedParams &p, int width)
{
    return p.channels == 4096 && width % 2 == 1 && width >= 3 && width <= 13 && !p.is_amplitude &&
           p.flags_mode == KSP_FLAGS_NONE && p.deviations == nullptr && p.work != nullptr &&
           p.baselines >= RING_STRIP && (p.vis_stride % 2) == 0 &&
           (size_t)p.vis_stride * 8 * 4096 < 0x7fffff00ull;
}

template <int WIDTH>
inline int launch_ring(int device, hipStream_t s, const FusedParams &p_in, int n_cu, hipEvent_t ev0,
                       hipEvent_t ev1)
{
    FusedParams p = p_in;
    p.n_strips = p.baselines / RING_STRIP;  // whole strips only (the caller does the rest)
    auto kern = flagger_ring_kernel<WIDTH>;
    static std::atomic<bool> attr_set[64];
    if (device < 0 || device >= 64 || !attr_set[device].load(std::memory_order_acquire)) {
        KSP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)RingLayout::LDS_BYTES));
        if (device >= 0 && device < 64) attr_set[device].store(true, std::memory_order_release);
    }
    const int grid = p.n_strips < n_cu ? p.n_strips : n_cu;
    if (ev0 != nullptr)
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(RING_THREADS), RingLayout::LDS_BYTES, s, ev0, ev1,
                              0, p);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(RING_THREADS), RingLayout::LDS_BYTES, s, p);
    KSP_LAUNCH_CHECK();
    return 0;
}
