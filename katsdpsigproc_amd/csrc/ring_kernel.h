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

#ifndef RING_STOP
#define RING_STOP 0  // diagnostic builds: 1 = ring and |z| only, 2 = + median, 3 = + MAD
#endif
#define RING_THREADS 512
#define RING_STRIP 8
#define RING_NSLOT 32            // step slots in the ring
#ifndef RING_G
#define RING_G 4                 // steps per chunk
#endif
#ifndef RING_PRIO
#define RING_PRIO 0  // experiments with s_setprio between the two wavefronts of a SIMD
#endif
#ifndef RING_STAGGER
#define RING_STAGGER 0           // x 8128 cycles: every other workgroup of an XCD starts late (diagnostic)
#endif
#define RING_SLOT_BYTES 4096     // 64 rows x 64 B
#define RING_LIST_DOUBLES 256    // per wavefront (MAD candidate list)
#define RING_XCD_GROUP 8         // strips per XCD run

struct RingLayout {
    static constexpr size_t RING_BYTES = (size_t)RING_NSLOT * RING_SLOT_BYTES;
    static constexpr size_t LIST_BYTES = sizeof(double) * RING_LIST_DOUBLES * RING_STRIP;
    static constexpr size_t CTRL_BYTES = 64;
    static constexpr size_t LDS_BYTES = RING_BYTES + LIST_BYTES + CTRL_BYTES;
};

#ifdef RING_TRACE
// diagnostic builds (-DRING_TRACE): shader-clock stamps [workgroup][wavefront][strip][8],
// dumped by launch_ring to the file named by KSP_RING_TRACE (tools/trace_ring.py)
#define RING_TRACE_STRIPS 24
#define RING_TRACE_SLOTS 16
__device__ unsigned long long ring_trace_buf[256 * 8 * RING_TRACE_STRIPS * RING_TRACE_SLOTS];
#endif

// workspace words used by this kernel (the 4-baseline kernels use [0], [1])
#define RING_WORK_LIST 2   // [2 .. 9]: next ticket of XCD list x
#define RING_WORK_DONE 10  // workgroups finished

// ticket t of list x -> strip (runs of RING_XCD_GROUP strips per XCD); monotone in t
__device__ __forceinline__ int ring_strip_of(int x, int t)
{
    return ((t / RING_XCD_GROUP) * 8 + x) * RING_XCD_GROUP + (t % RING_XCD_GROUP);
}

// Next strip for a workgroup of XCD list `x`, from the counters: its own list first, then the
// others'. Tickets count from `t0` (what the static part of the schedule has used of every
// list). Returns -1 when every list is exhausted. One lane calls this.
__device__ __forceinline__ int ring_take(unsigned *work, int x, int n_strips, int t0)
{
    for (int k = 0; k < 8; k++) {
        const int xx = (x + k) & 7;
        // (a list that has been seen exhausted keeps counting up: harmless, reset at the end)
        const int t = t0 + (int)atomicAdd(&work[RING_WORK_LIST + xx], 1u);
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: so is all that follows from it)
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

    // This wavefront's pieces, in the order it requests them: steps par, par + 2, ... of a
    // strip, then the same of the next one -- a running source address that advances by two
    // rows per piece and jumps to the next strip when step `par` of that strip comes up.
    unsigned long long run_src = 0;
    const unsigned long long rb2 = 2 * (unsigned long long)row_bytes;
    const unsigned piece_base = __builtin_amdgcn_readfirstlane(lds_base + q * 1024);
    auto issue = [&](int slot) {  // the piece at run_src -> ring slot `slot` (+ par)
        // (opaque per use: the 32 destination addresses are otherwise kept in scalar
        // registers across the strip loop and spilled)
        unsigned pb = piece_base;
        asm volatile("" : "+s"(pb));
        const unsigned dst = pb + (unsigned)(slot + par) * RING_SLOT_BYTES;
        // Inline assembly: the compiler orders every later LDS read behind an LDS-DMA it
        // knows about with s_waitcnt vmcnt(0), which would serialise the ring. M0 (the LDS
        // destination) is restored for the compiler.
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(run_src), "s"(dst) : "memory");
        run_src += rb2;
    };
    auto strip_base = [&](int strip) -> const char * {
        return (const char *)p.vis + (size_t)strip * (RING_STRIP * 8);
    };

    // ---- schedule. The first p.n_static strips of a workgroup are fixed: strip k of
    // workgroup b is ticket k * (grid / 8) + b / 8 of list b % 8 (the workgroups b, b + 8, ...
    // share an XCD). The rest come from the per-list counters, so that an XCD that is ahead
    // takes more: a ticket costs a returning atomic, whose wait also waits for every LDS-DMA
    // request this wavefront has in flight -- hence only the tail of the launch pays it.
    // Strips are known two ahead (the next strip's address is needed half way through this
    // one).
    const int xcd = blockIdx.x & 7;
    const int per_list = (int)gridDim.x >> 3;
    const int t0 = p.n_static * per_list;  // tickets of every list used by the static part
    auto static_strip = [&](int k) { return ring_strip_of(xcd, k * per_list + ((int)blockIdx.x >> 3)); };
    if (tid == 0) {
        int s0 = 0 < p.n_static ? static_strip(0) : ring_take(p.work, xcd, p.n_strips, t0);
        int s1 = -1;
        if (s0 >= 0) s1 = 1 < p.n_static ? static_strip(1) : ring_take(p.work, xcd, p.n_strips, t0);
        ctrl[0] = s0;
        ctrl[1] = s1;
    }
    __syncthreads();
    // (read back through readfirstlane: strip numbers and everything derived from them live
    // in scalar registers)
    int cur = __builtin_amdgcn_readfirstlane(ctrl[0]), nxt = __builtin_amdgcn_readfirstlane(ctrl[1]);
    int k_strip = 0;  // strips this workgroup has started before `cur`
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
#if RING_STAGGER > 0
    if ((blockIdx.x >> 3) & 1) {
#pragma unroll
        for (int i = 0; i < RING_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    auto start_of = [&](int strip) -> unsigned long long {  // step `par` of a strip
        const unsigned long long a = (unsigned long long)strip_base(strip) + (unsigned long long)par * row_bytes;
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);  // (the builtin returns int)
    };
    run_src = start_of(cur);
#pragma unroll
    for (int j = 0; j < NSLOT; j += 2) issue(j);
    // Two register sets of raw samples: a chunk is read from the ring while the one before
    // it is worked on. Chunk 0 of the first strip:
    float2 zz[2][G];
    {
        constexpr int AH0 = (K - 1) * PER_CHUNK;
        __builtin_amdgcn_s_waitcnt(0x0070 | (AH0 & 15) | ((AH0 >> 4) << 14));
        __syncthreads();
#pragma unroll
        for (int g = 0; g < G; g++) {
            const unsigned long long w = *(lds_cu64 *)(lds3 + rd_off + g * RING_SLOT_BYTES);
            zz[0][g] = make_float2(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
        }
    }
    const float nanv = __builtin_nanf("");
    const float2 *visf = (const float2 *)p.vis;

#if RING_PRIO == 1
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    int mad_hint = -1;  // top bits of the previous strip's MAD key bin (fused_common.h, mad_noise)
    const int lane_id = lane;
#ifdef RING_TRACE
    int trace_it = 0;
#define RING_STAMP(i)                                                                              \
    do {                                                                                           \
        if (lane_id == 0 && trace_it < RING_TRACE_STRIPS && blockIdx.x < 256)                      \
            ring_trace_buf[((blockIdx.x * 8 + wave) * RING_TRACE_STRIPS + trace_it) * RING_TRACE_SLOTS + (i)] =   \
                __builtin_amdgcn_s_memtime();                                                      \
    } while (0)
#else
#define RING_STAMP(i) do { } while (0)
#endif
    for (;;) {
        // (the lane number behind an opaque copy per strip: the MAD's 64 lane masks and the
        // like are otherwise computed once, ahead of the loop, and spilled)
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        const int bl = cur * RING_STRIP + wave;
        const bool more = nxt >= 0;
        const unsigned long long start_nxt = start_of(more ? nxt : cur);
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
        RING_STAMP(0);
#ifdef RING_TRACE
        unsigned long long wait_dma = 0, wait_bar = 0;
        if (lane_id == 0 && trace_it < RING_TRACE_STRIPS && blockIdx.x < 256)
            ring_trace_buf[((blockIdx.x * 8 + wave) * RING_TRACE_STRIPS + trace_it) * RING_TRACE_SLOTS + 4] = __builtin_amdgcn_s_memrealtime();
#endif
        float amp[STEPS];
        unsigned umax = 0;
        // chunk c: make its 4 steps visible, refill the previous chunk's slots, |z|
        auto load_chunk = [&](auto c_) {
            constexpr int c = decltype(c_)::value;
            // Chunk c sits in registers (read during chunk c - 1). Before anything of it is
            // used: this wavefront's pieces of chunk c + 1 have landed -- it has issued those
            // of chunks c + 2 .. c + K - 1 after them (the last strip stops at its own last
            // chunk) -- and its reads of chunk c have returned; behind the barrier that holds
            // for every wavefront.
            constexpr int AH = (K - 2) * PER_CHUNK;
#ifdef RING_TRACE
            unsigned long long w0;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w0));
#endif
            if (more || (c + K - 1) * G + G <= STEPS)
                __builtin_amdgcn_s_waitcnt(0x0070 | (AH & 15) | ((AH >> 4) << 14));
            else
                __builtin_amdgcn_s_waitcnt(0x0070);
#ifdef RING_TRACE
            unsigned long long w1;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1));
#endif
#ifndef RING_NOSYNC  // (diagnostic: what do the barriers cost? results are then wrong)
            __syncthreads();
#endif
#ifdef RING_TRACE
            unsigned long long w2;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w2));
            wait_dma += w1 - w0;
            wait_bar += w2 - w1;
#endif
            // the slots of chunk c take the steps NSLOT further on (of the next strip from
            // chunk NCHUNK - K on)
            ksp_static_for<G / 2>([&](auto u_) {
                constexpr int j = c * G + 2 * decltype(u_)::value + NSLOT;
                if constexpr (j < STEPS) {
                    issue(j % NSLOT);
                } else if (more) {
                    if constexpr (j == STEPS) run_src = start_nxt;
                    issue(j % NSLOT);
                }
            });
            // chunk c + 1 into the other register set (chunk 0 of the next strip at the end)
            if (c + 1 < NCHUNK || more) {
                // (one address per chunk, opaque: or the compiler keeps 16 slot addresses in
                // registers for good, or adds the slot offset once per read)
                unsigned ro = rd_off + (((c + 1) * G) % NSLOT) * RING_SLOT_BYTES;
                asm volatile("" : "+v"(ro));
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const unsigned long long w = *(lds_cu64 *)(lds3 + ro + g * RING_SLOT_BYTES);
                    zz[(c + 1) & 1][g] = make_float2(__uint_as_float((unsigned)w), __uint_as_float((unsigned)(w >> 32)));
                }
            }
#if RING_PRIO == 2
            if ((c + (wave >> 2)) & 1)
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(0);
#endif
            const float2 (&z)[G] = zz[c & 1];
            // |z| (numpy's). The packed short division is computed unconditionally -- straight
            // line code that the scheduler can weave into the median arithmetic of the samples
            // before, where a branch around it would leave its dependent chains (and the wait
            // states between packed operations) on their own -- and replaced by the general form
            // in the rare chunk that holds a magnitude outside the ordinary range (zero,
            // subnormal, huge, infinite, NaN) in some lane.
            unsigned key = ~0u;
#pragma unroll
            for (int g = 0; g < G; g++) key &= ksp_abs_range_key(z[g].x, z[g].y);
#pragma unroll
            for (int g = 0; g < G; g += 2)
                ksp_abs_c64_inrange_x2(z[g].x, z[g].y, z[g + 1].x, z[g + 1].y, amp[c * G + g],
                                       amp[c * G + g + 1]);
            if (ksp_any((key & KSP_ABS_RANGE_BIT) == 0)) {
#pragma unroll
                for (int g = 0; g < G; g++) {
                    amp[c * G + g] = ksp_abs_c64(z[g].x, z[g].y);
                    umax = max(umax, __float_as_uint(amp[c * G + g]));
                }
            }
        };

        float dev[R];
        float dmax;
        // Which deviations are exact as they stand (fused_common.h, mad_noise HAVE_EXACT)? x and
        // m are multiples of the unit in the last place of whichever has the smaller exponent
        // e, and so is their difference D; it fits 24 bits, and the float32 subtraction does
        // not round, if and only if |D| < 2^(e + 1) -- and then d = D, otherwise |d| >= 2^(e + 1)
        // as well: d is exact exactly when its exponent is at most e, i.e. |d| <= the largest
        // float32 with exponent e. One bit per output, shifted into two accumulators in the
        // order the outputs appear. (NaN compares false: not exact.)
        unsigned ex_a = 0, ex_b = 0;
        int tiny = 0;  // deviations of +-2^-150 (SortedWindow::tiny)
        auto note_exact = [&](unsigned &acc, float x, float m, float d) {
            unsigned t;
            asm("v_min_u32 %1, %2, %3\n\tv_or_b32 %1, 0x7fffff, %1\n\t"
                "v_cmp_ge_f32 vcc, %1, |%4|\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
                : "+v"(acc), "=&v"(t)
                : "v"(__float_as_uint(x)), "v"(__float_as_uint(m)), "v"(d)
                : "vcc");
        };
#if RING_STOP == 1
        // diagnostic: ring + |z| only
        dmax = 0.f;
        ksp_static_for<NCHUNK>([&](auto c_) {
            load_chunk(c_);
            constexpr int c = decltype(c_)::value;
#pragma unroll
            for (int g = 0; g < G; g++) {
                dev[c * G + g] = amp[c * G + g];
                dmax += amp[c * G + g];
            }
        });
#else
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
                            if (last) {
                                const double dd = (double)xc - ((double)lo + (double)med) * 0.5;
                                d = (float)dd;
                                tiny += (d == 0.0f && dd != 0.0);
                            }
                        }
                        // (pinned: the optimiser otherwise sinks the whole median below the
                        // last chunk, next to the first use of the deviations)
                        asm volatile("" : "+v"(d));
                        if constexpr (jv < 32)
                            note_exact(ex_a, xc, med, d);
                        else
                            note_exact(ex_b, xc, med, d);
                        // (the running maximum takes two outputs at a time)
                        if constexpr (jv & 1)
                            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(dmax) : "v"(dev[j - 1]), "v"(d));
                        else if constexpr (jv == RV - 1)
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
                        if (first) {
                            const double dd = (double)xc - ((double)lo + (double)med) * 0.5;
                            d = (float)dd;
                            tiny += (d == 0.0f && dd != 0.0);
                        }
                    }
                    note_exact(ex_b, xc, med, d);
                    dmax = mm.vmax(dmax, d);
                    dev[j] = d;
                });
            }
        }
#endif
        RING_STAMP(1);
#ifdef RING_TRACE
        if (lane_id == 0 && trace_it < RING_TRACE_STRIPS && blockIdx.x < 256) {
            ring_trace_buf[((blockIdx.x * 8 + wave) * RING_TRACE_STRIPS + trace_it) * RING_TRACE_SLOTS + 5] = wait_dma;
            ring_trace_buf[((blockIdx.x * 8 + wave) * RING_TRACE_STRIPS + trace_it) * RING_TRACE_SLOTS + 6] = wait_bar;
        }
#endif
        // bit j of `exact`: output j. ex_a holds outputs H .. H + 31 (the first at bit 31),
        // ex_b outputs H + 32 .. 63, then 0 .. H - 1. Windows with an even number of samples
        // (band edges) average two samples in float64: never exact in this sense.
        unsigned long long exact;
        {
            const unsigned ra = __builtin_bitreverse32(ex_a), rb = __builtin_bitreverse32(ex_b);
            exact = ((unsigned long long)ra << H) |
                    ((unsigned long long)(rb & ((1u << (32 - H)) - 1u)) << (H + 32)) |
                    (unsigned long long)(rb >> (32 - H));
            unsigned long long first_odd = 0, last_odd = 0;
#pragma unroll
            for (int j = 0; j < H; j++)
                if ((H - j) & 1) first_odd |= 1ull << j;
#pragma unroll
            for (int j = R - H; j < R; j++)
                if ((j + H - R + 1) & 1) last_odd |= 1ull << j;
            if (lane == 0) exact &= ~first_odd;
            if (lane == 63) exact &= ~last_odd;
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
            tiny = 0;
            median_phase_src<R, W>([&](int i) { return a2[i + H]; }, dev, dmax, &tiny);
            exact = 0;
        }
        // next but one strip, if it comes from the counters: the ticket of the own list is in
        // flight during the MAD
        const bool ticket = more && k_strip + 2 >= p.n_static;
        unsigned t_own = 0;
        if (tid == 0 && ticket) t_own = atomicAdd(&p.work[RING_WORK_LIST + xcd], 1u);

#if RING_STOP == 1 || RING_STOP == 2
        // diagnostic: no MAD, no thresholds (keep the deviations alive)
        {
            float acc = dmax;
#pragma unroll
            for (int j = 0; j < R; j++) acc += dev[j];
            if (acc == 12345.678f && p.noise != nullptr) p.noise[bl] = acc;
        }
#else
        const double noise64 = mad_noise<R, W, RING_LIST_DOUBLES, true>(dev, lane, list, fetch, RING_STOP > 30 ? RING_STOP : 0,
#ifdef RING_TRACE
                                                                      (trace_it < RING_TRACE_STRIPS && blockIdx.x < 256)
                                                                          ? &ring_trace_buf[((blockIdx.x * 8 + wave) * RING_TRACE_STRIPS + trace_it) * RING_TRACE_SLOTS]
                                                                          : nullptr,
#else
                                                                      nullptr,
#endif
                                                                      exact, &mad_hint, tiny);
        if (lane == 0 && p.noise != nullptr) p.noise[bl] = (float)noise64;
        RING_STAMP(2);
#if RING_STOP == 3 || RING_STOP > 30
        if (noise64 == 12345.678 && p.noise != nullptr) p.noise[bl] = dmax;
#else
        const unsigned long long fl = threshold_flags<R, W>(p, dev, dmax, noise64, lane, C, fetch);
        write_flags(p, fl, lane * R, bl, C);
        RING_STAMP(3);
#endif
#endif

#ifdef RING_TRACE
        if (lane_id == 0 && trace_it < RING_TRACE_STRIPS && blockIdx.x < 256)
            ring_trace_buf[((blockIdx.x * 8 + wave) * RING_TRACE_STRIPS + trace_it) * RING_TRACE_SLOTS + 7] = (unsigned long long)cur + 1;
        trace_it++;
#endif
        if (!more) break;
        cur = nxt;
        if (!ticket) {
            nxt = static_strip(k_strip + 2);
#ifdef RING_END_SYNC
            __syncthreads();
#endif
        } else {
            if (tid == 0) {
                int take = ring_strip_of(xcd, t0 + (int)t_own);
                if (take >= p.n_strips) take = ring_take(p.work, (xcd + 1) & 7, p.n_strips, t0);
                ctrl[2] = take;
            }
            __syncthreads();
            nxt = __builtin_amdgcn_readfirstlane(ctrl[2]);
            __syncthreads();  // ctrl[2] is read before thread 0 can overwrite it
        }
        k_strip++;
    }
    finish();
}

// =================================================================================
// Can this launch take the ring kernel? (whole band of 4096 channels, complex input, no
// input flags, at least one whole strip, at most 4 SumThreshold windows, lane offsets within
// 32 bits)
inline bool ring_supported(const FusedParams &p, int width)
{
    return p.channels == 4096 && width % 2 == 1 && width >= 3 && width <= 13 && !p.is_amplitude &&
           p.flags_mode == KSP_FLAGS_NONE && p.deviations == nullptr && p.work != nullptr &&
           p.baselines >= RING_STRIP && (p.vis_stride % 2) == 0 &&
           (p.threshold_kind != KSP_THRESHOLD_SUM || p.n_windows <= 4) &&
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
    // static part of the schedule: all but the last two strips of a workgroup's even share,
    // as far as every list's tickets below n_static * grid / 8 are strips of the array
    p.n_static = 0;
    if (grid % 8 == 0 && p.n_strips / grid >= 3) {
        const int by_share = p.n_strips / grid - 2;
        const int by_range = (8 * (p.n_strips / 64)) / (grid / 8);
        p.n_static = by_share < by_range ? by_share : by_range;
    }
#ifndef RING_STATIC
    // (measured: handing out every strip by ticket is faster -- 0.37 against 0.44 ms at
    // 4096 x 32768 -- although each ticket's wait also waits for the requests in flight:
    // fixed shares keep the workgroups of an XCD in step, and they then queue on the same
    // rows; -DRING_STATIC builds the fixed part)
    p.n_static = 0;
#endif
#ifdef RING_TRACE
    {
        void *tb = nullptr;
        KSP_CHECK(hipGetSymbolAddress(&tb, HIP_SYMBOL(ring_trace_buf)));
        KSP_CHECK(hipMemsetAsync(tb, 0, sizeof(ring_trace_buf), s));
    }
#endif
    if (ev0 != nullptr)
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(RING_THREADS), RingLayout::LDS_BYTES, s, ev0, ev1,
                              0, p);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(RING_THREADS), RingLayout::LDS_BYTES, s, p);
    KSP_LAUNCH_CHECK();
#ifdef RING_TRACE
    if (const char *path = getenv("KSP_RING_TRACE")) {
        KSP_CHECK(hipStreamSynchronize(s));
        const size_t n = sizeof(ring_trace_buf);
        void *host = malloc(n);
        KSP_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(ring_trace_buf), n));
        FILE *f = fopen(path, "wb");
        if (f != nullptr) {
            fwrite(host, 1, n, f);
            fclose(f);
        }
        free(host);
    }
#endif
    return 0;
}
