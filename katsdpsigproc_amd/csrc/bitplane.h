// Bit-plane helpers shared by the fused flagger's MAD and the standalone noise estimate.
#pragma once
#include "ksp_common.h"

// 32 x 32 bit-matrix transpose in registers: afterwards a[c] bit i = (old a[i]) bit c.
// The two coarse stages move whole bytes (v_perm_b32), the three fine ones are masked
// swaps (a shift and a bit-field insert per register).
__device__ __forceinline__ void transpose_bits32(unsigned (&a)[32])
{
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const unsigned x = a[k], y = a[k + 16];
        a[k] = __builtin_amdgcn_perm(y, x, 0x05040100u);
        a[k + 16] = __builtin_amdgcn_perm(y, x, 0x07060302u);
    }
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if (k & 8) continue;
        const unsigned x = a[k], y = a[k | 8];
        a[k] = __builtin_amdgcn_perm(y, x, 0x06020400u);
        a[k | 8] = __builtin_amdgcn_perm(y, x, 0x07030501u);
    }
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const int j = 4 >> s;
        const unsigned m = s == 0 ? 0x0f0f0f0fu : s == 1 ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            if (k & j) continue;
            // (the masked swap as two bit-field inserts: m << j == ~m for these masks)
            const unsigned x = a[k], y = a[k | j];
            a[k] = (x & m) | ((y << j) & ~m);
            a[k | j] = ((x >> j) & m) | (y & ~m);
        }
    }
}


// Bit-wise rank search on bit planes. The values of a row are spread over the lanes of
// one wavefront, 64 per lane, and transposed so that plane(bit, half) is a 32-bit mask
// over the lane's even (half 0) / odd (half 1) values with the bit CLEAR. The search
// walks from the top bit down keeping, per lane, which values still match the prefix
// (eq0/eq1) and, for the row, how many values lie below the prefix; each step decides
// one or two bits from population counts reduced over the wavefront with DPP.
// Afterwards `prefix` is the key of (0-based) rank `rank`, `below` the number of values
// strictly below it.
template <class PlaneFn>
struct PlaneSearch {
    unsigned eq0 = 0xffffffffu, eq1 = 0xffffffffu;  // even / odd values still matching the prefix
    unsigned prefix = 0;                            // bits decided so far
    int below = 0;                                  // values (whole row) below the prefix
    int rank;
    PlaneFn plane;  // plane(bit, half) -> INVERTED plane (bit clear), half 0 = even values

    __device__ __forceinline__ PlaneSearch(int rank_, PlaneFn plane_) : rank(rank_), plane(plane_) {}

    template <int HI, int LO>
    __device__ __forceinline__ void step2()
    {
        const unsigned a0 = eq0 & plane(HI, 0), a1 = eq1 & plane(HI, 1);
        const unsigned z00_0 = a0 & plane(LO, 0), z00_1 = a1 & plane(LO, 1);
        const unsigned z01_0 = a0 ^ z00_0, z01_1 = a1 ^ z00_1;
        const unsigned b0 = eq0 ^ a0, b1 = eq1 ^ a1;
        const unsigned z10_0 = b0 & plane(LO, 0), z10_1 = b1 & plane(LO, 1);
        const unsigned z11_0 = b0 ^ z10_0, z11_1 = b1 ^ z10_1;
        int s01 = (__popc(z00_0) + __popc(z00_1)) | ((__popc(z01_0) + __popc(z01_1)) << 16);
        int c10 = __popc(z10_0) + __popc(z10_1);
        ksp_wave_sum2_dpp(s01, c10);  // each field <= 4096
        const int n1 = below + (s01 & 0xffff), n2 = n1 + (s01 >> 16), n3 = n2 + c10;
        // the two bits are the number of thresholds with count(< threshold) <= rank
        // (n1 <= n2 <= n3); selected without branches
        static_assert(HI == LO + 1, "adjacent bits");
        const bool g1 = n1 <= rank, g2 = n2 <= rank, g3 = n3 <= rank;
        prefix |= (unsigned)((int)g1 + (int)g2 + (int)g3) << LO;
        below = g3 ? n3 : g2 ? n2 : g1 ? n1 : below;
        eq0 = g3 ? z11_0 : g2 ? z10_0 : g1 ? z01_0 : z00_0;
        eq1 = g3 ? z11_1 : g2 ? z10_1 : g1 ? z01_1 : z00_1;
    }

    template <int BIT>
    __device__ __forceinline__ void step1()
    {
        const unsigned z0 = eq0 & plane(BIT, 0), z1 = eq1 & plane(BIT, 1);
        const int c = below + ksp_wave_sum_dpp(__popc(z0) + __popc(z1));
        const bool take = c <= rank;
        prefix |= take ? (1u << BIT) : 0u;
        below = take ? c : below;
        eq0 = take ? (eq0 ^ z0) : z0;
        eq1 = take ? (eq1 ^ z1) : z1;
    }
};

