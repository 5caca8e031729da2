// Bit-plane helpers shared by the fused flagger's MAD and the standalone noise estimate.
#pragma once
#include "ksp_common.h"

// 32 x 32 bit-matrix transpose in registers: afterwards a[c] bit i = (old a[i]) bit c.
// The two coarse stages move whole bytes (v_perm_b32), the three fine ones are the
// classic masked-swap butterflies.
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
            const unsigned t = ((a[k] >> j) ^ a[k | j]) & m;
            a[k | j] ^= t;
            a[k] ^= t << j;
        }
    }
}

