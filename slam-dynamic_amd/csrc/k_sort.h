// Workgroup-level sort shared by the matcher, bag-of-words and detector post-processing kernels.
#pragma once
#include <hip/hip_runtime.h>

// ascending bitonic sort of n (power of two) 64-bit keys in LDS by one workgroup.  A thread owns compare-exchange PAIRS (no
// idle half), four per step with their eight LDS reads in flight together: a stage costs one LDS round trip per four pairs
// instead of one per key (the per-key form spent 160 us on 4096 keys, this one ~15).
__device__ __forceinline__ void sd_block_sort64(unsigned long long* keys, int n, int tid, int nthreads)
{
    const int half = n >> 1;
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q0 = tid; q0 < half; q0 += 4 * nthreads) {
                unsigned long long a[4], b[4];
                int tt[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + u * nthreads;
                    tt[u] = -1;
                    if (q < half) {
                        const int t = ((q & ~(j - 1)) << 1) | (q & (j - 1));      // pair q of this stage: t has bit j clear, partner t | j
                        tt[u] = t; a[u] = keys[t]; b[u] = keys[t | j];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = tt[u];
                    if (t >= 0) {
                        const bool up = (t & k) == 0;
                        if ((a[u] > b[u]) == up) { keys[t] = b[u]; keys[t | j] = a[u]; }
                    }
                }
            }
            __syncthreads();
        }
}
