/*
 * ky_measure.hpp -- measurement-only instrumentation of the render kernels.  NOT part of product builds: ky_device.hpp includes
 * this file only when one of KY_PROFILE_LANES / KY_PROFILE_CLOCKS / KY_MARKS is defined (tools/lane_probe.py, tools/phase_clocks.py,
 * tools/vgpr_peaks.py build such variants under build_variants/); otherwise KY_PROBE and KY_CLK expand to nothing.
 */
#pragma once
#include <hip/hip_runtime.h>

// lane-utilisation probes (debug builds with -DKY_PROFILE_LANES): slot k counts active lanes, slot k+16 counts visits
#ifdef KY_PROFILE_LANES
__device__ unsigned long long g_lane_probe[32];
#define KY_PROBE(k)                                                                                          \
    do {                                                                                                     \
        const unsigned long long m_ = __ballot(1);                                                           \
        if ((int)__lane_id() == __ffsll((long long)m_) - 1) {                                                \
            atomicAdd(&g_lane_probe[k], (unsigned long long)__popcll(m_));                                   \
            atomicAdd(&g_lane_probe[(k) + 16], 1ull);                                                        \
        }                                                                                                    \
    } while (0)
#else
#define KY_PROBE(k) do { } while (0)
#endif

// phase clocks (debug builds with -DKY_PROFILE_CLOCKS): KY_CLK(k) charges the wave's time since its previous mark to bucket k
#ifdef KY_PROFILE_CLOCKS
__device__ unsigned long long g_clk[16];
__device__ __noinline__ void ky_clk_mark(int k) {
    __shared__ unsigned long long last[16], acc[16][16];
    const int w = threadIdx.x >> 6;
    const unsigned long long m = __ballot(1);
    if ((int)__lane_id() == __ffsll((long long)m) - 1) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (k == -1) { for (int i = 0; i < 16; ++i) acc[w][i] = 0; }
        else if (k == -2) { for (int i = 0; i < 16; ++i) atomicAdd(&g_clk[i], acc[w][i]); }
        else acc[w][k] += t - last[w];
        last[w] = __builtin_amdgcn_s_memtime();
    }
}
#define KY_CLK(k) ky_clk_mark(k)
#elif defined(KY_MARKS)   // listing aid: a comment in the assembly at every phase boundary (tools/static_profile.py --marks)
#define KY_CLK(k) asm volatile("; KYMARK " #k)
#else
#define KY_CLK(k) do { } while (0)
#endif
