// micro-benchmark: does scalar work ride along for free?  Streams of v_fmac_f32 with 0, 1/2, 1 and 2 independent SALU instructions
// (s_add_u32 / s_and_b64 on registers nobody waits for) per VALU instruction, and SALU alone, at 1-8 waves per SIMD.
// The render kernel issues 0.53 SALU per VALU instruction (exec-mask bookkeeping of divergent code, scalar loads, loop control).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    float b = 1.0001f, c = 1e-6f;
    unsigned s0 = iters, s1 = 3;
    unsigned long long m0 = 0xffffffffull, m1 = 0xffull << iters;
    asm volatile("" : "+v"(b), "+v"(c), "+s"(s0), "+s"(s1), "+s"(m0), "+s"(m1));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE != 4) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 1 && (i & 1)) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
                if (MODE == 2 || MODE == 4) { if (i & 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc"); else asm volatile("s_and_b64 %0, %0, %1" : "+s"(m0) : "s"(m1) : "scc"); }
                if (MODE == 3) { asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc"); asm volatile("s_and_b64 %0, %0, %1" : "+s"(m0) : "s"(m1) : "scc"); }
                if (MODE == 6 && (i & 7) == 7) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");    // 1/8
                if (MODE == 7 && (i & 3) == 3) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");    // 1/4
                if (MODE == 8 && (i & 3) != 0) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");    // 3/4
                if (MODE == 5) { asm volatile("s_and_saveexec_b64 %0, %1\n s_or_b64 exec, exec, %0" : "+s"(m0) : "s"(m1) : "scc"); }   // the open / close of one divergent `if`
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)s0 + (float)(unsigned)m0;
}
template <int MODE>
void run(const char* name) {
    float* d; (void)hipMalloc(&d, 1 << 24);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int bpc : {1, 2, 4, 5, 6, 7, 8}) {
        const int iters = 4000, grid = 256 * bpc;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-46s waves/SIMD %d: %7.3f ms  %.2f ns per slot (one v_fmac + its scalar company) and SIMD\n", name, bpc, ms, ms * 1e6 / ((double)iters * 64 * bpc));
    }
    (void)hipFree(d);
}
int main() {
    run<0>("v_fmac alone"); run<1>("v_fmac + 1/2 SALU"); run<2>("v_fmac + 1 SALU"); run<3>("v_fmac + 2 SALU"); run<4>("1 SALU alone (no VALU)");
    run<5>("v_fmac + s_and_saveexec / s_or exec pair");
    run<6>("v_fmac + 1/8 SALU"); run<7>("v_fmac + 1/4 SALU"); run<8>("v_fmac + 3/4 SALU");
    return 0;
}
