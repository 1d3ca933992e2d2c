// micro-benchmark: what is the wave64 fp32 VALU issue ceiling of one gfx950 SIMD?
// Long straight-line streams (64 VALU per loop turn, 16 independent accumulators) so that loop overhead is 3 SALU per 64
// VALU; waves per SIMD 1..8; the shader clock is read with s_memtime so that cycles are real cycles, not ns x 2.4.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    float b = 1.0001f, c = 1e-6f;
    asm volatile("" : "+v"(b), "+v"(c));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));          // VOP3, 3 VGPR sources
                if (MODE == 1) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));         // VOP2, 3 VGPR sources
                if (MODE == 2) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                  // VOP2, 2 VGPR sources
                if (MODE == 3) asm volatile("v_mul_f32_e32 %0, 0x3f800347, %0" : "+v"(a[i]));                   // VOP2, literal + 1 VGPR
                if (MODE == 4) asm volatile("v_add_f32_e32 %0, 1.0, %0" : "+v"(a[i]));                          // VOP2, inline constant + 1 VGPR
                if (MODE == 5) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");  // 2 instr
                if (MODE == 6) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i]));                               // transcendental
                if (MODE == 7) { if (i & 3) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); else asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[i])); }  // 3:1 mix
                if (MODE == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                   // integer multiply
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, int per_instr) {
    float* d; unsigned long long* dc; hipMalloc(&d, 1 << 24); hipMalloc(&dc, 1 << 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc : {1, 2, 3, 4, 6, 8}) {
        const int iters = 4000, grid = 256 * bpc;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, dc, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, dc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2048]; hipMemcpy(h, dc, grid * 8, hipMemcpyDeviceToHost);
        double cyc = 0; for (int i = 0; i < grid; ++i) cyc += (double)h[i]; cyc /= grid;
        const double per_wave = (double)iters * 64 * per_instr;            // VALU instructions one wave issued
        const double per_simd = per_wave * bpc;                            // one wave of each block sits on each SIMD
        printf("%-34s waves/SIMD %d: %7.3f ms  clock %.2f GHz  %5.2f cycles per VALU per wave, %5.2f per SIMD (%.2f ns)\n", name, bpc, ms,
               cyc / (ms * 1e6), cyc / per_wave, cyc / per_simd, ms * 1e6 / per_simd);
    }
    hipFree(d); hipFree(dc);
}
int main() {
    run<0>("v_fma_f32 (VOP3, 3 vgpr)", 1); run<1>("v_fmac_f32_e32 (VOP2, 3 vgpr)", 1); run<2>("v_mul_f32_e32 (2 vgpr)", 1);
    run<3>("v_mul_f32_e32 (literal, 1 vgpr)", 1); run<4>("v_add_f32_e32 (inline, 1 vgpr)", 1); run<5>("v_cmp + v_cndmask", 2);
    run<6>("v_rcp_f32", 1); run<7>("3 fmac : 1 rcp", 1); run<8>("v_mul_lo_u32", 1);
    return 0;
}
