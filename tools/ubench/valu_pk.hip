// micro-benchmark: do the packed fp32 instructions (two fp32 operations per lane on a 64-bit register pair) issue at the rate of a
// plain VALU instruction on gfx950 when several waves share the SIMD?  Same harness as valu_peak.hip; a packed instruction is
// counted as ONE instruction, so "ns" below is per instruction: a v_pk_fma_f32 at the ns of a v_fma_f32 is twice the work per slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f2 a[16];
    for (int i = 0; i < 16; ++i) a[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    f2 b = {1.0001f, 0.9999f}, c = {1e-6f, 2e-6f};
    asm volatile("" : "+v"(b), "+v"(c));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b), "v"(c));   // second source broadcast from its low half
                if (MODE == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (MODE == 5) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if (MODE == 6) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x) : "v"(a[(i + 1) & 15].y));
                if (MODE == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));                  // reference: the unpacked VOP3
                if (MODE == 8) asm volatile("v_pk_mov_b32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            }
        }
    }
    f2 s = {0, 0};
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <int MODE>
void run(const char* name) {
    float* d; hipMalloc(&d, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bpc : {1, 2, 4, 6}) {
        const int iters = 4000, grid = 256 * bpc;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s waves/SIMD %d: %7.3f ms  %.2f ns per instruction and SIMD\n", name, bpc, ms, ms * 1e6 / ((double)iters * 64 * bpc));
    }
    hipFree(d);
}
int main() {
    run<7>("v_fma_f32 (reference)"); run<0>("v_pk_fma_f32"); run<1>("v_pk_mul_f32"); run<2>("v_pk_add_f32"); run<3>("v_pk_fma_f32 op_sel_hi broadcast");
    run<4>("v_max3_f32"); run<5>("v_med3_f32"); run<6>("v_mov_b32_dpp quad_perm"); run<8>("v_pk_mov_b32");
    return 0;
}
