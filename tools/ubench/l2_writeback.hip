// Does a wavefront-private, constantly re-used block of global memory stay in the L2, or does every store go out to the fabric?
// Every wavefront owns BYTES of a buffer and rewrites then rereads it ROUNDS times (the access pattern of the deferred shadow rays' stacks,
// ky_device.hpp).  Run under `rocprofv3 --pmc WRITE_SIZE` (and FETCH_SIZE in a second pass): WRITE_SIZE ~ footprint => write-back;
// WRITE_SIZE ~ ROUNDS x footprint => the stores leave the L2 as they are issued.
//   hipcc --offload-arch=gfx950 -O2 -o build_variants/l2_writeback tools/ubench/l2_writeback.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k(float4* buf, int bytes_per_wave, int rounds, float* out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    float4* mine = buf + (size_t)wave * (bytes_per_wave / 16);
    const int n = bytes_per_wave / 16 / 64;   // float4 per lane
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < n; ++i) {
            const float4 v = make_float4(r + acc, lane, i, 1.f);
            if (MODE == 0) mine[i * 64 + lane] = v;
            else __builtin_nontemporal_store(v.x, &mine[i * 64 + lane].x), __builtin_nontemporal_store(v.y, &mine[i * 64 + lane].y), __builtin_nontemporal_store(v.z, &mine[i * 64 + lane].z), __builtin_nontemporal_store(v.w, &mine[i * 64 + lane].w);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int i = 0; i < n; ++i) acc += mine[i * 64 + (lane ^ 1)].x * 1e-9f;
    }
    if (acc == 12345.f) out[0] = acc;
}
int main(int argc, char** argv) {
    const int bytes = argc > 1 ? atoi(argv[1]) : 3072, rounds = argc > 2 ? atoi(argv[2]) : 2000, per_cu = argc > 3 ? atoi(argv[3]) : 6;
    const int blocks = 256 * per_cu, waves = blocks * 4;
    float4* buf; float* out;
    hipMalloc(&buf, (size_t)waves * bytes); hipMalloc(&out, 4);
    hipMemset(buf, 0, (size_t)waves * bytes);
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, bytes, rounds, out);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, bytes, rounds, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("mode %d (%s): %d waves x %d B = %.1f MB footprint, %d rounds: %.1f GB stored, %.3f ms\n", mode, mode ? "nontemporal stores" : "plain stores", waves, bytes,
               waves * (double)bytes / 1e6, rounds, waves * (double)bytes * rounds / 1e9, ms);
    }
    return 0;
}
