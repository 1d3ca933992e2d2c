// what v_mul_legacy_f32 does on this chip, next to v_mul_f32: random operands, and the special cases (0 x inf, 0 x NaN, denormals)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
__global__ void k(const float* a, const float* b, float* leg, float* mul, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    asm("v_mul_legacy_f32_e64 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(b[i]));
    leg[i] = r;
    mul[i] = a[i] * b[i];
}
int main() {
    const int n = 1 << 16;
    std::vector<float> a(n), b(n), l(n), m(n);
    unsigned s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return s; };
    for (int i = 0; i < n; ++i) {
        unsigned x = rnd(), y = rnd();
        a[i] = ldexpf((float)(x >> 8) / 16777216.f + 1.f, (int)(rnd() % 40) - 20) * ((x & 1) ? -1.f : 1.f);
        b[i] = ldexpf((float)(y >> 8) / 16777216.f + 1.f, (int)(rnd() % 40) - 20) * ((y & 1) ? -1.f : 1.f);
    }
    const float sp[][2] = {{0.f, INFINITY}, {0.f, -INFINITY}, {0.f, NAN}, {5000.f, -INFINITY}, {0.f, 3.f}, {1e-40f, 2.f}, {1e-20f, 1e-20f}, {-0.f, INFINITY}, {2.f, NAN}};
    for (int i = 0; i < 9; ++i) { a[i] = sp[i][0]; b[i] = sp[i][1]; }
    float *da, *db, *dl, *dm;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dl, n * 4); hipMalloc(&dm, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dl, dm, n);
    hipMemcpy(l.data(), dl, n * 4, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 9; ++i) printf("%g x %g: legacy %g  mul %g\n", a[i], b[i], l[i], m[i]);
    int diff = 0; double worst = 0;
    for (int i = 9; i < n; ++i) if (memcmp(&l[i], &m[i], 4)) { ++diff; worst = fmax(worst, fabs((double)l[i] - m[i]) / fabs((double)m[i])); }
    printf("random operands: %d of %d products differ in their bits, worst relative difference %.3g\n", diff, n - 9, worst);
    return 0;
}
