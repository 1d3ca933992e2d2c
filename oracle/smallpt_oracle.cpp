/*
 * smallpt_oracle.cpp -- CPU restatement of smallpt2pbrt/smallpt.cpp (the literal scene of BASELINE.json configs[0]):
 * Vec / Ray / Sphere::intersect (10-38), the 9-sphere scene (42-52), intersect() (57-61), the RECURSIVE radiance()
 * (63-89) and main()'s loop nest (91-118), all in double precision.
 *
 * TEST INFRASTRUCTURE ONLY, like ky_oracle.cpp: used by tests/ (and nothing under ky_amd/).  It checks
 * ky_amd/csrc/ky_smallpt.hpp, which unrolls the recursion onto a stack; this file keeps the recursion.
 *
 * PARITY UNPINNED: smallpt.cpp / smallpt_milo.cpp do not compile in this image without stand-ins (`errno_t`, `fopen_s`,
 * `__forceinline` are MSVC-only; smallpt.cpp:120, smallpt_milo.cpp:12) and the reference holds no image or vector of
 * this scene.  Two random-number modes:
 *   rng_mode 0  one splitmix64 stream per (seed, pixel, subpixel, sample) -- what the HIP path uses; per-sample comparable;
 *   rng_mode 1  smallpt's own scheme: erand48 (the 48-bit LCG a = 0x5DEECE66D, c = 0xB of erand48.h:33-75), seeded
 *               {0, 0, y*y*y} per image row and walked sequentially along the row (smallpt.cpp:98) -- for statistical
 *               cross-checks of mode 0 on the CPU.
 * Where the recursion follows both rays at the glass sphere (88), C++ leaves the order of the two calls open; both
 * modes evaluate the reflected ray first.
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/kyhip.h"

namespace {

struct Vec {
    double x, y, z;
    Vec(double x_ = 0, double y_ = 0, double z_ = 0) : x(x_), y(y_), z(z_) {}
    Vec operator+(const Vec& b) const { return Vec(x + b.x, y + b.y, z + b.z); }
    Vec operator-(const Vec& b) const { return Vec(x - b.x, y - b.y, z - b.z); }
    Vec operator*(double b) const { return Vec(x * b, y * b, z * b); }
    Vec mult(const Vec& b) const { return Vec(x * b.x, y * b.y, z * b.z); }
    Vec& norm() { return *this = *this * (1 / std::sqrt(x * x + y * y + z * z)); }
    double dot(const Vec& b) const { return x * b.x + y * b.y + z * b.z; }
    Vec cross(const Vec& b) const { return Vec(y * b.z - z * b.y, z * b.x - x * b.z, x * b.y - y * b.x); }   // operator%, 19
};

struct Rng {
    int mode;
    uint64_t s;            // mode 0: splitmix64 state
    unsigned short xi[3];  // mode 1: erand48 state
};

uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void rng_start(Rng& r, uint32_t seed, uint32_t subpixel_index, uint32_t sample) {
    r.mode = 0;
    r.s = mix64(((uint64_t)seed << 32) ^ (uint64_t)subpixel_index) + (uint64_t)sample * 0xD1B54A32D192ED03ull;
    r.s = mix64(r.s);
}
double rnd(Rng& r) {
    if (r.mode == 0) {
        r.s += 0x9E3779B97F4A7C15ull;
        return (double)(mix64(r.s) >> 11) * (1.0 / 9007199254740992.0);
    }
    // erand48 (erand48.h:47-75): X' = (0x5DEECE66D * X + 0xB) mod 2^48, value = X' / 2^48
    uint64_t x = (uint64_t)r.xi[0] | ((uint64_t)r.xi[1] << 16) | ((uint64_t)r.xi[2] << 32);
    x = (x * 0x5DEECE66Dull + 0xBull) & 0xFFFFFFFFFFFFull;
    r.xi[0] = (unsigned short)x; r.xi[1] = (unsigned short)(x >> 16); r.xi[2] = (unsigned short)(x >> 32);
    return std::ldexp((double)r.xi[0], -48) + std::ldexp((double)r.xi[1], -32) + std::ldexp((double)r.xi[2], -16);
}

struct Scene {
    const ky_smallpt_sphere* s;
    int n, max_depth;
};

double sphere_intersect(const ky_smallpt_sphere& s, const Vec& o, const Vec& d) {   // 33-38
    Vec op = Vec(s.p[0], s.p[1], s.p[2]) - o;
    double t, eps = 1e-4, b = op.dot(d), det = b * b - op.dot(op) + s.rad * s.rad;
    if (det < 0) return 0;
    det = std::sqrt(det);
    return (t = b - det) > eps ? t : ((t = b + det) > eps ? t : 0);
}

bool intersect(const Scene& sc, const Vec& o, const Vec& d, double& t, int& id) {   // 57-61
    double dd, inf = t = 1e20;
    for (int i = sc.n; i--;)
        if ((dd = sphere_intersect(sc.s[i], o, d)) && dd < t) { t = dd; id = i; }
    return t < inf;
}

Vec radiance(const Scene& sc, const Vec& ro, const Vec& rd, int depth, Rng& rng) {   // 63-89
    double t;
    int id = 0;
    if (!intersect(sc, ro, rd, t, id)) return Vec();
    const ky_smallpt_sphere& obj = sc.s[id];
    const Vec e(obj.e[0], obj.e[1], obj.e[2]);
    if (depth > sc.max_depth) return e;
    Vec x = ro + rd * t, n = (x - Vec(obj.p[0], obj.p[1], obj.p[2])).norm(), nl = n.dot(rd) < 0 ? n : n * -1, f(obj.c[0], obj.c[1], obj.c[2]);
    double p = f.x > f.y && f.x > f.z ? f.x : f.y > f.z ? f.y : f.z;
    if (++depth > 5) {
        if (rnd(rng) < p) f = f * (1 / p);
        else return e;
    }
    if (obj.refl == KY_SP_DIFF) {
        double r1 = 2 * 3.141592653589793238462643 * rnd(rng), r2 = rnd(rng), r2s = std::sqrt(r2);
        Vec w = nl, u = ((std::fabs(w.x) > .1 ? Vec(0, 1) : Vec(1)).cross(w)).norm(), v = w.cross(u);
        Vec d = (u * std::cos(r1) * r2s + v * std::sin(r1) * r2s + w * std::sqrt(1 - r2)).norm();
        return e + f.mult(radiance(sc, x, d, depth, rng));
    } else if (obj.refl == KY_SP_SPEC) {
        return e + f.mult(radiance(sc, x, rd - n * 2 * n.dot(rd), depth, rng));
    }
    Vec refl_d = rd - n * 2 * n.dot(rd);
    bool into = n.dot(nl) > 0;
    double nc = 1, nt = 1.5, nnt = into ? nc / nt : nt / nc, ddn = rd.dot(nl), cos2t;
    if ((cos2t = 1 - nnt * nnt * (1 - ddn * ddn)) < 0) return e + f.mult(radiance(sc, x, refl_d, depth, rng));
    Vec tdir = (rd * nnt - n * ((into ? 1 : -1) * (ddn * nnt + std::sqrt(cos2t)))).norm();
    double a = nt - nc, b = nt + nc, R0 = a * a / (b * b), c = 1 - (into ? -ddn : tdir.dot(n));
    double Re = R0 + (1 - R0) * c * c * c * c * c, Tr = 1 - Re, P = .25 + .5 * Re, RP = Re / P, TP = Tr / (1 - P);
    if (depth > 2) {
        if (rnd(rng) < P) return e + f.mult(radiance(sc, x, refl_d, depth, rng) * RP);
        return e + f.mult(radiance(sc, x, tdir, depth, rng) * TP);
    }
    const Vec a_refl = radiance(sc, x, refl_d, depth, rng) * Re;   // the reflected ray first (see the header)
    const Vec a_tran = radiance(sc, x, tdir, depth, rng) * Tr;
    return e + f.mult(a_refl + a_tran);
}

struct Camera {
    Vec o, d, cx, cy;
};
Camera make_camera(int w, int h) {   // 93-94
    Camera c;
    c.o = Vec(50, 52, 295.6);
    c.d = Vec(0, -0.042612, -1).norm();
    c.cx = Vec(w * .5135 / h);
    c.cy = (c.cx.cross(c.d)).norm() * .5135;
    return c;
}

Vec camera_sample_radiance(const Scene& sc, const Camera& cam, int w, int h, int x, int y, int sx, int sy, Rng& rng) {   // 104-109
    double r1 = 2 * rnd(rng), dx = r1 < 1 ? std::sqrt(r1) - 1 : 1 - std::sqrt(2 - r1);
    double r2 = 2 * rnd(rng), dy = r2 < 1 ? std::sqrt(r2) - 1 : 1 - std::sqrt(2 - r2);
    Vec d = cam.cx * (((sx + .5 + dx) / 2 + x) / w - .5) + cam.cy * (((sy + .5 + dy) / 2 + y) / h - .5) + cam.d;
    Vec dn = d;
    dn.norm();
    return radiance(sc, cam.o + d * 140, dn, 0, rng);
}

double clamp01(double x) { return x < 0 ? 0 : x > 1 ? 1 : x; }

bool valid(const ky_smallpt_sphere* s, int n, const ky_smallpt_params* p) {
    return s && p && n > 0 && p->width > 0 && p->height > 0 && p->samps > 0 && p->max_depth >= 0;
}

}  // namespace

extern "C" {

// main()'s loop nest, smallpt.cpp:96-114.  image_rgb as kyhip_smallpt_render.  rng_mode: see the header.
int kyo_smallpt_render(const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p, int rng_mode, double* image_rgb) {
    if (!valid(spheres, n, p) || !image_rgb) return -1;
    const Scene sc{spheres, n, p->max_depth};
    const int w = p->width, h = p->height, samps = p->samps;
    const Camera cam = make_camera(w, h);
    std::memset(image_rgb, 0, sizeof(double) * 3 * (size_t)w * h);
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < h; y++) {
        Rng rng;
        rng.mode = 1; rng.s = 0;
        rng.xi[0] = 0; rng.xi[1] = 0; rng.xi[2] = (unsigned short)(y * y * y);   // 98
        for (int x = 0; x < w; x++)
            for (int sy = 0, i = (h - y - 1) * w + x; sy < 2; sy++)
                for (int sx = 0; sx < 2; sx++) {
                    Vec r;
                    for (int s = 0; s < samps; s++) {
                        if (rng_mode == 0) rng_start(rng, p->seed, (uint32_t)((y * w + x) * 4 + sy * 2 + sx), (uint32_t)s);
                        r = r + camera_sample_radiance(sc, cam, w, h, x, y, sx, sy, rng) * (1. / samps);
                    }
                    double* c = image_rgb + (size_t)i * 3;
                    c[0] = c[0] + clamp01(r.x) * .25; c[1] = c[1] + clamp01(r.y) * .25; c[2] = c[2] + clamp01(r.z) * .25;
                }
    }
    return 0;
}

// radiance of samples [s0, s0 + count) of one subpixel, rng_mode 0 (kyhip_smallpt_kat_radiance's twin)
int kyo_smallpt_radiance(const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p, int x, int y, int sx, int sy, int s0,
                         int count, double* out3) {
    if (!valid(spheres, n, p) || !out3) return -1;
    const Scene sc{spheres, n, p->max_depth};
    const Camera cam = make_camera(p->width, p->height);
    for (int i = 0; i < count; ++i) {
        Rng rng;
        rng_start(rng, p->seed, (uint32_t)((y * p->width + x) * 4 + sy * 2 + sx), (uint32_t)(s0 + i));
        const Vec L = camera_sample_radiance(sc, cam, p->width, p->height, x, y, sx, sy, rng);
        out3[3 * i] = L.x; out3[3 * i + 1] = L.y; out3[3 * i + 2] = L.z;
    }
    return 0;
}

// the scene table, restated independently of the product's kyhip_smallpt_scene (smallpt.cpp:42-52)
int kyo_smallpt_scene(ky_smallpt_sphere* out) {
    const double rows[9][10] = {{1e5, 1e5 + 1, 40.8, 81.6, 0, 0, 0, .75, .25, .25},     {1e5, -1e5 + 99, 40.8, 81.6, 0, 0, 0, .25, .25, .75},
                                {1e5, 50, 40.8, 1e5, 0, 0, 0, .75, .75, .75},           {1e5, 50, 40.8, -1e5 + 170, 0, 0, 0, 0, 0, 0},
                                {1e5, 50, 1e5, 81.6, 0, 0, 0, .75, .75, .75},           {1e5, 50, -1e5 + 81.6, 81.6, 0, 0, 0, .75, .75, .75},
                                {16.5, 27, 16.5, 47, 0, 0, 0, .999, .999, .999},        {16.5, 73, 16.5, 78, 0, 0, 0, .999, .999, .999},
                                {600, 50, 681.6 - .27, 81.6, 12, 12, 12, 0, 0, 0}};
    const int refl[9] = {KY_SP_DIFF, KY_SP_DIFF, KY_SP_DIFF, KY_SP_DIFF, KY_SP_DIFF, KY_SP_DIFF, KY_SP_SPEC, KY_SP_REFR, KY_SP_DIFF};
    for (int i = 0; i < 9; ++i) {
        out[i].rad = rows[i][0];
        for (int j = 0; j < 3; ++j) { out[i].p[j] = rows[i][1 + j]; out[i].e[j] = rows[i][4 + j]; out[i].c[j] = rows[i][7 + j]; }
        out[i].refl = refl[i];
        out[i].pad_ = 0;
    }
    return 9;
}

}  // extern "C"
