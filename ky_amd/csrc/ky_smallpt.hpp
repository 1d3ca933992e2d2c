/*
 * ky_smallpt.hpp -- SURVEY 8(f)4: smallpt's own scene and radiance() in double precision (smallpt2pbrt/smallpt.cpp).
 *
 * One thread per (pixel, subpixel) runs that subpixel's `samps` samples in order (smallpt.cpp:106-111) and writes its
 * clamped mean; smallpt_resolve_kernel adds the four subpixels of a pixel in smallpt's order (112).  radiance()'s
 * recursion (56-89) is unrolled onto a small explicit stack: every vertex adds throughput * emission, and the only
 * branching point -- the glass sphere, where both the reflected and the refracted ray are followed while depth <= 2
 * (86-88) -- pushes the refracted ray and goes on with the reflected one, which is the recursion's own depth-first
 * order (reflection subtree, then transmission subtree) and therefore its order of random numbers.
 *
 * All arithmetic is fp64 with contraction off (the CPU build of smallpt has no FMA), IEEE sqrt and division.
 * Included by kyhip.hip only.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kyhip.h"

namespace kysp {

struct SpVec {
    double x, y, z;
};
#define KY_SP_DEV __device__ __forceinline__
KY_SP_DEV SpVec spv(double x, double y, double z) { return SpVec{x, y, z}; }
KY_SP_DEV SpVec operator+(SpVec a, SpVec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
KY_SP_DEV SpVec operator-(SpVec a, SpVec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
KY_SP_DEV SpVec operator*(SpVec a, double b) { return {a.x * b, a.y * b, a.z * b}; }
KY_SP_DEV SpVec mult(SpVec a, SpVec b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
KY_SP_DEV double dot(SpVec a, SpVec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
KY_SP_DEV SpVec norm(SpVec a) { return a * (1 / sqrt(a.x * a.x + a.y * a.y + a.z * a.z)); }       // smallpt.cpp:17
KY_SP_DEV SpVec cross(SpVec a, SpVec b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }  // operator%, 19

constexpr int SP_MAX_SPHERES = 32;
constexpr int SP_STACK = 4;   // the split happens at depth 1 and 2 only: at most 2 refracted rays wait at any time

struct SpSphere {
    double rad, sq_rad;
    double p[3], e[3], c[3];
    int refl, pad;
};

struct SpConst {
    int w, h, samps, n, max_depth;
    uint32_t seed;
    double cx[3], cy[3], cam_o[3], cam_d[3];
};

// one stream of doubles in [0, 1) per (pixel, subpixel, sample): splitmix64 started from a hash of the key
struct SpRng {
    uint64_t s;
};
KY_SP_DEV uint64_t sp_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
KY_SP_DEV void sp_rng_start(SpRng& r, uint32_t seed, uint32_t subpixel_index, uint32_t sample) {
    r.s = sp_mix64(((uint64_t)seed << 32) ^ (uint64_t)subpixel_index) + (uint64_t)sample * 0xD1B54A32D192ED03ull;
    r.s = sp_mix64(r.s);
}
KY_SP_DEV double sp_next(SpRng& r) {   // erand48's role (smallpt.cpp:1): uniform in [0, 1)
    r.s += 0x9E3779B97F4A7C15ull;
    return (double)(sp_mix64(r.s) >> 11) * (1.0 / 9007199254740992.0);
}

// Sphere::intersect, smallpt.cpp:33-38: distance, 0 if no hit
KY_SP_DEV double sp_sphere_intersect(const SpSphere& s, SpVec o, SpVec d) {
#pragma clang fp contract(off)
    const SpVec op = spv(s.p[0], s.p[1], s.p[2]) - o;
    const double eps = 1e-4, b = dot(op, d);
    double det = b * b - dot(op, op) + s.sq_rad;
    if (det < 0) return 0;
    det = sqrt(det);
    double t;
    return (t = b - det) > eps ? t : ((t = b + det) > eps ? t : 0);
}

// intersect(), smallpt.cpp:57-61: spheres are tested from the last to the first, strict `<` keeps the earlier winner
KY_SP_DEV bool sp_intersect(const SpSphere* __restrict__ sph, int n, SpVec o, SpVec d, double& t, int& id) {
    const double inf = 1e20;
    t = inf;
    for (int i = n; i--;) {
        const double dd = sp_sphere_intersect(sph[i], o, d);
        if (dd != 0 && dd < t) { t = dd; id = i; }
    }
    return t < inf;
}

// radiance(r, 0, Xi), smallpt.cpp:63-89
KY_SP_DEV SpVec sp_radiance(const SpSphere* __restrict__ sph, int n, int max_depth, SpVec ro, SpVec rd, SpRng& rng) {
#pragma clang fp contract(off)
    SpVec L = spv(0, 0, 0);
    SpVec st_o[SP_STACK], st_d[SP_STACK], st_f[SP_STACK];
    int st_depth[SP_STACK];
    int sp = 0;
    SpVec thr = spv(1, 1, 1);
    int depth = 0;
    for (;;) {
        bool alive = true;
        double t;
        int id = 0;
        if (!sp_intersect(sph, n, ro, rd, t, id)) alive = false;   // if miss, return black (66)
        if (alive) {
            const SpSphere& obj = sph[id];
            const SpVec e = spv(obj.e[0], obj.e[1], obj.e[2]);
            L = L + mult(thr, e);                                    // every return path of 68-89 adds obj.e
            if (depth > max_depth) alive = false;                    // 69
            if (alive) {
                const SpVec x = ro + rd * t, nrm = norm(x - spv(obj.p[0], obj.p[1], obj.p[2]));
                const SpVec nl = dot(nrm, rd) < 0 ? nrm : nrm * -1;
                SpVec f = spv(obj.c[0], obj.c[1], obj.c[2]);
                const double p = f.x > f.y && f.x > f.z ? f.x : f.y > f.z ? f.y : f.z;   // max refl (72)
                if (++depth > 5) {                                                       // R.R. (73)
                    if (sp_next(rng) < p) f = f * (1 / p);
                    else alive = false;
                }
                if (alive) {
                    if (obj.refl == KY_SP_DIFF) {  // ideal diffuse reflection, 75-79
                        const double r1 = 2 * 3.141592653589793238462643 * sp_next(rng), r2 = sp_next(rng), r2s = sqrt(r2);
                        const SpVec w = nl, u = norm(cross(fabs(w.x) > .1 ? spv(0, 1, 0) : spv(1, 0, 0), w)), v = cross(w, u);
                        rd = norm(u * cos(r1) * r2s + v * sin(r1) * r2s + w * sqrt(1 - r2));
                        ro = x;
                        thr = mult(thr, f);
                    } else if (obj.refl == KY_SP_SPEC) {  // ideal specular reflection, 81-82
                        rd = rd - nrm * 2 * dot(nrm, rd);
                        ro = x;
                        thr = mult(thr, f);
                    } else {  // ideal dielectric refraction, 84-96
                        const SpVec refl_d = rd - nrm * 2 * dot(nrm, rd);
                        const bool into = dot(nrm, nl) > 0;
                        const double nc = 1, nt = 1.5, nnt = into ? nc / nt : nt / nc, ddn = dot(rd, nl);
                        const double cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
                        if (cos2t < 0) {  // total internal reflection, 87-88
                            rd = refl_d;
                            ro = x;
                            thr = mult(thr, f);
                        } else {
                            const SpVec tdir = norm(rd * nnt - nrm * ((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))));
                            const double a = nt - nc, b = nt + nc, R0 = a * a / (b * b), c = 1 - (into ? -ddn : dot(tdir, nrm));
                            const double Re = R0 + (1 - R0) * c * c * c * c * c, Tr = 1 - Re, P = .25 + .5 * Re, RP = Re / P, TP = Tr / (1 - P);
                            if (depth > 2) {  // Russian roulette between the two rays, 93-94
                                if (sp_next(rng) < P) { rd = refl_d; thr = mult(thr, f) * RP; }
                                else { rd = tdir; thr = mult(thr, f) * TP; }
                                ro = x;
                            } else {          // both rays, 95: the reflected one first
                                if (sp < SP_STACK) {
                                    st_o[sp] = x; st_d[sp] = tdir; st_f[sp] = mult(thr, f) * Tr; st_depth[sp] = depth;
                                    ++sp;
                                }
                                rd = refl_d;
                                ro = x;
                                thr = mult(thr, f) * Re;
                            }
                        }
                    }
                }
            }
        }
        if (!alive) {
            if (sp == 0) break;
            --sp;
            ro = st_o[sp]; rd = st_d[sp]; thr = st_f[sp]; depth = st_depth[sp];
        }
    }
    return L;
}

// the camera sample of smallpt.cpp:104-109
KY_SP_DEV void sp_camera_ray(const SpConst& k, int x, int y, int sx, int sy, SpRng& rng, SpVec& o, SpVec& d) {
#pragma clang fp contract(off)
    const double r1 = 2 * sp_next(rng), dx = r1 < 1 ? sqrt(r1) - 1 : 1 - sqrt(2 - r1);
    const double r2 = 2 * sp_next(rng), dy = r2 < 1 ? sqrt(r2) - 1 : 1 - sqrt(2 - r2);
    const SpVec cx = spv(k.cx[0], k.cx[1], k.cx[2]), cy = spv(k.cy[0], k.cy[1], k.cy[2]), cd = spv(k.cam_d[0], k.cam_d[1], k.cam_d[2]);
    const SpVec dir = cx * (((sx + .5 + dx) / 2 + x) / k.w - .5) + cy * (((sy + .5 + dy) / 2 + y) / k.h - .5) + cd;
    o = spv(k.cam_o[0], k.cam_o[1], k.cam_o[2]) + dir * 140;   // camera rays are pushed forward to start in the interior
    d = norm(dir);
}

KY_SP_DEV double sp_clamp(double x) { return x < 0 ? 0 : x > 1 ? 1 : x; }   // 54

__global__ __launch_bounds__(256) void smallpt_kernel(const SpSphere* __restrict__ g_sph, SpConst k, double* __restrict__ sub) {
#pragma clang fp contract(off)
    __shared__ SpSphere sph[SP_MAX_SPHERES];
    for (int i = threadIdx.x; i < k.n * (int)(sizeof(SpSphere) / 8); i += blockDim.x)
        reinterpret_cast<double*>(sph)[i] = reinterpret_cast<const double*>(g_sph)[i];
    __syncthreads();
    // 8 x 8 pixel blocks x 4 subpixels per 256 threads: neighbouring lanes follow similar paths
    const int bw = (k.w + 7) / 8;
    const int block_x = blockIdx.x % bw, block_y = blockIdx.x / bw;
    const int t = threadIdx.x;
    const int x = block_x * 8 + (t & 7), y = block_y * 8 + ((t >> 3) & 7), sx = (t >> 6) & 1, sy = t >> 7;
    if (x >= k.w || y >= k.h) return;
    const uint32_t si = (uint32_t)((y * k.w + x) * 4 + sy * 2 + sx);
    SpVec r = spv(0, 0, 0);
    const double inv = 1. / k.samps;
    for (int s = 0; s < k.samps; ++s) {
        SpRng rng;
        sp_rng_start(rng, k.seed, si, (uint32_t)s);
        SpVec o, d;
        sp_camera_ray(k, x, y, sx, sy, rng, o, d);
        r = r + sp_radiance(sph, k.n, k.max_depth, o, d, rng) * inv;   // 109
    }
    double* out = sub + (size_t)si * 3;
    out[0] = sp_clamp(r.x); out[1] = sp_clamp(r.y); out[2] = sp_clamp(r.z);   // 112
}

// c[i] = c[i] + Vec(clamp(r.x), clamp(r.y), clamp(r.z)) * .25 for sy, sx in loop order (102-103, 112); i = (h - y - 1) * w + x
__global__ void smallpt_resolve_kernel(const double* __restrict__ sub, double* __restrict__ image, int w, int h) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= w * h) return;
    const int x = p % w, y = p / w;
    const double* s = sub + (size_t)p * 12;
    double* c = image + ((size_t)(h - y - 1) * w + x) * 3;
    for (int ch = 0; ch < 3; ++ch) {
        double a = 0.0;
        for (int q = 0; q < 4; ++q) a = a + s[q * 3 + ch] * .25;
        c[ch] = a;
    }
}

__global__ void smallpt_kat_kernel(const SpSphere* __restrict__ sph, SpConst k, int x, int y, int sx, int sy, int s0, int n, double* __restrict__ out3) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    SpRng rng;
    sp_rng_start(rng, k.seed, (uint32_t)((y * k.w + x) * 4 + sy * 2 + sx), (uint32_t)(s0 + i));
    SpVec o, d;
    sp_camera_ray(k, x, y, sx, sy, rng, o, d);
    const SpVec L = sp_radiance(sph, k.n, k.max_depth, o, d, rng);
    out3[3 * i] = L.x; out3[3 * i + 1] = L.y; out3[3 * i + 2] = L.z;
}

// ---- host side ----------------------------------------------------------------------------------
// Ray cam(Vec(50,52,295.6), Vec(0,-0.042612,-1).norm()); cx = Vec(w*.5135/h); cy = (cx % cam.d).norm() * .5135  (93-94)
inline void sp_make_const(const ky_smallpt_params* p, int n, SpConst& k) {
    k.w = p->width; k.h = p->height; k.samps = p->samps; k.n = n; k.max_depth = p->max_depth; k.seed = p->seed;
    const double dx = 0, dy = -0.042612, dz = -1;
    const double il = 1 / std::sqrt(dx * dx + dy * dy + dz * dz);
    const double cd[3] = {dx * il, dy * il, dz * il};
    const double cx[3] = {p->width * .5135 / p->height, 0, 0};
    double cy[3] = {cx[1] * cd[2] - cx[2] * cd[1], cx[2] * cd[0] - cx[0] * cd[2], cx[0] * cd[1] - cx[1] * cd[0]};
    const double cl = 1 / std::sqrt(cy[0] * cy[0] + cy[1] * cy[1] + cy[2] * cy[2]);
    for (int i = 0; i < 3; ++i) { cy[i] = cy[i] * cl * .5135; k.cx[i] = cx[i]; k.cy[i] = cy[i]; k.cam_d[i] = cd[i]; }
    k.cam_o[0] = 50; k.cam_o[1] = 52; k.cam_o[2] = 295.6;
}

inline void sp_pack(const ky_smallpt_sphere* in, int n, SpSphere* out) {
    for (int i = 0; i < n; ++i) {
        out[i].rad = in[i].rad;
        out[i].sq_rad = in[i].rad * in[i].rad;
        for (int j = 0; j < 3; ++j) { out[i].p[j] = in[i].p[j]; out[i].e[j] = in[i].e[j]; out[i].c[j] = in[i].c[j]; }
        out[i].refl = in[i].refl;
        out[i].pad = 0;
    }
}

}  // namespace kysp
