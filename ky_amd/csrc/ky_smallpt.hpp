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
 * Variant 1 (KY_SP_VARIANT_REWRITE) is smallpt2pbrt/smallpt_rewrite.cpp, the pbrt-style step between smallpt and ky.cpp and
 * the one reference program this image can build and run (oracle/_ref/smallpt_rewrite): the same nine spheres mirrored in z
 * (1199-1244), PerspectiveCamera with a 53 degree field of view (651-694, 1391), one uniformly jittered camera sample per
 * sample (RandomSampler, 369-392), RecursionPathIntegrater (1335-1372: no split at the glass sphere, Schlick reflectance as
 * the selection probability, roulette on the BSDF value's largest component once ++depth > 5, depth cap 10), polar disk
 * mapping for the cosine lobe (251-265), one clamp per PIXEL (1316).  Its recursion is a straight chain, so it becomes a
 * plain loop over path vertices with a running throughput.
 *
 * All arithmetic is fp64 with contraction off (the CPU builds have no FMA), IEEE sqrt and division.
 * Included by ky_launch.hip only.
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
    int variant;
    double cx[3], cy[3], cam_o[3], cam_d[3];   // variant 1: right, up, position, front of PerspectiveCamera
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

// ---- variant 1: smallpt_rewrite.cpp ---------------------------------------------------------------
// RNG::UniformFloat2 (305-308) is `Float2(UniformFloat(), UniformFloat())`; the reference binary evaluates the arguments
// right to left, so the FIRST number drawn is .y (pinned by the oracle's byte-exact match with that binary)
struct SpVec2 {
    double x, y;
};
KY_SP_DEV SpVec2 sp_next2(SpRng& r) {
    const double second_arg = sp_next(r);
    const double first_arg = sp_next(r);
    return SpVec2{first_arg, second_arg};
}

struct SpFrame {   // Frame, 122-173
    SpVec s, t, n;
};
KY_SP_DEV SpFrame sp_make_frame(SpVec normal) {   // Frame(n) 131-135, SetFromZ 161-166 (the constant is the float 0.99f)
#pragma clang fp contract(off)
    SpFrame f;
    f.n = norm(normal);
    const SpVec tmp_s = (fabs(f.n.x) > (double)0.99f) ? spv(0, 1, 0) : spv(1, 0, 0);
    f.t = norm(cross(f.n, tmp_s));
    f.s = norm(cross(f.t, f.n));
    return f;
}

// Scene::Intersect (1184-1197) over Sphere::Intersect (758-782): list order, the ray's distance shrinks, strict `<`
KY_SP_DEV bool sp_intersect_rw(const SpSphere* __restrict__ sph, int n, SpVec o, SpVec d, double& t, int& id) {
#pragma clang fp contract(off)
    double distance = __builtin_huge_val();
    bool any = false;
    for (int i = 0; i < n; ++i) {
        const SpVec oc = spv(sph[i].p[0], sph[i].p[1], sph[i].p[2]) - o;
        const double neg_b = dot(oc, d);
        const double det = neg_b * neg_b - dot(oc, oc) + sph[i].sq_rad;
        if (det >= 0) {
            const double sq = sqrt(det), eps = 1e-4;
            const double t0 = neg_b - sq, t1 = neg_b + sq;
            double tt = 0;
            bool hit = false;
            if (t0 > eps && t0 < distance) { tt = t0; hit = true; }
            else if (t1 > eps && t1 < distance) { tt = t1; hit = true; }
            if (hit) { distance = tt; id = i; any = true; }
        }
    }
    t = distance;
    return any;
}

// RecursionPathIntegrater::Li (1345-1372) as a loop: every return path adds the vertex's Le, weighted by the product of
// f * |cos| / pdf of the vertices before it
KY_SP_DEV SpVec sp_radiance_rw(const SpSphere* __restrict__ sph, int n, int max_depth, SpVec ro, SpVec rd, SpRng& rng) {
#pragma clang fp contract(off)
    const double inv_pi = 0.318309886183790671538, pi = 3.14159265358979323846;
    SpVec L = spv(0, 0, 0), thr = spv(1, 1, 1);
    int depth = 0;
    for (;;) {
        double t;
        int id = 0;
        if (!sp_intersect_rw(sph, n, ro, rd, t, id)) break;                     // miss: black (1348-1349)
        const SpSphere& obj = sph[id];
        const SpVec x = ro + rd * t;                                            // Ray::operator(), 189-192
        const SpVec nrm = norm(x - spv(obj.p[0], obj.p[1], obj.p[2]));
        const SpVec wo_w = rd * -1.0;
        const SpVec e = spv(obj.e[0], obj.e[1], obj.e[2]);
        const bool is_light = !((e.x <= 0) && (e.y <= 0) && (e.z <= 0));       // the primitive carries the AreaLight (1241)
        if (is_light && dot(nrm, wo_w) > 0) L = L + mult(thr, e);              // AreaLight::Le, 1114-1117
        if (depth > max_depth) break;                                           // 1351
        const SpFrame fr = sp_make_frame(nrm);
        const SpVec wo = spv(dot(fr.s, wo_w), dot(fr.t, wo_w), dot(fr.n, wo_w));
        const SpVec2 u = sp_next2(rng);                                         // sampler.Get2D(), 1354
        const SpVec col = spv(obj.c[0], obj.c[1], obj.c[2]);
        SpVec f = spv(0, 0, 0), wi = spv(0, 0, 0);
        double pdf = 0;
        if (obj.refl == KY_SP_DIFF) {          // LambertionReflection::Sample_f_, 888-904 + CosineSampleHemisphere 251-265
            const double radius = sqrt(u.x), theta = 2 * pi * u.y;
            const double px = radius * cos(theta), py = radius * sin(theta);
            double z = sqrt(fmax(0.0, 1 - px * px - py * py));
            if (wo.z < 0) z = z * -1;
            wi = spv(px, py, z);
            pdf = (wo.z * wi.z > 0) ? fabs(wi.z) * inv_pi : 0;
            f = col * inv_pi;
        } else if (obj.refl == KY_SP_SPEC) {   // SpecularReflection::Sample_f_, 918-930
            wi = spv(-wo.x, -wo.y, wo.z);
            pdf = 1;
            const double ac = fabs(wi.z);
            f = spv(col.x / ac, col.y / ac, col.z / ac);
        } else {                               // FresnelSpecular::Sample_f_ with etaI = 1, etaT = 1.5, 946-1020
            const double eta_i = 1, eta_t = 1.5;
            const bool into = wo.z > 0;
            const double nz = into ? 1.0 : -1.0;
            const double eta = into ? eta_i / eta_t : eta_t / eta_i;
            const double cos_i = wo.z * nz;
            const double cos_t2 = 1 - eta * eta * (1 - cos_i * cos_i);
            if (!(cos_t2 < 0)) {               // else total internal reflection: f = 0, pdf = 0 (972-975)
                const double cos_t = sqrt(cos_t2);
                const double k = cos_i * eta - cos_t;
                const SpVec refr = norm(spv(-wo.x * eta + 0.0 * k, -wo.y * eta + 0.0 * k, -wo.z * eta + nz * k));
                const double a = eta_t - eta_i, b = eta_t + eta_i, R0 = a * a / (b * b);
                const double c = 1 - (into ? cos_i : cos_t);
                const double Re = R0 + (1 - R0) * c * c * c * c * c, Tr = 1 - Re;
                if (u.x < Re) { wi = spv(-wo.x, -wo.y, wo.z); pdf = Re; const double ac = fabs(wi.z); f = spv(col.x * Re / ac, col.y * Re / ac, col.z * Re / ac); }
                else { wi = refr; pdf = Tr; const double ac = fabs(wi.z); f = spv(col.x * Tr / ac, col.y * Tr / ac, col.z * Tr / ac); }
            }
        }
        const SpVec wi_w = fr.s * wi.x + fr.t * wi.y + fr.n * wi.z;             // ToWorld, 147-153
        if (((f.x <= 0) && (f.y <= 0) && (f.z <= 0)) || pdf == 0.0) break;    // 1355-1356
        if (++depth > 5) {                                                      // russian roulette, 1359-1366
            const double mc = fmax(f.x, fmax(f.y, f.z));
            if (sp_next(rng) < mc) f = f * (1 / mc);
            else break;
        }
        const double ad = fabs(dot(wi_w, nrm));
        thr = spv(thr.x * (f.x * ad / pdf), thr.y * (f.y * ad / pdf), thr.z * (f.z * ad / pdf));   // 1369
        ro = x;                                                                 // Ray wi(isect.position, bs.wi): no offset
        rd = wi_w;
    }
    return L;
}

// RandomSampler::GetCameraSample (388-391) + PerspectiveCamera::GenerateRay (669-677)
KY_SP_DEV void sp_camera_ray_rw(const SpConst& k, int x, int y, SpRng& rng, SpVec& o, SpVec& d) {
#pragma clang fp contract(off)
    const SpVec2 u = sp_next2(rng);
    const double fx = (double)x + u.x, fy = (double)y + u.y;
    const SpVec right = spv(k.cx[0], k.cx[1], k.cx[2]), up = spv(k.cy[0], k.cy[1], k.cy[2]), front = spv(k.cam_d[0], k.cam_d[1], k.cam_d[2]);
    const SpVec dir = front + right * (fx / k.w - 0.5) + up * (0.5 - fy / k.h);
    o = spv(k.cam_o[0], k.cam_o[1], k.cam_o[2]) + dir * 140;
    d = norm(dir);
}

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
    if (k.variant == KY_SP_VARIANT_REWRITE) {
        // the pixel's samples are dealt round-robin to its four threads; the per-thread partial means are added and clamped
        // once per pixel by the resolve kernel (Integrater::Render, 1306-1316)
        for (int s = sx + 2 * sy; s < k.samps; s += 4) {
            SpRng rng;
            sp_rng_start(rng, k.seed, (uint32_t)(y * k.w + x), (uint32_t)s);
            SpVec o, d;
            sp_camera_ray_rw(k, x, y, rng, o, d);
            r = r + sp_radiance_rw(sph, k.n, k.max_depth, o, d, rng) * inv;
        }
        double* out = sub + (size_t)si * 3;
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
        return;
    }
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
__global__ void smallpt_resolve_kernel(const double* __restrict__ sub, double* __restrict__ image, int w, int h, int variant) {
#pragma clang fp contract(off)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= w * h) return;
    const int x = p % w, y = p / w;
    const double* s = sub + (size_t)p * 12;
    if (variant == KY_SP_VARIANT_REWRITE) {   // film.add_color(x, y, Clamp(L)) into a cleared film, row 0 = top (1316, 517-526)
        double* c = image + (size_t)p * 3;
        for (int ch = 0; ch < 3; ++ch) c[ch] = sp_clamp(((s[ch] + s[3 + ch]) + s[6 + ch]) + s[9 + ch]);
        return;
    }
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
    SpVec o, d, L;
    if (k.variant == KY_SP_VARIANT_REWRITE) {
        sp_rng_start(rng, k.seed, (uint32_t)(y * k.w + x), (uint32_t)(s0 + i));
        sp_camera_ray_rw(k, x, y, rng, o, d);
        L = sp_radiance_rw(sph, k.n, k.max_depth, o, d, rng);
    } else {
        sp_rng_start(rng, k.seed, (uint32_t)((y * k.w + x) * 4 + sy * 2 + sx), (uint32_t)(s0 + i));
        sp_camera_ray(k, x, y, sx, sy, rng, o, d);
        L = sp_radiance(sph, k.n, k.max_depth, o, d, rng);
    }
    out3[3 * i] = L.x; out3[3 * i + 1] = L.y; out3[3 * i + 2] = L.z;
}

// ---- host side ----------------------------------------------------------------------------------
// Ray cam(Vec(50,52,295.6), Vec(0,-0.042612,-1).norm()); cx = Vec(w*.5135/h); cy = (cx % cam.d).norm() * .5135  (93-94)
inline void sp_make_const(const ky_smallpt_params* p, int n, SpConst& k) {
    k.w = p->width; k.h = p->height; k.samps = p->samps; k.n = n; k.max_depth = p->max_depth; k.seed = p->seed;
    k.variant = p->variant;
    if (p->variant == KY_SP_VARIANT_REWRITE) {
        // main(): PerspectiveCamera({50, 52, -295.6}, normalize({0, -0.042612, 1}), {0, 1, 0}, 53, resolution) (1391-1392);
        // constructor 654-666: right = normalize(up x front) * tan(fov / 2) * aspect, up = normalize(front x right) * tan(fov / 2)
        auto nrm = [](double* v) { const double il = 1 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] = v[0] * il; v[1] = v[1] * il; v[2] = v[2] * il; };
        auto crs = [](const double* a, const double* b, double* r) { r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0]; };
        double front[3] = {0, -0.042612, 1}, up0[3] = {0, 1, 0}, right[3], up[3];
        nrm(front);
        const double tan_fov = std::tan(((3.14159265358979323846 / 180) * 53) / 2);
        const double aspect = (double)p->width / (double)p->height;
        crs(up0, front, right); nrm(right);
        for (int i = 0; i < 3; ++i) right[i] = right[i] * tan_fov * aspect;
        crs(front, right, up); nrm(up);
        for (int i = 0; i < 3; ++i) up[i] = up[i] * tan_fov;
        for (int i = 0; i < 3; ++i) { k.cx[i] = right[i]; k.cy[i] = up[i]; k.cam_d[i] = front[i]; }
        k.cam_o[0] = 50; k.cam_o[1] = 52; k.cam_o[2] = -295.6;
        return;
    }
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
