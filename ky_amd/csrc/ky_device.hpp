/*
 * ky_device.hpp -- gfx950 device code of the path-tracing hot path (one path vertex per call).
 *
 * Written for CDNA4: 64-lane wavefronts, one lane = one pixel.  The primitive list is read with
 * wave-uniform indices (scalar loads into SGPRs, no VGPR or LDS traffic for the traversal); the
 * tables that are looked up with a per-lane index AFTER the nearest hit is known (surface ->
 * normal / material / light, materials, light radiance) are staged in LDS once per workgroup.
 * No MFMA: there is no dense contraction on this path.
 *
 * Every function names the reference function it implements (file = /root/reference/ky.cpp).
 * This file is independent of oracle/ (the CPU checker): nothing is shared between the two.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kyhip.h"

#define KY_DEV __device__ __forceinline__

namespace kyd {

// ---------------------------------------------------------------------------------------------
// vectors (ky.cpp:226-388)
// ---------------------------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};
KY_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
KY_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
KY_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
KY_DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
KY_DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
KY_DEV f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
KY_DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
KY_DEV f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
KY_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
KY_DEV f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
KY_DEV float length_sq(f3 a) { return dot(a, a); }
KY_DEV f3 normalize(f3 a) { return a * (1.0f / sqrtf(dot(a, a))); }  // vec3_t::normalize, 314
KY_DEV float max3(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
KY_DEV bool is_black(f3 c) { return (c.x <= 0) && (c.y <= 0) && (c.z <= 0); }  // color_t::is_black, 258

constexpr float K_PI = 3.14159265358979323846f;
constexpr float K_2PI = 2.f * K_PI;
constexpr float K_PI_OVER2 = K_PI / 2.f;
constexpr float K_PI_OVER4 = K_PI / 4.f;
constexpr float K_INV_PI = 0.318309886183790671538f;
constexpr float K_INV_2PI = K_INV_PI / 2.f;
constexpr float K_SHAPE_EPS = 1e-3f;   // shape_t::epsilon, 1093
constexpr float K_RAY_OFFSET = 1e-2f;  // offset_ray_origin, 616
constexpr float K_INF = __builtin_huge_valf();

// ---------------------------------------------------------------------------------------------
// device scene layout (HBM, read-only; 16-byte aligned records)
// ---------------------------------------------------------------------------------------------
struct DSurf {  // traversal record: the shape of one surface_t, read with a wave-uniform index
    float p[4][3];
    float n[3];
    float radius_sq;
    int32_t kind;
    float radius;
    int32_t pad[2];
};  // 80 B

struct DHit {  // what is needed once the nearest surface is known; gathered per lane from LDS
    float n[3];  // stored normal, or the sphere centre
    int32_t kind;
    int32_t material;
    int32_t area_light;
    int32_t pad[2];
};  // 32 B

struct DMat {  // ky_material, gathered per lane from LDS
    float c0[3];
    int32_t kind;
    float c1[3];
    float eta;
    float exponent, p_diffuse, p_specular, pad;
};  // 48 B

struct DLight {  // light_t + the shape an area light samples; wave-uniform index
    float color[3];
    int32_t kind;
    float position[3];
    float world_radius;
    float direction[3];
    int32_t shape_kind;
    float p[4][3];  // sampled shape
    float n[3];
    float radius;
    float area, pad[3];
};  // 128 B

struct DScene {
    int32_t n_surfaces, n_lights, n_materials, env_light;
    float cam_position[3], cam_w;
    float cam_front[3], cam_h;
    float cam_right[3], pad0;
    float cam_up[3], pad1;
    DSurf surf[KYHIP_MAX_SURFACES];
    DHit hit[KYHIP_MAX_SURFACES];
    DMat mat[KYHIP_MAX_MATERIALS];
    DLight light[KYHIP_MAX_LIGHTS];
};

struct LdsScene {  // the per-workgroup LDS copy of the tables that are indexed per lane
    DHit hit[KYHIP_MAX_SURFACES];
    DMat mat[KYHIP_MAX_MATERIALS];
    float light_color[KYHIP_MAX_LIGHTS][4];
};

KY_DEV f3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

// cooperative copy global -> LDS, whole workgroup
KY_DEV void stage_scene(LdsScene& L, const DScene* __restrict__ S) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const uint32_t* src_h = reinterpret_cast<const uint32_t*>(S->hit);
    uint32_t* dst_h = reinterpret_cast<uint32_t*>(L.hit);
    for (int i = tid; i < S->n_surfaces * (int)(sizeof(DHit) / 4); i += nt) dst_h[i] = src_h[i];
    const uint32_t* src_m = reinterpret_cast<const uint32_t*>(S->mat);
    uint32_t* dst_m = reinterpret_cast<uint32_t*>(L.mat);
    for (int i = tid; i < S->n_materials * (int)(sizeof(DMat) / 4); i += nt) dst_m[i] = src_m[i];
    for (int i = tid; i < S->n_lights; i += nt) {
        L.light_color[i][0] = S->light[i].color[0];
        L.light_color[i][1] = S->light[i].color[1];
        L.light_color[i][2] = S->light[i].color[2];
        L.light_color[i][3] = 0.f;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// random numbers: counter-based, keyed (seed, pixel, sample, dimension) -- DESIGN.md
// sampler_t semantics of ky.cpp:877-975
// ---------------------------------------------------------------------------------------------
KY_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
constexpr uint32_t KY_DIM_LOBE = 0xFFFF0000u;

struct Sampler {
    uint32_t k0, k1, dim;
};
KY_DEV void sampler_start(Sampler& s, uint32_t seed, uint32_t pixel_index, uint32_t sample_index) {
    const uint32_t h = mix32(pixel_index ^ mix32(seed));
    s.k0 = mix32(h + sample_index * 0x9E3779B9u);
    s.k1 = mix32((h ^ 0x6A09E667u) + sample_index * 0x85EBCA6Bu);
    s.dim = 0;
}
template <bool DEBUG_SAMPLER>
KY_DEV float sampler_at(const Sampler& s, uint32_t d) {
    if (DEBUG_SAMPLER) return 0.5f;  // debug_sampler_t, 933-941
    const uint32_t x = mix32(s.k0 ^ mix32(s.k1 + d));
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------------------------------------
// camera_t::generate_ray, ky.cpp:1884-1892
// ---------------------------------------------------------------------------------------------
KY_DEV void generate_ray(const DScene* __restrict__ S, float px, float py, f3& o, f3& d) {
    const float sx = px / S->cam_w - 0.5f;
    const float sy = 0.5f - py / S->cam_h;
    f3 dir = ld3(S->cam_front) + ld3(S->cam_right) * sx + ld3(S->cam_up) * sy;
    o = ld3(S->cam_position);
    d = normalize(dir);
}

// offset_ray_origin, ky.cpp:614-620
KY_DEV f3 offset_ray_origin(f3 position, f3 normal, f3 direction) {
    f3 offset = normal * K_RAY_OFFSET;
    if (dot(normal, direction) < 0) offset = -offset;
    return position + offset;
}

// ---------------------------------------------------------------------------------------------
// shape_t::intersect x4 -- distance only.  Returns true and sets t when eps < t < tmax.
// ---------------------------------------------------------------------------------------------
KY_DEV bool is_equal_zero(float x) {  // is_equal(x, 0.f), ky.cpp:212-220
    const float eps = 1.1920929e-07f;
    return fabsf(x) <= eps * fmaxf(1.f, fabsf(x));
}

template <typename SHAPE>  // SHAPE has p[4][3], n[3], radius, radius_sq, kind
KY_DEV bool shape_hit(const SHAPE& S, int kind, f3 o, f3 d, float tmax, float& t_out) {
    if (kind == KY_SHAPE_RECTANGLE) {  // rectangle_t::intersect, 1261-1297
        const f3 oa = ld3(S.p[0]) - o, ob = ld3(S.p[1]) - o, oc = ld3(S.p[2]) - o, od = ld3(S.p[3]) - o;
        const float v0d = dot(cross(oc, ob), d), v1d = dot(cross(ob, oa), d), v2d = dot(cross(oa, od), d), v3d = dot(cross(od, oc), d);
        const bool neg = (v0d < 0.f) && (v1d < 0.f) && (v2d < 0.f) && (v3d < 0.f);
        const bool pos = (v0d >= 0.f) && (v1d >= 0.f) && (v2d >= 0.f) && (v3d >= 0.f);
        const f3 n = ld3(S.n);
        const float t = dot(n, oa) / dot(n, d);
        t_out = t;
        return (neg || pos) && (t > K_SHAPE_EPS) && (t < tmax);
    } else if (kind == KY_SHAPE_SPHERE) {  // sphere_t::intersect, 1336-1393
        const f3 oc = ld3(S.p[0]) - o;
        const float neg_b = dot(oc, d);
        const float discr = neg_b * neg_b - dot(oc, oc) + S.radius_sq;
        bool hit = false;
        float t = 0.f;
        if (discr >= 0) {
            const float sq = sqrtf(discr);
            const float t0 = neg_b - sq, t1 = neg_b + sq;
            if (t0 > K_SHAPE_EPS && t0 < tmax) { hit = true; t = t0; }
            else if (t1 > K_SHAPE_EPS && t1 < tmax) { hit = true; t = t1; }
        }
        t_out = t;
        return hit;
    } else if (kind == KY_SHAPE_TRIANGLE) {  // triangle_t::intersect, 1179-1215
        const f3 oa = ld3(S.p[0]) - o, ob = ld3(S.p[1]) - o, oc = ld3(S.p[2]) - o;
        const float v0d = dot(cross(oc, ob), d), v1d = dot(cross(ob, oa), d), v2d = dot(cross(oa, oc), d);
        const bool neg = (v0d < 0.f) && (v1d < 0.f) && (v2d < 0.f);
        const bool pos = (v0d >= 0.f) && (v1d >= 0.f) && (v2d >= 0.f);
        const f3 n = ld3(S.n);
        const float t = dot(n, oa) / dot(n, d);
        t_out = t;
        return (neg || pos) && (t > K_SHAPE_EPS) && (t < tmax);
    } else {  // disk_t::intersect, 1111-1132
        const f3 n = ld3(S.n), c = ld3(S.p[0]);
        const float nd = dot(d, n);
        const float t = dot(n, c - o) / dot(n, d);
        const f3 hp = o + t * d;
        const f3 r = c - hp;
        t_out = t;
        return !is_equal_zero(nd) && (t > K_SHAPE_EPS) && (t < tmax) && (sqrtf(dot(r, r)) <= S.radius);
    }
}

// scene_t::intersect, ky.cpp:3172-3184: linear scan in surface order, tmax shrinks, first of equals wins.
KY_DEV int trace_nearest(const DScene* __restrict__ S, f3 o, f3 d, float& tmax) {
    int best = -1;
    const int n = S->n_surfaces;
    for (int i = 0; i < n; ++i) {
        float t;
        if (shape_hit(S->surf[i], S->surf[i].kind, o, d, tmax, t)) {
            tmax = t;
            best = i;
        }
    }
    return best;
}

// scene_t::occluded's traversal (3193-3195): any hit inside (eps, tmax) occludes.
KY_DEV bool trace_any(const DScene* __restrict__ S, f3 o, f3 d, float tmax) {
    bool occ = false;
    const int n = S->n_surfaces;
    for (int i = 0; i < n; ++i) {
        float t;
        occ = occ || shape_hit(S->surf[i], S->surf[i].kind, o, d, tmax, t);
        if (__all(occ)) break;
    }
    return occ;
}

// normal the shape reports for a hit (1125, 1208, 1289, 1389)
KY_DEV f3 hit_normal(const DHit& H, f3 position, f3 d) {
    const f3 n = ld3(H.n);
    if (H.kind == KY_SHAPE_SPHERE) return normalize(position - n);
    if (H.kind == KY_SHAPE_RECTANGLE) return dot(n, d) <= 0 ? n : -n;
    return n;
}

// ---------------------------------------------------------------------------------------------
// frame_t (ky.cpp:526-578)
// ---------------------------------------------------------------------------------------------
struct Frame {
    f3 s, t, n;
};
KY_DEV Frame make_frame(f3 normal) {  // frame_t(normal_t), 537-541, 566-571
    Frame f;
    f.n = normalize(normal);
    const f3 a = (fabsf(f.n.x) > 0.99f) ? mk3(0, 1, 0) : mk3(1, 0, 0);
    f.t = normalize(cross(f.n, a));
    f.s = normalize(cross(f.t, f.n));
    return f;
}
KY_DEV f3 to_local(const Frame& f, f3 w) { return {dot(f.s, w), dot(f.t, w), dot(f.n, w)}; }
KY_DEV f3 to_world(const Frame& f, f3 l) { return f.s * l.x + f.t * l.y + f.n * l.z; }

// ---------------------------------------------------------------------------------------------
// BSDFs (ky.cpp:2092-2555), local shading frame
// ---------------------------------------------------------------------------------------------
enum : int { LOBE_LAMBERT = 0, LOBE_MIRROR = 1, LOBE_GLASS = 2, LOBE_PHONG = 3 };
enum : int { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16 };

struct Bsdf {
    int lobe;
    f3 a, b;       // lambert albedo | mirror R | glass R, T | phong Ks
    float eta_t;   // glass (eta_i = 1, 2630)
    float exponent;
};
KY_DEV bool bsdf_is_delta(const Bsdf& B) { return B.lobe == LOBE_MIRROR || B.lobe == LOBE_GLASS; }

// material_t::scattering x4 (2587, 2604, 2628, 2661)
KY_DEV Bsdf make_bsdf(const DMat& M, float lobe_random) {
    Bsdf B;
    B.a = ld3(M.c0);
    B.b = ld3(M.c1);
    B.eta_t = M.eta;
    B.exponent = M.exponent;
    B.lobe = LOBE_LAMBERT;
    if (M.kind == KY_MATERIAL_MIRROR) B.lobe = LOBE_MIRROR;
    else if (M.kind == KY_MATERIAL_GLASS) B.lobe = LOBE_GLASS;
    else if (M.kind == KY_MATERIAL_PLASTIC) {
        if (lobe_random < M.p_specular) { B.lobe = LOBE_PHONG; B.a = ld3(M.c1) / M.p_specular; }
        else { B.a = ld3(M.c0) / M.p_diffuse; }
    }
    return B;
}

// fresnel_dielectric, 1963-1996
KY_DEV float fresnel_dielectric(float cos_theta_i, float eta_i, float eta_t) {
    cos_theta_i = fminf(fmaxf(cos_theta_i, -1.f), 1.f);
    if (!(cos_theta_i > 0.f)) {
        const float tmp = eta_i; eta_i = eta_t; eta_t = tmp;
        cos_theta_i = fabsf(cos_theta_i);
    }
    const float sin_theta_i = sqrtf(fmaxf(0.f, 1 - cos_theta_i * cos_theta_i));
    const float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1) return 1;
    const float cos_theta_t = sqrtf(fmaxf(0.f, 1 - sin_theta_t * sin_theta_t));
    const float r_para = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    const float r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return (r_para * r_para + r_perp * r_perp) / 2;
}

// phong helpers: wr = reflect(wo, (0,0,1)) = (-wo.x, -wo.y, wo.z), 2495 / 2504 / 2517
KY_DEV float phong_pow(float base, float exponent) { return powf(base, exponent); }

// bsdf eval_ and pdf_ at one (wo, wi) pair
KY_DEV void bsdf_eval_pdf(const Bsdf& B, f3 wo, f3 wi, f3& f, float& pdf) {
    f = mk3(0, 0, 0);
    pdf = 0.f;
    const bool same = wo.z * wi.z > 0;  // same_hemisphere, 1921
    if (B.lobe == LOBE_LAMBERT) {       // 2227-2240
        if (same) { f = B.a * K_INV_PI; pdf = fabsf(wi.z) * K_INV_PI; }
    } else if (B.lobe == LOBE_PHONG) {  // 2489-2508, 2545-2550
        const float cos_alpha = -wo.x * wi.x - wo.y * wi.y + wo.z * wi.z;
        // eval: cos_alpha is not clamped (a negative base with an even integral exponent is positive);
        // pdf: clamped at 0, no hemisphere test.
        const float pe = phong_pow(cos_alpha, B.exponent);
        // pow(max(0, cos_alpha), n): equals pe for a positive base, pow(0, n) otherwise
        const float p0 = B.exponent == 0.f ? 1.f : (B.exponent > 0.f ? 0.f : K_INF);
        const float pp = cos_alpha > 0.f ? pe : p0;
        if (same) f = (B.a * (B.exponent + 2.f) * K_INV_2PI) * pe;
        pdf = (B.exponent + 1.f) * pp * K_INV_2PI;
    }
    // mirror / glass: eval 0, pdf 0 (2289-2290, 2352-2353)
}

struct BsdfSample {
    f3 f, wi;
    float pdf;
    int flags;
};

// bsdf sample_ x4
KY_DEV BsdfSample bsdf_sample_local(const Bsdf& B, f3 wo, float u0, float u1) {
    BsdfSample s;
    s.f = mk3(0, 0, 0);
    s.wi = mk3(0, 0, 0);
    s.pdf = 0.f;
    s.flags = 0;
    if (B.lobe == LOBE_LAMBERT) {  // 2242-2257 + cosine_hemisphere_sample 737-743 + concentric_disk_sample 710-733
        const float rx = 2.f * u0 - 1, ry = 2.f * u1 - 1;
        float px = 0.f, py = 0.f;
        if (!(rx == 0 && ry == 0)) {
            float radius, theta;
            if (fabsf(rx) > fabsf(ry)) { radius = rx; theta = K_PI_OVER4 * (ry / rx); }
            else { radius = ry; theta = K_PI_OVER2 - K_PI_OVER4 * (rx / ry); }
            float sn, cs;
            sincosf(theta, &sn, &cs);
            px = cs * radius;
            py = sn * radius;
        }
        float z = sqrtf(fmaxf(0.f, 1 - px * px - py * py));
        if (wo.z < 0) z *= -1;
        s.wi = mk3(px, py, z);
        bsdf_eval_pdf(B, wo, s.wi, s.f, s.pdf);
        s.flags = BSDF_REFLECTION | BSDF_DIFFUSE;
    } else if (B.lobe == LOBE_MIRROR) {  // 2292-2307
        s.wi = mk3(-wo.x, -wo.y, wo.z);
        s.f = B.a / fabsf(s.wi.z);
        s.pdf = 1;
        s.flags = BSDF_REFLECTION | BSDF_SPECULAR;
    } else if (B.lobe == LOBE_GLASS) {  // 2355-2412
        const float reflect_percent = fresnel_dielectric(wo.z, 1.f, B.eta_t);
        const float refract_percent = 1 - reflect_percent;
        if (u0 < reflect_percent) {
            s.wi = mk3(-wo.x, -wo.y, wo.z);
            s.pdf = reflect_percent;
            s.f = (B.a * reflect_percent) / fabsf(s.wi.z);
            s.flags = BSDF_REFLECTION | BSDF_SPECULAR;
        } else {
            const bool into = wo.z > 0;
            const float nz = into ? 1.f : -1.f;
            const float eta = into ? 1.f / B.eta_t : B.eta_t / 1.f;
            // refract(wo, (0,0,nz), eta), 1931-1957
            const float cos_theta_i = nz * wo.z;
            const float sin_theta_i_sq = fmaxf(0.f, 1 - cos_theta_i * cos_theta_i);
            const float sin_theta_t_sq = eta * eta * sin_theta_i_sq;
            if (!(sin_theta_t_sq >= 1)) {
                const float cos_theta_t = sqrtf(1 - sin_theta_t_sq);
                const float k = eta * cos_theta_i - cos_theta_t;
                s.wi = mk3(eta * -wo.x, eta * -wo.y, eta * -wo.z + k * nz);
                s.pdf = refract_percent;
                s.f = (B.b * refract_percent) / fabsf(s.wi.z);
                s.flags = BSDF_TRANSMISSION | BSDF_SPECULAR;
            }
            // else total internal reflection: f = 0, pdf = 0 (2407)
        }
    } else {  // phong, 2510-2529 + 2533-2543
        const float phi = 2.f * K_PI * u0;
        const float ct = powf(u1, 1.f / (B.exponent + 1.f));
        const float st = sqrtf(1.f - ct * ct);
        float sn, cs;
        sincosf(phi, &sn, &cs);
        const f3 local = mk3(cs * st, sn * st, ct);
        const Frame fr = make_frame(mk3(-wo.x, -wo.y, wo.z));
        f3 wi = to_world(fr, local);
        if (wo.z < 0) wi.z *= -1;
        s.wi = wi;
        bsdf_eval_pdf(B, wo, wi, s.f, s.pdf);
        s.flags = BSDF_REFLECTION | BSDF_GLOSSY;
    }
    return s;
}

// ---------------------------------------------------------------------------------------------
// path vertex (isect_t, 642-690)
// ---------------------------------------------------------------------------------------------
struct Vertex {
    f3 position, normal, wo;
    Frame frame;
    Bsdf bsdf;
    int surface;
};

// ---------------------------------------------------------------------------------------------
// lights (ky.cpp:2764-3062) and the shape sampling they call (1028-1090, 1404-1513)
// ---------------------------------------------------------------------------------------------
struct LightSample {
    f3 position, wi, Li;
    float pdf;
};

KY_DEV float light_shape_area(const DLight& L) { return L.area; }

KY_DEV f3 uniform_sphere_sample(float u0, float u1) {  // 761-769
    const float z = 1 - 2 * u0;
    const float radius = sqrtf(fmaxf(0.f, 1.f - z * z));
    const float phi = 2 * K_PI * u1;
    float sn, cs;
    sincosf(phi, &sn, &cs);
    return mk3(radius * cs, radius * sn, z);
}

// shape_t::sample_position x4 (1144, 1225, 1307, 1404)
KY_DEV void shape_sample_position(const DLight& L, float u0, float u1, f3& position, f3& normal) {
    if (L.shape_kind == KY_SHAPE_RECTANGLE) {
        const f3 p0 = ld3(L.p[0]), p1 = ld3(L.p[1]), p2 = ld3(L.p[2]);
        position = p1 + (p0 - p1) * u0 + (p2 - p1) * u1;
        normal = normalize(ld3(L.n));
    } else if (L.shape_kind == KY_SHAPE_SPHERE) {
        const f3 dir = uniform_sphere_sample(u0, u1);
        position = ld3(L.p[0]) + L.radius * dir;
        normal = normalize(dir);
    } else if (L.shape_kind == KY_SHAPE_TRIANGLE) {
        const float su0 = sqrtf(u0);
        const float bx = 1 - su0, by = u1 * su0;
        position = bx * ld3(L.p[0]) + by * ld3(L.p[1]) + (1 - bx - by) * ld3(L.p[2]);
        normal = ld3(L.n);
    } else {  // disk
        const Frame fr = make_frame(ld3(L.n));
        const float rx = 2.f * u0 - 1, ry = 2.f * u1 - 1;
        float px = 0.f, py = 0.f;
        if (!(rx == 0 && ry == 0)) {
            float radius, theta;
            if (fabsf(rx) > fabsf(ry)) { radius = rx; theta = K_PI_OVER4 * (ry / rx); }
            else { radius = ry; theta = K_PI_OVER2 - K_PI_OVER4 * (rx / ry); }
            float sn, cs;
            sincosf(theta, &sn, &cs);
            px = cs * radius;
            py = sn * radius;
        }
        position = ld3(L.p[0]) + L.radius * (fr.s * px + fr.t * py);
        normal = normalize(ld3(L.n));
    }
}

// shape_t::sample_direction (1028-1051) and sphere_t::sample_direction (1419-1501)
KY_DEV void shape_sample_direction(const DLight& L, f3 p, f3 p_normal, float u0, float u1, f3& lposition, f3& lnormal, float& pdf) {
    const bool sphere = L.shape_kind == KY_SHAPE_SPHERE;
    const f3 c = ld3(L.p[0]);
    if (sphere && !(length_sq(p - c) <= L.radius * L.radius)) {
        // outside the sphere: uniform cone sampling, 1458-1500
        const float dist = sqrtf(length_sq(p - c));
        const float inv_dist = 1 / dist;
        const float sin_theta_max = L.radius * inv_dist;
        const float sin_theta_max_sq = sin_theta_max * sin_theta_max;
        const float inv_sin_theta_max = 1 / sin_theta_max;
        const float cos_theta_max = sqrtf(fmaxf(0.f, 1 - sin_theta_max_sq));
        float cos_theta = (cos_theta_max - 1) * u0 + 1;
        float sin_theta_sq = 1 - cos_theta * cos_theta;
        if (sin_theta_max_sq < 0.00068523f) {
            sin_theta_sq = sin_theta_max_sq * u0;
            cos_theta = sqrtf(1 - sin_theta_sq);
        }
        const float cos_alpha = sin_theta_sq * inv_sin_theta_max +
                                cos_theta * sqrtf(fmaxf(0.f, 1.f - sin_theta_sq * inv_sin_theta_max * inv_sin_theta_max));
        const float sin_alpha = sqrtf(fmaxf(0.f, 1.f - cos_alpha * cos_alpha));
        const float phi = u1 * 2 * K_PI;
        const Frame fr = make_frame((c - p) * inv_dist);
        float sn, cs;
        sincosf(phi, &sn, &cs);
        const f3 world_normal = sin_alpha * cs * (-fr.s) + sin_alpha * sn * (-fr.t) + cos_alpha * (-fr.n);  // 431-439
        lposition = c + L.radius * world_normal;
        lnormal = world_normal;
        pdf = 1 / (2 * K_PI * (1 - cos_theta_max));
        return;
    }
    shape_sample_position(L, u0, u1, lposition, lnormal);
    const float area_pdf = 1 / L.area;
    f3 wi = lposition - p;
    const float d2 = length_sq(wi);
    if (d2 == 0) {
        pdf = 0;
    } else {
        wi = normalize(wi);
        // inside-sphere case divides by the SHADE POINT's normal (quirk, 1436); the base class by the light's (1044)
        const f3 nn = sphere ? p_normal : lnormal;
        pdf = area_pdf * d2 / fabsf(dot(nn, -wi));
        if (isinf(pdf)) pdf = 0.f;
    }
}

// shape_t::pdf_direction (1055-1090) and sphere_t::pdf_direction (1503-1513)
KY_DEV float shape_pdf_direction(const DLight& L, f3 p, f3 p_normal, f3 wi) {
    const f3 c = ld3(L.p[0]);
    if (L.shape_kind == KY_SHAPE_SPHERE && !(length_sq(p - c) <= L.radius * L.radius)) {
        const float sin_theta_max_sq = L.radius * L.radius / length_sq(p - c);
        const float cos_theta_max = sqrtf(fmaxf(0.f, 1 - sin_theta_max_sq));
        return 1 / (2 * K_PI * (1 - cos_theta_max));  // uniform_cone_pdf, 798; never tests the hit (quirk 13)
    }
    // base class: re-intersect the light's OWN shape with isect.spawn_ray(wi)
    const f3 o = offset_ray_origin(p, p_normal, wi);
    struct { float p[4][3]; float n[3]; float radius; float radius_sq; } sh;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 3; ++j) sh.p[k][j] = L.p[k][j];
    for (int j = 0; j < 3; ++j) sh.n[j] = L.n[j];
    sh.radius = L.radius;
    sh.radius_sq = L.radius * L.radius;
    float t;
    if (!shape_hit(sh, L.shape_kind, o, wi, K_INF, t)) return 0.f;
    const f3 hp = o + t * wi;
    f3 ln = ld3(L.n);
    if (L.shape_kind == KY_SHAPE_SPHERE) ln = normalize(hp - c);
    else if (L.shape_kind == KY_SHAPE_RECTANGLE) ln = dot(ln, wi) <= 0 ? ln : -ln;
    float pdf = length_sq(p - hp) / (fabsf(dot(ln, -wi)) * L.area);
    if (isinf(pdf)) pdf = 0.f;
    return pdf;
}

// light_t::sample_Li x4 (2825, 2891, 2964, 3026)
KY_DEV LightSample light_sample_Li(const DLight& L, f3 p, f3 p_normal, float u0, float u1) {
    LightSample s;
    s.position = mk3(0, 0, 0);
    s.wi = mk3(0, 0, 0);
    s.Li = mk3(0, 0, 0);
    s.pdf = 0.f;
    if (L.kind == KY_LIGHT_AREA) {
        f3 lposition, lnormal;
        shape_sample_direction(L, p, p_normal, u0, u1, lposition, lnormal, s.pdf);
        s.position = lposition;
        const f3 dv = lposition - p;
        if (!(s.pdf == 0 || length_sq(dv) == 0)) {
            s.wi = normalize(dv);
            // areal_radiance(light_isect, -wi) with the SAMPLED (stored) normal: one-sided (quirk 5), 2957-2960
            if (dot(lnormal, -s.wi) > 0) s.Li = ld3(L.color);
        }
    } else if (L.kind == KY_LIGHT_POINT) {
        const f3 lp = ld3(L.position);
        s.position = lp;
        s.wi = normalize(lp - p);
        s.pdf = 1.f;
        s.Li = ld3(L.color) / length_sq(lp - p);
    } else if (L.kind == KY_LIGHT_DIRECTION) {
        s.wi = -ld3(L.direction);
        s.position = p + s.wi * 2 * L.world_radius;
        s.pdf = 1;
        s.Li = ld3(L.color);
    } else {  // environment: uniform sphere direction with pdf 1/(2 pi^2 sin(theta)) (quirk 4), 3026-3041
        s.wi = uniform_sphere_sample(u0, u1);
        s.position = p + s.wi * 2 * L.world_radius;
        const float theta = acosf(fminf(fmaxf(s.wi.z, -1.f), 1.f));
        const float sin_theta = sinf(theta);
        s.pdf = 1 / (2 * K_PI * K_PI * sin_theta);
        if (sin_theta == 0) s.pdf = 0;
        s.Li = ld3(L.color);
    }
    return s;
}

// light_t::pdf_Li x4 (2855, 2903, 2984, 3043)
KY_DEV float light_pdf_Li(const DLight& L, f3 p, f3 p_normal, f3 wi) {
    if (L.kind == KY_LIGHT_AREA) return shape_pdf_direction(L, p, p_normal, wi);
    if (L.kind == KY_LIGHT_ENVIRONMENT) {
        const float theta = acosf(fminf(fmaxf(wi.z, -1.f), 1.f));
        const float sin_theta = sinf(theta);
        if (sin_theta == 0) return 0;
        return 1 / (2 * K_PI * K_PI * sin_theta);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// direct lighting (ky.cpp:3834-4088)
// ---------------------------------------------------------------------------------------------

// emission seen along a ray that hit `surface` at `position` (surface_t::intersect 3084 + areal_radiance 2957)
KY_DEV f3 surface_emission(const LdsScene& Lds, int surface, f3 normal, f3 wo) {
    const int al = Lds.hit[surface].area_light;
    f3 e = mk3(0, 0, 0);
    if (al >= 0 && dot(normal, wo) > 0) e = mk3(Lds.light_color[al][0], Lds.light_color[al][1], Lds.light_color[al][2]);
    return e;
}

// BSDF-sampling half of an estimator: by_bsdf (3889-3930, MIS=false) and by_bsdf_mis (3968-4033, MIS=true)
template <bool MIS>
KY_DEV f3 estimate_by_bsdf(const DScene* __restrict__ S, const LdsScene& Lds, const Vertex& v, int li, float u0, float u1) {
    const DLight& L = S->light[li];
    f3 Ld = mk3(0, 0, 0);
    if (L.kind == KY_LIGHT_POINT || L.kind == KY_LIGHT_DIRECTION) return Ld;  // light.is_delta(), 3894 / 3977
    BsdfSample bs = bsdf_sample_local(v.bsdf, to_local(v.frame, v.wo), u0, u1);
    bs.wi = to_world(v.frame, bs.wi);  // 2176
    const f3 f_cos = bs.f * fabsf(dot(bs.wi, v.normal));
    const bool dead = is_black(f_cos) || (MIS ? (bs.pdf <= 0) : (bs.pdf == 0));
    if (!dead) {
        const f3 o = offset_ray_origin(v.position, v.normal, bs.wi);  // isect.spawn_ray, 665-668
        float t = K_INF;
        const int hs = trace_nearest(S, o, bs.wi, t);
        f3 Li = mk3(0, 0, 0);
        if (hs >= 0) {
            if (Lds.hit[hs].area_light == li) {  // 3912 / 3994
                const f3 hp = o + t * bs.wi;
                const f3 hn = hit_normal(Lds.hit[hs], hp, bs.wi);
                Li = surface_emission(Lds, hs, hn, -bs.wi);
            }
        } else if (L.kind == KY_LIGHT_ENVIRONMENT) {
            Li = ld3(L.color);  // light.environmental_radiance(ray), 3918 / 4000
        }
        if (!is_black(Li)) {
            if (MIS) {
                const float light_pdf = light_pdf_Li(L, v.position, v.normal, bs.wi);
                if (light_pdf > 0) Ld = 2.f * (f_cos * Li) / (bs.pdf + light_pdf);  // 4028
            } else {
                Ld = f_cos * Li / bs.pdf;  // 3924
            }
        }
    }
    return Ld;
}

// light-sampling half: by_emitter (3933-3962, MIS=false) and by_emitter_mis (4035-4074, MIS=true)
template <bool MIS>
KY_DEV f3 estimate_by_emitter(const DScene* __restrict__ S, const LdsScene& Lds, const Vertex& v, int li, float u0, float u1) {
    const DLight& L = S->light[li];
    f3 Ld = mk3(0, 0, 0);
    const LightSample ls = light_sample_Li(L, v.position, v.normal, u0, u1);
    const bool dead = is_black(ls.Li) || (MIS ? (ls.pdf <= 0) : (ls.pdf == 0));
    if (!dead) {
        // scene_t::occluded(isect, ls.position), 3187-3201
        const f3 to = ls.position - v.position;
        const f3 dir = normalize(to);
        const float dist = sqrtf(length_sq(v.position - ls.position));
        const f3 o = offset_ray_origin(v.position, v.normal, dir);
        const bool occ = trace_any(S, o, dir, dist - 2e-3f);
        if (!occ) {
            f3 f;
            float bsdf_pdf;
            bsdf_eval_pdf(v.bsdf, to_local(v.frame, v.wo), to_local(v.frame, ls.wi), f, bsdf_pdf);
            const f3 f_cos = f * fabsf(dot(ls.wi, v.normal));
            if (!is_black(f_cos)) {
                const bool delta_light = L.kind == KY_LIGHT_POINT || L.kind == KY_LIGHT_DIRECTION;
                if (!MIS || delta_light) Ld = f_cos * ls.Li / ls.pdf;        // 3956 / 4057
                else Ld = 2 * (f_cos * ls.Li) / (ls.pdf + bsdf_pdf);         // 4070
            }
        }
    }
    return Ld;
}

// sample_all_light, 3834-3872.  Consumes 4 dimensions per light (+2 for the plain bsdf strategy, 3900).
template <bool DEBUG_SAMPLER>
KY_DEV f3 sample_all_light(const DScene* __restrict__ S, const LdsScene& Lds, const Vertex& v, Sampler& smp, int strategy) {
    f3 Ld = mk3(0, 0, 0);
    const int nl = S->n_lights;
    for (int li = 0; li < nl; ++li) {
        // the reference's GCC build draws random_bsdf first, then random_light (3866-3868)
        const float ub0 = sampler_at<DEBUG_SAMPLER>(smp, smp.dim), ub1 = sampler_at<DEBUG_SAMPLER>(smp, smp.dim + 1);
        const float ul0 = sampler_at<DEBUG_SAMPLER>(smp, smp.dim + 2), ul1 = sampler_at<DEBUG_SAMPLER>(smp, smp.dim + 3);
        smp.dim += 4;
        if (strategy == KY_DIRECT_BOTH_MIS) {  // 4076-4088
            const f3 Lb = estimate_by_bsdf<true>(S, Lds, v, li, ub0, ub1);
            const f3 Ll = estimate_by_emitter<true>(S, Lds, v, li, ul0, ul1);
            Ld = Ld + (0.5f * Lb + 0.5f * Ll);
        } else if (strategy == KY_DIRECT_BSDF_MIS) {
            Ld = Ld + estimate_by_bsdf<true>(S, Lds, v, li, ub0, ub1);
        } else if (strategy == KY_DIRECT_LIGHT_MIS) {
            Ld = Ld + estimate_by_emitter<true>(S, Lds, v, li, ul0, ul1);
        } else if (strategy == KY_DIRECT_LIGHT) {
            Ld = Ld + estimate_by_emitter<false>(S, Lds, v, li, ul0, ul1);
        } else if (strategy == KY_DIRECT_BSDF) {
            const int lk = S->light[li].kind;
            if (!(lk == KY_LIGHT_POINT || lk == KY_LIGHT_DIRECTION)) {  // the third float2 is drawn after the delta test (3894-3900)
                const float u0 = sampler_at<DEBUG_SAMPLER>(smp, smp.dim), u1 = sampler_at<DEBUG_SAMPLER>(smp, smp.dim + 1);
                smp.dim += 2;
                Ld = Ld + estimate_by_bsdf<false>(S, Lds, v, li, u0, u1);
            }
        }
        // KY_DIRECT_IDLE: estimate_direct_lighting_idle, 3880-3886
    }
    return Ld;
}

// ---------------------------------------------------------------------------------------------
// one path = one camera sample.  path_tracing_iteration_t::Li (4529-4617), direct_lighting_t::Li
// (4136-4154) and debug_integrator_t::Li (4105-4122) share this state machine: step() advances
// the path by one vertex and returns false when the path has ended (radiance complete in Lo).
// ---------------------------------------------------------------------------------------------
struct PathState {
    f3 o, d;       // current ray
    f3 beta, Lo;
    Sampler smp;
    int bounces;
    bool prev_specular;
};

struct RenderConst {  // wave-uniform launch constants
    int integrator, max_path_depth, strategy;
    uint32_t seed;
    int width, height, spp;
    float inv_spp;
};

template <bool DEBUG_SAMPLER>
KY_DEV void path_begin(PathState& ps, const DScene* __restrict__ S, const RenderConst& rc, int x, int y, int sample) {
    sampler_start(ps.smp, rc.seed, (uint32_t)(y * rc.width + x), (uint32_t)sample);
    // get_camera_sample, 943-946 / 971-974
    const float u0 = sampler_at<DEBUG_SAMPLER>(ps.smp, 0), u1 = sampler_at<DEBUG_SAMPLER>(ps.smp, 1);
    ps.smp.dim = 2;
    generate_ray(S, (float)x + u0, (float)y + u1, ps.o, ps.d);
    ps.beta = mk3(1, 1, 1);
    ps.Lo = mk3(0, 0, 0);
    ps.bounces = 0;
    ps.prev_specular = false;
}

template <bool DEBUG_SAMPLER>
KY_DEV bool path_step(PathState& ps, const DScene* __restrict__ S, const LdsScene& Lds, const RenderConst& rc) {
    float t = K_INF;
    const int hs = trace_nearest(S, ps.o, ps.d, t);  // scene->intersect, 4542
    const bool hit = hs >= 0;

    Vertex v;
    f3 emission = mk3(0, 0, 0);
    if (hit) {
        v.position = ps.o + t * ps.d;
        v.normal = hit_normal(Lds.hit[hs], v.position, ps.d);
        v.wo = -ps.d;
        v.surface = hs;
        emission = surface_emission(Lds, hs, v.normal, v.wo);
    }

    if (rc.integrator == KY_INTEGRATOR_PATH_TRACING_ITERATION || rc.integrator == KY_INTEGRATOR_DIRECT_LIGHTING) {
        if (ps.bounces == 0 || ps.prev_specular) {  // 4548-4559
            if (hit) ps.Lo = ps.Lo + ps.beta * emission;
            else if (S->env_light >= 0) ps.Lo = ps.Lo + ps.beta * ld3(S->light[S->env_light].color);  // environment_lighting, 3231
        }
        if (!hit) return false;  // 4563 / 4141
        if (rc.integrator == KY_INTEGRATOR_PATH_TRACING_ITERATION && ps.bounces >= rc.max_path_depth) return false;  // 4563
    } else if (!hit) {
        return false;  // debug integrators return black on a miss (4121)
    }

    // material->scattering(isect) for the nearest hit (3083)
    const float lobe_u = sampler_at<DEBUG_SAMPLER>(ps.smp, KY_DIM_LOBE + (uint32_t)ps.bounces);
    v.bsdf = make_bsdf(Lds.mat[Lds.hit[hs].material], lobe_u);
    v.frame = make_frame(v.normal);

    if (rc.integrator < KY_INTEGRATOR_DIRECT_LIGHTING) {  // debug_integrator_t, 4110-4118
        if (rc.integrator == KY_INTEGRATOR_POSITION) ps.Lo = normalize(v.position);
        else if (rc.integrator == KY_INTEGRATOR_NORMAL) ps.Lo = normalize(v.normal);
        else {
            float pdf;
            bsdf_eval_pdf(v.bsdf, to_local(v.frame, v.wo), to_local(v.frame, v.normal), ps.Lo, pdf);
        }
        return false;
    }

    const bool delta = bsdf_is_delta(v.bsdf);
    if (!delta) {  // 4571-4576
        const f3 Ld = sample_all_light<DEBUG_SAMPLER>(S, Lds, v, ps.smp, rc.strategy);
        ps.Lo = ps.Lo + ps.beta * Ld;
    }
    if (rc.integrator == KY_INTEGRATOR_DIRECT_LIGHTING) return false;  // 4153

    // sample BSDF to get the new path direction, 4586
    const float u0 = sampler_at<DEBUG_SAMPLER>(ps.smp, ps.smp.dim), u1 = sampler_at<DEBUG_SAMPLER>(ps.smp, ps.smp.dim + 1);
    ps.smp.dim += 2;
    BsdfSample bs = bsdf_sample_local(v.bsdf, to_local(v.frame, v.wo), u0, u1);
    bs.wi = to_world(v.frame, bs.wi);
    if (is_black(bs.f) || bs.pdf == 0.f) return false;  // 4588
    ps.beta = ps.beta * (bs.f * fabsf(dot(bs.wi, v.normal)) / bs.pdf);  // 4592
    ps.prev_specular = (bs.flags & BSDF_SPECULAR) != 0;  // 4596
    ps.o = offset_ray_origin(v.position, v.normal, bs.wi);  // 4597
    ps.d = bs.wi;

    if (ps.bounces > 3) {  // Russian roulette, 4601-4612
        const float q = fmaxf(0.05f, 1 - max3(ps.beta));
        const float u = sampler_at<DEBUG_SAMPLER>(ps.smp, ps.smp.dim);
        ps.smp.dim += 1;
        if (u < q) return false;
        ps.beta = ps.beta * (1 / (1 - q));
    }
    ps.bounces += 1;
    // The vertex at bounces == max_depth can only add emission after a delta bounce (4548, 4563):
    // when the previous bounce was not specular that last traversal cannot change Lo, so skip it.
    if (ps.bounces >= rc.max_path_depth && !ps.prev_specular) return false;
    return true;
}

}  // namespace kyd
