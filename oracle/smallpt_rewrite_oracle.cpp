/*
 * smallpt_rewrite_oracle.cpp -- CPU restatement of smallpt2pbrt/smallpt_rewrite.cpp ("structured smallpt": the pbrt-style
 * double-precision step between smallpt and ky.cpp), the ONE translation unit of the reference that this image's
 * toolchain builds unmodified (oracle/Makefile `ref` -> oracle/_ref/smallpt_rewrite).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ (and nothing under ky_amd/).  It checks variant 1 of the product's fp64 sphere
 * path (ky_amd/csrc/ky_smallpt.hpp, ky_smallpt_params::variant = KY_SP_VARIANT_REWRITE).
 *
 * PARITY PINNED: with rng_mode 1 this file consumes std::mt19937_64 / std::uniform_real_distribution<double> exactly like
 * the reference (a fresh generator seeded 1234 per image row: smallpt_rewrite.cpp:282-317, 1300), in the reference's
 * operation order, and reproduces the reference binary's image BYTE FOR BYTE (tests/test_smallpt_rewrite.py: against
 * oracle/_ref/smallpt_rewrite where it exists, and against tests/golden/smallpt_rewrite_16.npz everywhere).
 * rng_mode 0 replaces only the number source by the HIP path's per-sample splitmix64 streams, which makes the product
 * comparable sample by sample.
 *
 * Every function names the lines of /root/reference/smallpt2pbrt/smallpt_rewrite.cpp it follows.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "../include/kyhip.h"

namespace {

constexpr double kPi = 3.14159265358979323846;      // std::numbers::pi, 33
constexpr double kInvPi = 0.318309886183790671538;  // std::numbers::inv_pi, 34

struct V3 {  // Vector3, 71-119
    double x = 0, y = 0, z = 0;
};
V3 v3(double x, double y, double z) { return V3{x, y, z}; }
V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
V3 operator*(V3 a, double b) { return v3(a.x * b, a.y * b, a.z * b); }
V3 operator/(V3 a, double b) { return v3(a.x / b, a.y / b, a.z / b); }
V3 mul(V3 a, V3 c) { return v3(a.x * c.x, a.y * c.y, a.z * c.z); }                                   // Color * Color, 104
V3 normalize(V3 a) { return a * (1 / std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z)); }              // 90
double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                                 // 91
V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }  // 92
double max_component(V3 c) { return std::max({c.x, c.y, c.z}); }                                    // 106-109
bool is_black(V3 c) { return (c.x <= 0) && (c.y <= 0) && (c.z <= 0); }                              // 111

struct Frame {  // 122-173
    V3 s, t, n;
};
Frame make_frame(V3 normal) {  // Frame(const Normal3&) 131-135 + SetFromZ 161-166
    Frame f;
    f.n = normalize(normal);
    const V3 tmp_s = (std::fabs(f.n.x) > 0.99f) ? v3(0, 1, 0) : v3(1, 0, 0);   // the literal is a float (0.99f), 163
    f.t = normalize(cross(f.n, tmp_s));
    f.s = normalize(cross(f.t, f.n));
    return f;
}
V3 to_local(const Frame& f, V3 w) { return v3(dot(f.s, w), dot(f.t, w), dot(f.n, w)); }            // 139-145
V3 to_world(const Frame& f, V3 l) { return f.s * l.x + f.t * l.y + f.n * l.z; }                    // 147-153

// ---- random numbers -----------------------------------------------------------------------------
struct Rng {
    int mode = 0;
    uint64_t s = 0;                                         // mode 0: splitmix64 (the HIP path's stream)
    std::mt19937_64 engine{1234};                           // mode 1: RNG::rngEngine, 283 / 312
    std::uniform_real_distribution<double> dist{0.0, 1.0};  //         RNG::float01Dist, 316
};
uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void rng_start(Rng& r, uint32_t seed, uint32_t pixel_index, uint32_t sample) {   // same keying as smallpt_oracle.cpp / ky_smallpt.hpp
    r.mode = 0;
    r.s = mix64(((uint64_t)seed << 32) ^ (uint64_t)pixel_index) + (uint64_t)sample * 0xD1B54A32D192ED03ull;
    r.s = mix64(r.s);
}
double uniform_float(Rng& r) {   // RNG::UniformFloat, 299-302
    if (r.mode == 1) return r.dist(r.engine);
    r.s += 0x9E3779B97F4A7C15ull;
    return (double)(mix64(r.s) >> 11) * (1.0 / 9007199254740992.0);
}
// RNG::UniformFloat2 (305-308) is `Float2(UniformFloat(), UniformFloat())`: C++ leaves the order of the two calls open, and
// the reference binary built here (g++ 11, x86-64) evaluates them RIGHT TO LEFT -- the first number drawn becomes .y
// (established by the byte-exact match of the whole image; the other order does not match).
struct V2 {
    double x, y;
};
V2 uniform_float2(Rng& r) {
    const double second_arg = uniform_float(r);
    const double first_arg = uniform_float(r);
    return V2{first_arg, second_arg};
}

// ---- sampling (251-270) -------------------------------------------------------------------------
V3 cosine_sample_hemisphere(V2 u) {
    const double radius = std::sqrt(u.x);         // UniformSampleDisk, 253-255
    const double theta = 2 * kPi * u.y;
    const double px = radius * std::cos(theta), py = radius * std::sin(theta);
    const double z = std::sqrt(std::max(0.0, 1 - px * px - py * py));   // 263
    return v3(px, py, z);
}

// ---- scene (706-787, 1133-1149, 1197-1244) --------------------------------------------------------
struct Hit {
    V3 position, normal, wo, emission;
    int prim = -1;
};

// Sphere::Intersect, 758-782; ray_distance is Ray::distance (in/out)
bool sphere_intersect(const ky_smallpt_sphere& s, V3 origin, V3 direction, double& ray_distance, Hit& isect) {
    const V3 center = v3(s.p[0], s.p[1], s.p[2]);
    const V3 oc = center - origin;
    const double neg_b = dot(oc, direction);
    const double det = neg_b * neg_b - dot(oc, oc) + s.rad * s.rad;
    bool hit = false;
    double distance = 0;
    if (det >= 0) {
        const double sqrt_det = std::sqrt(det);
        const double epsilon = 1e-4;
        if (distance = neg_b - sqrt_det; distance > epsilon && distance < ray_distance) hit = true;
        else if (distance = neg_b + sqrt_det; distance > epsilon && distance < ray_distance) hit = true;
    }
    if (hit) {
        ray_distance = distance;
        const V3 hit_point = origin + direction * distance;   // Ray::operator(), 189-192
        isect.position = hit_point;
        isect.normal = normalize(hit_point - center);
        isect.wo = -direction;
    }
    return hit;
}

struct Scene {
    const ky_smallpt_sphere* s;
    int n, max_depth;
};

// Scene::Intersect (1184-1197) over Primitive::Intersect (1139-1149): every primitive in list order, the ray's distance
// shrinks; emission = AreaLight::Le (1114-1117) of the last (nearest) hit
bool scene_intersect(const Scene& sc, V3 origin, V3 direction, Hit& isect) {
    double distance = INFINITY;
    bool any = false;
    for (int i = 0; i < sc.n; ++i) {
        if (sphere_intersect(sc.s[i], origin, direction, distance, isect)) {
            any = true;
            isect.prim = i;
            const V3 e = v3(sc.s[i].e[0], sc.s[i].e[1], sc.s[i].e[2]);
            const bool is_light = !is_black(e);   // the primitive carries an AreaLight (1241)
            isect.emission = (is_light && dot(isect.normal, isect.wo) > 0) ? e : v3(0, 0, 0);
        }
    }
    return any;
}

struct BsdfSample {  // 794-799
    V3 f, wi;
    double pdf = 0;
};

// Material::Scattering (1047, 1064, 1090) + BSDF::Sample_f (834-840) + the three Sample_f_ (888-904, 918-930, 946-1020)
BsdfSample bsdf_sample(const ky_smallpt_sphere& prim, const Hit& isect, V2 random) {
    const Frame fr = make_frame(isect.normal);
    const V3 wo = to_local(fr, isect.wo);
    const V3 color = v3(prim.c[0], prim.c[1], prim.c[2]);
    BsdfSample sample;
    if (prim.refl == KY_SP_DIFF) {          // LambertionReflection
        sample.wi = cosine_sample_hemisphere(random);
        if (wo.z < 0) sample.wi.z *= -1;
        sample.pdf = (wo.z * sample.wi.z > 0) ? std::abs(sample.wi.z) * kInvPi : 0;   // Pdf_, 883-886
        sample.f = color * kInvPi;                                                    // f_, 881
    } else if (prim.refl == KY_SP_SPEC) {   // SpecularReflection
        sample.wi = v3(-wo.x, -wo.y, wo.z);
        sample.pdf = 1;
        sample.f = color / std::abs(sample.wi.z);
    } else {                                // FresnelSpecular(R = T = color, etaI = 1, etaT = 1.5), 1090 / 1239
        const double eta_i = 1, eta_t = 1.5;
        const V3 normal = v3(0, 0, 1);
        const bool into = dot(normal, wo) > 0;
        const V3 wo_normal = into ? normal : normal * -1;
        const double eta = into ? eta_i / eta_t : eta_t / eta_i;
        const V3 reflect_direction = v3(-wo.x, -wo.y, wo.z);
        const double cos_theta_i = dot(wo, wo_normal);
        const double cos_theta_t2 = 1 - eta * eta * (1 - cos_theta_i * cos_theta_i);
        if (cos_theta_t2 < 0) return sample;   // total internal reflection: f = 0, pdf = 0 (972-975)
        const double cos_theta_t = std::sqrt(cos_theta_t2);
        const V3 refract_direction = normalize(-wo * eta + wo_normal * (cos_theta_i * eta - cos_theta_t));
        const double a = eta_t - eta_i, b = eta_t + eta_i;
        const double R0 = a * a / (b * b);
        const double c = 1 - (into ? cos_theta_i : cos_theta_t);
        const double Re = R0 + (1 - R0) * c * c * c * c * c;
        const double Tr = 1 - Re;
        if (random.x < Re) {
            sample.wi = reflect_direction;
            sample.pdf = Re;
            sample.f = (color * Re) / std::abs(sample.wi.z);
        } else {
            sample.wi = refract_direction;
            sample.pdf = Tr;
            sample.f = (color * Tr) / std::abs(sample.wi.z);
        }
    }
    sample.wi = to_world(fr, sample.wi);
    return sample;
}

// RecursionPathIntegrater::Li, 1345-1372 (recursive, like the reference: the product unrolls it into a loop)
V3 Li(const Scene& sc, V3 origin, V3 direction, Rng& rng, int depth) {
    Hit isect;
    if (!scene_intersect(sc, origin, direction, isect)) return v3(0, 0, 0);
    if (depth > sc.max_depth) return isect.emission;
    BsdfSample bs = bsdf_sample(sc.s[isect.prim], isect, uniform_float2(rng));
    if (is_black(bs.f) || bs.pdf == 0.f) return isect.emission;
    if (++depth > 5) {   // russian roulette on the BSDF VALUE's largest component
        const double max_c = max_component(bs.f);
        if (uniform_float(rng) < max_c) bs.f = bs.f * (1 / max_c);
        else return isect.emission;
    }
    const V3 Lin = Li(sc, isect.position, bs.wi, rng, depth);
    return isect.emission + (mul(bs.f, Lin) * std::abs(dot(bs.wi, isect.normal)) / bs.pdf);
}

struct Camera {  // PerspectiveCamera, 651-694
    V3 position, front, right, up;
    double res_x, res_y;
};
Camera make_camera(int w, int h) {   // main, 1391-1392 + the constructor 654-666
    Camera c;
    c.position = v3(50, 52, -295.6);
    c.front = normalize(v3(0, -0.042612, 1));
    c.res_x = w; c.res_y = h;
    const double tan_fov = std::tan(((kPi / 180) * 53) / 2);   // radians(fov) / 2, 37 / 662
    const V3 up0 = v3(0, 1, 0);
    c.right = normalize(cross(up0, c.front)) * tan_fov * (c.res_x / c.res_y);
    c.up = normalize(cross(c.front, c.right)) * tan_fov;
    return c;
}
void generate_ray(const Camera& c, V2 p_film, V3& origin, V3& direction) {   // 669-677
    const V3 d = c.front + c.right * (p_film.x / c.res_x - 0.5) + c.up * (0.5 - p_film.y / c.res_y);
    origin = c.position + d * 140;
    direction = normalize(d);
}

double clamp01(double x) { return x < 0 ? 0 : x > 1 ? 1 : x; }   // 491

bool valid(const ky_smallpt_sphere* s, int n, const ky_smallpt_params* p) {
    return s && p && n > 0 && p->width > 0 && p->height > 0 && p->samps > 0 && p->max_depth >= 0;
}

}  // namespace

extern "C" {

// Integrater::Render, 1287-1318.  film_rgb: height x width x 3 doubles, row 0 = y 0 = TOP (Film::operator(), 517-520); the
// clamped mean radiance is ADDED (add_color, 522-526).  spp = params->samps (main: argv[1] / 4, 1388).
// rng_mode 1: the reference's own generator, re-created (seed 1234) for every row by Sampler::Clone (1300);
// rng_mode 0: one splitmix64 stream per (seed, pixel, sample).
int kyo_sprw_render(const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p, int rng_mode, double* film_rgb) {
    if (!valid(spheres, n, p) || !film_rgb) return -1;
    const Scene sc{spheres, n, p->max_depth};
    const int w = p->width, h = p->height, spp = p->samps;
    const Camera cam = make_camera(w, h);
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < h; y++) {
        Rng rng;   // a fresh generator per row
        rng.mode = rng_mode ? 1 : 0;
        for (int x = 0; x < w; x++) {
            V3 L;
            for (int s = 0; s < spp; ++s) {
                if (rng_mode == 0) rng_start(rng, p->seed, (uint32_t)(y * w + x), (uint32_t)s);
                const V2 u = uniform_float2(rng);                     // RandomSampler::GetCameraSample, 388-391
                V3 origin, direction;
                generate_ray(cam, V2{(double)x + u.x, (double)y + u.y}, origin, direction);
                L = L + Li(sc, origin, direction, rng, 0) * (1. / spp);   // 1312
            }
            double* c = film_rgb + ((size_t)y * w + x) * 3;
            c[0] += clamp01(L.x); c[1] += clamp01(L.y); c[2] += clamp01(L.z);   // 1316
        }
    }
    return 0;
}

// Li of samples [s0, s0 + count) of pixel (x, y), rng_mode 0 (the twin of kyhip_smallpt_kat_radiance with variant 1)
int kyo_sprw_radiance(const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p, int x, int y, int s0, int count, double* out3) {
    if (!valid(spheres, n, p) || !out3) return -1;
    const Scene sc{spheres, n, p->max_depth};
    const Camera cam = make_camera(p->width, p->height);
    for (int i = 0; i < count; ++i) {
        Rng rng;
        rng_start(rng, p->seed, (uint32_t)(y * p->width + x), (uint32_t)(s0 + i));
        const V2 u = uniform_float2(rng);
        V3 origin, direction;
        generate_ray(cam, V2{(double)x + u.x, (double)y + u.y}, origin, direction);
        const V3 L = Li(sc, origin, direction, rng, 0);
        out3[3 * i] = L.x; out3[3 * i + 1] = L.y; out3[3 * i + 2] = L.z;
    }
    return 0;
}

// GammaEncoding (494) of every float of a film, with the C library's pow like the reference: int(pow(clamp(x), 1 / 2.2) * 255 + .5)
int kyo_sprw_gamma_bytes(const double* film, size_t count, uint8_t* out) {
    if (!film || !out) return -1;
    for (size_t i = 0; i < count; ++i) out[i] = (uint8_t)int(std::pow(clamp01(film[i]), 1 / 2.2) * 255 + .5);
    return 0;
}

// Scene::CreateSmallptScene (1199-1244): smallpt's nine spheres mirrored in z, restated independently of the product's table
int kyo_sprw_scene(ky_smallpt_sphere* out) {
    const double rows[9][10] = {{1e5, 1e5 + 1, 40.8, -81.6, 0, 0, 0, .75, .25, .25},     {1e5, -1e5 + 99, 40.8, -81.6, 0, 0, 0, .25, .25, .75},
                                {1e5, 50, 40.8, -1e5, 0, 0, 0, .75, .75, .75},           {1e5, 50, 40.8, 1e5 - 170, 0, 0, 0, 0, 0, 0},
                                {1e5, 50, 1e5, -81.6, 0, 0, 0, .75, .75, .75},           {1e5, 50, -1e5 + 81.6, -81.6, 0, 0, 0, .75, .75, .75},
                                {16.5, 27, 16.5, -47, 0, 0, 0, .999, .999, .999},        {16.5, 73, 16.5, -78, 0, 0, 0, .999, .999, .999},
                                {600, 50, 681.6 - .27, -81.6, 12, 12, 12, 0, 0, 0}};
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
