/*
 * ky_oracle.cpp -- CPU restatement of ky.cpp's iterative path-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may build, load or call it, and only as the checker.  The product
 * (libkyhip.so) never links or calls anything in this directory.
 *
 * PARITY PINNING STATUS: "parity unpinned" in the strict sense.
 *   - The reference (infancy/ky @ /root/reference) ships no tests, golden vectors or fixtures for
 *     this path.
 *   - The reference cannot be built in this image: ky.cpp needs the C++20/23 headers <format> and
 *     <print> (ky.cpp:13, 20), which neither GCC 11.4's libstdc++ nor the ROCm clang (no libc++
 *     headers) provide, plus an MSVC-only std::exception(const char*) constructor (ky.cpp:81).
 *     Writing stand-in headers is not allowed, so there is no oracle/_ref build.
 *   What this restatement IS pinned against (tests/test_oracle_pins.py):
 *     (1) the reference values SURVEY.md Appendix A records from the surveyor's run of the
 *         reference (camera rays, sphere centres, bounding sphere, plastic lobe probabilities),
 *     (2) the reference's work-per-sample counters recorded in SURVEY.md section 6 / BASELINE.md
 *         (path iterations, traversals, shadow rays, ... per camera sample), statistically,
 *     (3) the images the reference itself published (the png/jpg files under docs/images), statistically, via
 *         block means committed under tests/golden/ (see tests/golden/make_image_fixtures.py).
 *
 * Every function cites the reference lines it restates (file = /root/reference/ky.cpp).
 * Arithmetic is fp32 in the operation order of the reference.  Where the reference calls an
 * unqualified libm name on a float (sqrt at ky.cpp:310, 314, 1373) this restatement uses the
 * float overload, which is what the reference's native toolchain (MSVC: global float overloads in
 * <cmath>) resolves to; GCC's double-then-narrow differs by at most 1 ulp in normalize().
 *
 * Deliberate, observable-behaviour-preserving simplifications:
 *   - material_t::scattering (ky.cpp:3083) is evaluated once for the nearest hit instead of once
 *     per candidate hit; BSDFs of superseded hits and of shadow/MIS rays are never observed.
 *   - The random numbers: the reference uses std::mt19937_64 seeded 1234 and re-seeded per image
 *     row (ky.cpp:833, 3701) plus a private, thread-racy generator inside plastic_material_t
 *     (ky.cpp:2663, 2681).  Neither stream is reproducible on a GPU, so this restatement (and the
 *     HIP path) use the counter-based generator documented in DESIGN.md keyed by
 *     (seed, pixel, sample); the ORDER in which numbers are consumed is the reference's
 *     (SURVEY.md 8(a) "Random-number consumption order"), with the plastic lobe number drawn from the
 *     same stream when a path vertex lands on a plastic surface.
 *
 * Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp -shared).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/kyhip.h"

namespace kyo {

// ============================================================================================
// math, ky.cpp:169-222
// ============================================================================================
constexpr float k_infinity = std::numeric_limits<float>::infinity();
constexpr float k_pi       = 3.14159265358979323846;   // ky.cpp:182 (float)std::numbers::pi
constexpr float k_2pi      = 2.f * k_pi;               // 183
constexpr float k_pi_over2 = k_pi / 2.f;               // 184
constexpr float k_pi_over4 = k_pi / 4.f;               // 185
constexpr float k_inv_pi   = 0.318309886183790671538;  // 186 (float)std::numbers::inv_pi
constexpr float k_inv_2pi  = k_inv_pi / 2.f;           // 187
constexpr float k_inv_4pi  = k_inv_pi / 4.f;           // 188

inline float radians(float degree) { return (k_pi / 180.f) * degree; }  // 190

// is_equal, ky.cpp:212-220 (float branch, epsilon = numeric_limits<float>::epsilon())
inline bool is_equal(float x, float y) {
    const float epsilon = std::numeric_limits<float>::epsilon();
    return std::abs(x - y) <= epsilon * std::max({1.f, std::abs(x), std::abs(y)});
}

// ============================================================================================
// geometry, ky.cpp:226-694
// ============================================================================================
struct color_t {  // 226-270
    float r{}, g{}, b{};
    color_t operator*(float s) const { return {r * s, g * s, b * s}; }
    color_t operator/(float s) const { return {r / s, g / s, b / s}; }
    color_t operator+(color_t c) const { return {r + c.r, g + c.g, b + c.b}; }
    color_t operator*(color_t c) const { return {r * c.r, g * c.g, b * c.b}; }
    color_t& operator+=(color_t c) { r += c.r, g += c.g, b += c.b; return *this; }
    color_t& operator*=(color_t c) { r *= c.r, g *= c.g, b *= c.b; return *this; }
    color_t& operator*=(float s) { r *= s, g *= s, b *= s; return *this; }
    float max_component_value() const { return std::max({r, g, b}); }                     // 244
    float luminance() const { return 0.212671f * r + 0.715160f * g + 0.072169f * b; }     // 249
    bool is_black() const { return (r <= 0) && (g <= 0) && (b <= 0); }                    // 258
};
inline color_t operator*(float s, color_t c) { return {s * c.r, s * c.g, s * c.b}; }      // 242

struct vec2_t { float x{}, y{}; };

struct vec3_t {  // 292-369
    float x{}, y{}, z{};
    vec3_t() = default;
    vec3_t(float x, float y, float z) : x(x), y(y), z(z) {}
    explicit vec3_t(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
    vec3_t operator-() const { return {-x, -y, -z}; }
    vec3_t operator+(vec3_t v) const { return {x + v.x, y + v.y, z + v.z}; }
    vec3_t operator-(vec3_t v) const { return {x - v.x, y - v.y, z - v.z}; }
    vec3_t operator*(float s) const { return {x * s, y * s, z * s}; }
    vec3_t operator/(float s) const { return {x / s, y / s, z / s}; }
    float magnitude_squared() const { return x * x + y * y + z * z; }                     // 311
    float magnitude() const { return std::sqrt(magnitude_squared()); }                    // 310
    vec3_t normalize() const { return *this * (1 / std::sqrt(x * x + y * y + z * z)); }   // 314
    float dot(vec3_t v) const { return x * v.x + y * v.y + z * v.z; }                     // 317
    vec3_t cross(vec3_t v) const {                                                        // 318-329
        return {y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x};
    }
};
inline vec3_t operator*(float s, vec3_t v) { return {v.x * s, v.y * s, v.z * s}; }        // 334
inline float dot(vec3_t u, vec3_t v) { return u.dot(v); }
inline float abs_dot(vec3_t u, vec3_t v) { return std::abs(u.dot(v)); }
inline vec3_t cross(vec3_t u, vec3_t v) { return u.cross(v); }
inline vec3_t normalize(vec3_t v) { return v.normalize(); }
inline vec3_t lerp(vec3_t u, vec3_t v, float t) { return u + t * (v - u); }               // 343
inline vec3_t vmin(vec3_t a, vec3_t b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline vec3_t vmax(vec3_t a, vec3_t b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
inline float distance(vec3_t a, vec3_t b) { return (a - b).magnitude(); }                 // 381
inline float distance_squared(vec3_t a, vec3_t b) { return (a - b).magnitude_squared(); } // 385

inline float spherical_theta(vec3_t v) { return std::acos(std::clamp(v.z, -1.f, 1.f)); }  // 410
// spherical_to_direction with explicit basis, ky.cpp:431-439
inline vec3_t spherical_to_direction(float sin_theta, float cos_theta, float phi, vec3_t x, vec3_t y, vec3_t z) {
    return sin_theta * std::cos(phi) * x + sin_theta * std::sin(phi) * y + cos_theta * z;
}

// bounds3_t, ky.cpp:461-516
struct bounds3_t {
    vec3_t min_, max_;
    bounds3_t() {
        constexpr float lo = std::numeric_limits<float>::lowest();
        constexpr float hi = std::numeric_limits<float>::max();
        min_ = vec3_t(hi, hi, hi);
        max_ = vec3_t(lo, lo, lo);
    }
    bounds3_t(vec3_t p1, vec3_t p2) : min_(vmin(p1, p2)), max_(vmax(p1, p2)) {}          // 477
    bounds3_t join(vec3_t p) const { return bounds3_t(vmin(min_, p), vmax(max_, p)); }    // 485
    bounds3_t join(const bounds3_t& b) const { return bounds3_t(vmin(min_, b.min_), vmax(max_, b.max_)); }
    bool contain(vec3_t p) const {                                                        // 498
        return p.x >= min_.x && p.x <= max_.x && p.y >= min_.y && p.y <= max_.y && p.z >= min_.z && p.z <= max_.z;
    }
    void bounding_sphere(vec3_t* center, float* radius) const {                           // 508-512
        *center = lerp(min_, max_, 0.5f);
        *radius = contain(*center) ? distance(*center, max_) : 0;
    }
};

// frame_t, ky.cpp:526-578
struct frame_t {
    vec3_t s_{1, 0, 0}, t_{0, 1, 0}, n_{0, 0, 1};
    frame_t() = default;
    explicit frame_t(vec3_t n) : n_(n.normalize()) {                                      // 537-541
        vec3_t tmp_s = (std::abs(n_.x) > 0.99f) ? vec3_t(0, 1, 0) : vec3_t(1, 0, 0);      // 568
        t_ = normalize(cross(n_, tmp_s));                                                 // 569
        s_ = normalize(cross(t_, n_));                                                    // 570
    }
    vec3_t to_local(vec3_t w) const { return {dot(s_, w), dot(t_, w), dot(n_, w)}; }      // 545
    vec3_t to_world(vec3_t l) const { return s_ * l.x + t_ * l.y + n_ * l.z; }            // 553
    vec3_t binormal() const { return s_; }
    vec3_t tangent() const { return t_; }
    vec3_t normal() const { return n_; }
};

// ray_t, ky.cpp:582-611.  distance_ is the current tmax and shrinks as hits are found.
struct ray_t {
    vec3_t origin, direction;
    mutable float distance = k_infinity;
    vec3_t operator()(float t) const { return origin + t * direction; }                   // 601
};

// offset_ray_origin, ky.cpp:614-620
inline vec3_t offset_ray_origin(vec3_t position, vec3_t normal, vec3_t direction) {
    vec3_t offset = normal * 1e-2f;
    if (dot(normal, direction) < 0) offset = -offset;
    return position + offset;
}

// ============================================================================================
// sampling functions, ky.cpp:698-822
// ============================================================================================
inline vec2_t concentric_disk_sample(vec2_t random) {                                     // 710-733
    random = vec2_t{2.f * random.x - 1, 2.f * random.y - 1};
    if (random.x == 0 && random.y == 0) return vec2_t{0, 0};
    float radius{}, theta{};
    if (std::abs(random.x) > std::abs(random.y)) {
        radius = random.x;
        theta  = k_pi_over4 * (random.y / random.x);
    } else {
        radius = random.y;
        theta  = k_pi_over2 - k_pi_over4 * (random.x / random.y);
    }
    return vec2_t{std::cos(theta) * radius, std::sin(theta) * radius};
}
inline vec3_t cosine_hemisphere_sample(vec2_t random) {                                   // 737-743
    vec2_t p = concentric_disk_sample(random);
    float z  = std::sqrt(std::max(0.f, 1 - p.x * p.x - p.y * p.y));
    return vec3_t(p.x, p.y, z);
}
inline float cosine_hemisphere_pdf(float cos_theta) { return cos_theta * k_inv_pi; }      // 745
inline vec3_t uniform_sphere_sample(vec2_t random) {                                      // 761-769
    float z      = 1 - 2 * random.x;
    float radius = std::sqrt(std::max(0.f, 1.f - z * z));
    float phi    = 2 * k_pi * random.y;
    return vec3_t(radius * std::cos(phi), radius * std::sin(phi), z);
}
inline float uniform_cone_pdf(float cos_theta_max) { return 1 / (2 * k_pi * (1 - cos_theta_max)); }  // 798
inline vec2_t uniform_triangle_sample(vec2_t random) {                                    // 804-808
    float su0 = std::sqrt(random.x);
    return vec2_t{1 - su0, random.y * su0};
}

// ============================================================================================
// sampler, ky.cpp:829-975 (semantics) on a counter-based generator (DESIGN.md "Random numbers")
// ============================================================================================
inline uint32_t mix32(uint32_t x) {  // "lowbias32" integer finaliser
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
struct sampler_t {
    int      kind = KY_SAMPLER_RANDOM;
    uint32_t s0 = 0, s1 = 1;
    const float* tape = nullptr;   // function-level KATs: the next numbers to hand out, instead of the generator's
    int tape_pos = 0;
    // one camera sample = one stream; sampler_t::start_pixel / next_sample (900-908) select it.
    // (s0, s1) made from (seed, pixel, sample) is the state of a xoroshiro64+ generator (Blackman / Vigna, a = 26, b = 9, c = 13;
    // s1 is made odd, which excludes the all-zero state): s0 a hash of the pixel's key and the sample index, s1 = rotl(s0, 16) ^ key,
    // so two samples share a stream only if s0 and the pixel's key both collide.  (Rounds 1-3: PCG-RXS-M-XS-32.  The reference's own mt19937_64 stream, re-seeded per image row and
    // raced on by its plastic material, is reproduced by neither: DESIGN.md "Random numbers".  No committed fixture holds values
    // of this stream: tests/golden/ is reference-produced data and explicit-input KATs.)
    static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    void start_sample(uint32_t seed, uint32_t pixel_index, uint32_t sample_index) {
        uint32_t h = mix32(pixel_index ^ mix32(seed));
        s0 = mix32(h + sample_index * 0x9E3779B9u);
        s1 = (rotl(s0, 16) ^ h) | 1u;   // round 5: one hash per sample (the device's sampler_start, ky_amd/csrc/ky_device.hpp); two before
    }
    float get_float() {                                                                   // 960
        if (tape) return tape[tape_pos++];
        if (kind == KY_SAMPLER_DEBUG) return 0.5f;                                        // 933-941
        const uint32_t word = s0 + s1;
        s1 ^= s0;
        s0 = rotl(s0, 26) ^ s1 ^ (s1 << 9);
        s1 = rotl(s1, 13);
        return (float)(word >> 9) * (1.0f / 8388608.0f);                                  // [0, 1): the sum's upper 23 bits (exact: the device builds the same value as a mantissa)
    }
    vec2_t get_float2() { float a = get_float(); float b = get_float(); return {a, b}; }  // 965, 856-859
    // get_camera_sample, 943-946 / 971-974
    vec2_t get_camera_sample(vec2_t p_film) {
        vec2_t u = get_float2();
        return {p_film.x + u.x, p_film.y + u.y};
    }
};

// ============================================================================================
// isect_t (642-690) + bsdf (2092-2555) folded into one POD: no heap object per hit
// ============================================================================================
enum bsdf_flags : int { bsdf_none = 0, bsdf_reflection = 1, bsdf_transmission = 2, bsdf_diffuse = 4, bsdf_glossy = 8, bsdf_specular = 16 };  // 2092-2102
enum lobe_kind : int { LOBE_LAMBERT, LOBE_MIRROR, LOBE_GLASS, LOBE_PHONG };

struct bsdf_sample_t {  // 2137-2145
    color_t f{};
    vec3_t  wi{};
    float   pdf{};
    int     bsdf_type{};
    bool is_delta_bsdf() const { return (bsdf_type & bsdf_specular) != 0; }               // 2105, 2144
};

inline float cos_theta(vec3_t w) { return w.z; }
inline float abs_cos_theta(vec3_t w) { return std::abs(w.z); }
inline bool same_hemisphere(vec3_t w, vec3_t wp) { return w.z * wp.z > 0; }               // 1921
inline vec3_t reflect(vec3_t wo, vec3_t normal) { return -wo + 2 * dot(wo, normal) * normal; }  // 1923

inline bool refract(vec3_t wi, vec3_t normal, float eta_ratio, vec3_t* out_wt) {          // 1931-1957
    float cos_theta_i    = dot(normal, wi);
    float sin_theta_i_sq = std::max(0.f, 1 - cos_theta_i * cos_theta_i);
    float sin_theta_t_sq = eta_ratio * eta_ratio * sin_theta_i_sq;
    if (sin_theta_t_sq >= 1) return false;
    float cos_theta_t = std::sqrt(1 - sin_theta_t_sq);
    *out_wt = eta_ratio * -wi + (eta_ratio * cos_theta_i - cos_theta_t) * normal;
    return true;
}

inline float fresnel_dielectric(float cos_theta_i, float eta_i, float eta_t) {            // 1963-1996
    cos_theta_i   = std::clamp(cos_theta_i, -1.f, 1.f);
    bool entering = cos_theta_i > 0.f;
    if (!entering) {
        std::swap(eta_i, eta_t);
        cos_theta_i = std::abs(cos_theta_i);
    }
    float sin_theta_i = std::sqrt(std::max(0.f, 1 - cos_theta_i * cos_theta_i));
    float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1) return 1;
    float cos_theta_t = std::sqrt(std::max(0.f, 1 - sin_theta_t * sin_theta_t));
    float r_para = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    float r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return (r_para * r_para + r_perp * r_perp) / 2;
}

struct bsdf_t {
    int     lobe = LOBE_LAMBERT;
    frame_t shading_frame;          // 2206
    color_t a{}, b{};               // lambert albedo_ / mirror reflectance_ / glass reflectance_, transmittance_ / phong specular_reflectance_
    float   eta_i = 1, eta_t = 1;   // glass
    float   exponent = 0;           // phong

    bool is_delta() const { return lobe == LOBE_MIRROR || lobe == LOBE_GLASS; }           // 2225, 2287, 2350, 2487

    // ---- local-frame lobes ----
    color_t eval_(vec3_t wo, vec3_t wi) const {
        switch (lobe) {
        case LOBE_LAMBERT:                                                                // 2227-2235
            if (!same_hemisphere(wo, wi)) return {};
            return a * k_inv_pi;
        case LOBE_PHONG: {                                                                // 2489-2500
            if (!same_hemisphere(wo, wi)) return {};
            const vec3_t wr       = reflect(wo, vec3_t(0, 0, 1));
            const float cos_alpha = dot(wr, wi);
            const color_t rho     = a * (exponent + 2.f) * k_inv_2pi;
            return rho * std::pow(cos_alpha, exponent);   // cos_alpha is NOT clamped (quirk 6)
        }
        default: return {};                                                               // 2289, 2352
        }
    }
    float pdf_(vec3_t wo, vec3_t wi) const {
        switch (lobe) {
        case LOBE_LAMBERT:                                                                // 2237-2240
            return same_hemisphere(wo, wi) ? cosine_hemisphere_pdf(abs_cos_theta(wi)) : 0;
        case LOBE_PHONG: {                                                                // 2502-2508, 2545-2550
            const vec3_t wr      = reflect(wo, vec3_t(0, 0, 1));
            const float cosTheta = std::max(0.f, dot(wr, wi));
            return (exponent + 1.f) * std::pow(cosTheta, exponent) * k_inv_2pi;           // no hemisphere test (quirk 6)
        }
        default: return 0;                                                                // 2290, 2353
        }
    }
    bsdf_sample_t sample_(vec3_t wo, vec2_t random) const {
        bsdf_sample_t sample;
        switch (lobe) {
        case LOBE_LAMBERT:                                                                // 2242-2257
            sample.wi = cosine_hemisphere_sample(random);
            if (wo.z < 0) sample.wi.z *= -1;
            sample.f         = eval_(wo, sample.wi);
            sample.pdf       = pdf_(wo, sample.wi);
            sample.bsdf_type = bsdf_reflection | bsdf_diffuse;
            break;
        case LOBE_MIRROR:                                                                 // 2292-2307
            sample.wi        = vec3_t(-wo.x, -wo.y, wo.z);
            sample.f         = a / abs_cos_theta(sample.wi);
            sample.pdf       = 1;
            sample.bsdf_type = bsdf_reflection | bsdf_specular;
            break;
        case LOBE_GLASS: {                                                                // 2355-2412
            float reflect_percent = fresnel_dielectric(cos_theta(wo), eta_i, eta_t);
            float refract_percent = 1 - reflect_percent;
            float Pr_reflect = reflect_percent, Pr_refract = refract_percent;
            if (random.x < Pr_reflect) {
                sample.wi        = vec3_t(-wo.x, -wo.y, wo.z);
                sample.pdf       = Pr_reflect;
                sample.f         = (a * reflect_percent) / abs_cos_theta(sample.wi);
                sample.bsdf_type = bsdf_reflection | bsdf_specular;
            } else {
                vec3_t normal(0, 0, 1);
                bool into        = normal.dot(wo) > 0;
                vec3_t wo_normal = into ? normal : normal * -1;
                float eta        = into ? eta_i / eta_t : eta_t / eta_i;
                if (refract(wo, wo_normal, eta, &sample.wi)) {
                    sample.pdf       = Pr_refract;
                    sample.f         = (b * refract_percent) / abs_cos_theta(sample.wi);
                    sample.bsdf_type = bsdf_transmission | bsdf_specular;
                } else {
                    sample.f = color_t();  // total internal reflection: pdf stays 0, path terminates at 4588
                }
            }
            break;
        }
        case LOBE_PHONG: {                                                                // 2510-2529
            // cosine_hemisphere_sample_phong, 2533-2543
            const float phi = 2.f * k_pi * random.x;
            const float ct  = std::pow(random.y, 1.f / (exponent + 1.f));
            const float st  = std::sqrt(1.f - ct * ct);
            sample.wi       = vec3_t(std::cos(phi) * st, std::sin(phi) * st, ct);
            const vec3_t wr = reflect(wo, vec3_t(0, 0, 1));
            frame_t frame{wr};
            sample.wi = frame.to_world(sample.wi);
            if (wo.z < 0) sample.wi.z *= -1;
            sample.f         = eval_(wo, sample.wi);
            sample.pdf       = pdf_(wo, sample.wi);
            sample.bsdf_type = bsdf_reflection | bsdf_glossy;
            break;
        }
        }
        return sample;
    }

    // ---- world-space wrappers, 2162-2179 ----
    color_t eval(vec3_t world_wo, vec3_t world_wi) const { return eval_(shading_frame.to_local(world_wo), shading_frame.to_local(world_wi)); }
    float pdf(vec3_t world_wo, vec3_t world_wi) const { return pdf_(shading_frame.to_local(world_wo), shading_frame.to_local(world_wi)); }
    bsdf_sample_t sample(vec3_t world_wo, vec2_t random) const {
        bsdf_sample_t s = sample_(shading_frame.to_local(world_wo), random);
        s.wi = shading_frame.to_world(s.wi);  // 2176
        return s;
    }
};

struct isect_t {  // 642-690
    vec3_t  position{}, normal{}, wo{};
    int     surface = -1;
    bsdf_t  bsdf;
    color_t emission{};
    ray_t spawn_ray(vec3_t direction) const { return ray_t{offset_ray_origin(position, normal, direction), direction, k_infinity}; }  // 665-668
};

// ============================================================================================
// shapes, ky.cpp:1009-1519
// ============================================================================================
constexpr float shape_epsilon = 1e-3f;  // 1093

struct shape_t {
    int    kind;
    vec3_t p0, p1, p2, p3, normal_;
    float  radius_ = 0, radius_sq_ = 0;

    explicit shape_t(const ky_shape& s)
        : kind(s.kind), p0(s.p[0]), p1(s.p[1]), p2(s.p[2]), p3(s.p[3]), normal_(s.normal), radius_(s.radius),
          radius_sq_(s.radius * s.radius) {}                                              // 1332

    // ---- intersect: on a hit shrinks ray.distance and fills position / normal / wo ----
    bool intersect(const ray_t& ray, isect_t* out) const {
        switch (kind) {
        case KY_SHAPE_DISK: {                                                             // 1111-1132
            if (is_equal(dot(ray.direction, normal_), 0.f)) return false;
            const vec3_t op      = p0 - ray.origin;
            const float distance = dot(normal_, op) / dot(normal_, ray.direction);
            if ((distance > shape_epsilon) && (distance < ray.distance)) {
                vec3_t hit_point = ray(distance);
                if (kyo::distance(p0, hit_point) <= radius_) {
                    ray.distance = distance;
                    out->position = hit_point; out->normal = normal_; out->wo = -ray.direction;
                    return true;
                }
            }
            return false;
        }
        case KY_SHAPE_TRIANGLE: {                                                         // 1179-1215
            const vec3_t oa = p0 - ray.origin, ob = p1 - ray.origin, oc = p2 - ray.origin;
            const vec3_t v0 = cross(oc, ob), v1 = cross(ob, oa), v2 = cross(oa, oc);
            const float v0d = dot(v0, ray.direction), v1d = dot(v1, ray.direction), v2d = dot(v2, ray.direction);
            if (((v0d < 0.f) && (v1d < 0.f) && (v2d < 0.f)) || ((v0d >= 0.f) && (v1d >= 0.f) && (v2d >= 0.f))) {
                const float distance = dot(normal_, oa) / dot(normal_, ray.direction);
                if ((distance > shape_epsilon) && (distance < ray.distance)) {
                    ray.distance = distance;
                    out->position = ray(distance); out->normal = normal_; out->wo = -ray.direction;  // normal NOT flipped
                    return true;
                }
            }
            return false;
        }
        case KY_SHAPE_RECTANGLE: {                                                        // 1261-1297
            const vec3_t oa = p0 - ray.origin, ob = p1 - ray.origin, oc = p2 - ray.origin, od = p3 - ray.origin;
            const vec3_t v0 = cross(oc, ob), v1 = cross(ob, oa), v2 = cross(oa, od), v3 = cross(od, oc);
            const float v0d = dot(v0, ray.direction), v1d = dot(v1, ray.direction), v2d = dot(v2, ray.direction),
                        v3d = dot(v3, ray.direction);
            if (((v0d < 0.f) && (v1d < 0.f) && (v2d < 0.f) && (v3d < 0.f)) ||
                ((v0d >= 0.f) && (v1d >= 0.f) && (v2d >= 0.f) && (v3d >= 0.f))) {
                const float distance = dot(normal_, oa) / dot(normal_, ray.direction);
                if ((distance > shape_epsilon) && (distance < ray.distance)) {
                    ray.distance = distance;
                    out->position = ray(distance);
                    out->normal   = dot(normal_, ray.direction) <= 0 ? normal_ : -normal_;  // 1289: faces the ray
                    out->wo       = -ray.direction;
                    return true;
                }
            }
            return false;
        }
        case KY_SHAPE_SPHERE: {                                                           // 1336-1393
            vec3_t oc   = p0 - ray.origin;
            float neg_b = dot(oc, ray.direction);
            float discr = neg_b * neg_b - dot(oc, oc) + radius_sq_;
            float distance = 0;
            bool hit = false;
            if (discr >= 0) {
                float sqrt_discr = std::sqrt(discr);
                if (distance = neg_b - sqrt_discr; distance > shape_epsilon && distance < ray.distance) hit = true;
                else if (distance = neg_b + sqrt_discr; distance > shape_epsilon && distance < ray.distance) hit = true;
            }
            if (hit) {
                ray.distance = distance;
                vec3_t hit_point = ray(distance);
                out->position = hit_point; out->normal = (hit_point - p0).normalize(); out->wo = -ray.direction;
            }
            return hit;
        }
        }
        return false;
    }

    bounds3_t world_bound() const {
        switch (kind) {
        case KY_SHAPE_DISK: {                                                             // 1134-1139
            frame_t frame{normal_};
            vec3_t offset = frame.binormal() * radius_ + frame.tangent() * radius_;
            return bounds3_t(p0 - offset, p0 + offset);
        }
        case KY_SHAPE_TRIANGLE: return bounds3_t(p0, p1).join(p2);                         // 1217
        case KY_SHAPE_RECTANGLE: return bounds3_t(p0, p1).join(p2).join(p3);               // 1299
        default: { vec3_t half(radius_, radius_, radius_); return bounds3_t(p0 + half, p0 - half); }  // 1395
        }
    }

    float area() const {
        switch (kind) {
        case KY_SHAPE_DISK: return k_pi * radius_ * radius_;                               // 1141
        case KY_SHAPE_TRIANGLE: return 0.5f * cross(p1 - p0, p2 - p0).magnitude();         // 1222
        case KY_SHAPE_RECTANGLE: return cross(p0 - p1, p2 - p1).magnitude();               // 1304
        default: return 4 * k_pi * radius_sq_;                                             // 1401
        }
    }

    // sample_position: returns position+normal of a point on the shape, and its area pdf
    void sample_position(vec2_t random, vec3_t* position, vec3_t* normal, float* area_pdf) const {
        switch (kind) {
        case KY_SHAPE_DISK: {                                                             // 1144-1156
            frame_t frame{normal_};
            vec2_t sp = concentric_disk_sample(random);
            *position = p0 + radius_ * (frame.binormal() * sp.x + frame.tangent() * sp.y);
            *normal   = normalize(normal_);
            break;
        }
        case KY_SHAPE_TRIANGLE: {                                                         // 1225-1235
            vec2_t b  = uniform_triangle_sample(random);
            *position = b.x * p0 + b.y * p1 + (1 - b.x - b.y) * p2;
            *normal   = normal_;
            break;
        }
        case KY_SHAPE_RECTANGLE:                                                          // 1307-1315
            *position = p1 + (p0 - p1) * random.x + (p2 - p1) * random.y;
            *normal   = normalize(normal_);
            break;
        default: {                                                                        // 1404-1416
            vec3_t direction = uniform_sphere_sample(random);
            *position = p0 + radius_ * direction;
            *normal   = normalize(direction);
            break;
        }
        }
        *area_pdf = 1 / area();
    }

    // shape_t::sample_direction, 1028-1051 (base) / sphere_t override 1419-1501
    void sample_direction(const isect_t& isect, vec2_t random, vec3_t* lposition, vec3_t* lnormal, float* solid_angle_pdf) const {
        if (kind != KY_SHAPE_SPHERE) {
            float area_pdf{};
            sample_position(random, lposition, lnormal, &area_pdf);
            vec3_t wi = *lposition - isect.position;
            if (wi.magnitude_squared() == 0) {
                *solid_angle_pdf = 0;
            } else {
                wi = normalize(wi);
                *solid_angle_pdf = area_pdf * distance_squared(*lposition, isect.position) / abs_dot(*lnormal, -wi);
                if (std::isinf(*solid_angle_pdf)) *solid_angle_pdf = 0.f;
            }
            return;
        }
        // sphere: shade point inside -> area sampling (uses isect.normal: quirk, 1436)
        if (distance_squared(isect.position, p0) <= radius_ * radius_) {                  // 1422-1443
            float area_pdf{};
            sample_position(random, lposition, lnormal, &area_pdf);
            vec3_t wi = *lposition - isect.position;
            if (wi.magnitude_squared() == 0) {
                *solid_angle_pdf = 0;
            } else {
                wi = normalize(wi);
                *solid_angle_pdf = area_pdf * distance_squared(*lposition, isect.position) / abs_dot(isect.normal, -wi);
            }
            if (std::isinf(*solid_angle_pdf)) *solid_angle_pdf = 0.f;
            return;
        }
        // outside: uniform cone sampling, 1458-1500
        float dist     = distance(isect.position, p0);
        float inv_dist = 1 / dist;
        float sin_theta_max     = radius_ * inv_dist;
        float sin_theta_max_sq  = sin_theta_max * sin_theta_max;
        float inv_sin_theta_max = 1 / sin_theta_max;
        float cos_theta_max     = std::sqrt(std::max(0.f, 1 - sin_theta_max_sq));
        float cos_theta    = (cos_theta_max - 1) * random.x + 1;
        float sin_theta_sq = 1 - cos_theta * cos_theta;
        if (sin_theta_max_sq < 0.00068523f) {
            sin_theta_sq = sin_theta_max_sq * random.x;
            cos_theta    = std::sqrt(1 - sin_theta_sq);
        }
        float cos_alpha = sin_theta_sq * inv_sin_theta_max +
                          cos_theta * std::sqrt(std::max(0.f, 1.f - sin_theta_sq * inv_sin_theta_max * inv_sin_theta_max));
        float sin_alpha = std::sqrt(std::max(0.f, 1.f - cos_alpha * cos_alpha));
        float phi       = random.y * 2 * k_pi;
        vec3_t normal = (p0 - isect.position) * inv_dist;
        frame_t frame{normal};
        vec3_t world_normal = spherical_to_direction(sin_alpha, cos_alpha, phi, -frame.binormal(), -frame.tangent(), -frame.normal());
        *lposition       = p0 + radius_ * vec3_t(world_normal.x, world_normal.y, world_normal.z);
        *lnormal         = world_normal;
        *solid_angle_pdf = 1 / (2 * k_pi * (1 - cos_theta_max));
    }

    // shape_t::pdf_direction, 1055-1090 (base) / sphere_t override 1503-1513
    float pdf_direction(const isect_t& isect, vec3_t world_wi) const {
        if (kind == KY_SHAPE_SPHERE && !(distance_squared(isect.position, p0) <= radius_ * radius_)) {
            float sin_theta_max_sq = radius_ * radius_ / distance_squared(isect.position, p0);
            float cos_theta_max    = std::sqrt(std::max(0.f, 1 - sin_theta_max_sq));
            return uniform_cone_pdf(cos_theta_max);   // never tests that wi hits the sphere (quirk 13)
        }
        ray_t ray = isect.spawn_ray(world_wi);
        isect_t light_isect;
        if (!intersect(ray, &light_isect)) return 0.f;
        float pdf = distance_squared(isect.position, light_isect.position) / (abs_dot(light_isect.normal, -world_wi) * area());
        if (std::isinf(pdf)) pdf = 0.f;
        return pdf;
    }
};

// ============================================================================================
// lights, ky.cpp:2692-3062
// ============================================================================================
struct light_sample_t { vec3_t position{}, wi{}; float pdf{}; color_t Li{}; };            // 2744-2759

struct scene_t;

struct light_t {
    int     kind;
    int     shape;
    color_t color;
    vec3_t  world_position, world_direction;
    float   world_radius;
    bool is_delta() const { return kind == KY_LIGHT_POINT || kind == KY_LIGHT_DIRECTION; }  // 2819, 2880, 2935, 3009
};

// ============================================================================================
// scene, ky.cpp:3071-3237
// ============================================================================================
struct counters_t {
    uint64_t camera_samples = 0, traversals = 0, shadow_rays = 0, primitive_tests = 0, nee_vertices = 0, light_estimates = 0,
             bsdf_path_samples = 0, path_iterations = 0, mis_bsdf_rays = 0, rr_draws = 0, shadow_occluded = 0;
    void add(const counters_t& o) {
        camera_samples += o.camera_samples; traversals += o.traversals; shadow_rays += o.shadow_rays;
        primitive_tests += o.primitive_tests; nee_vertices += o.nee_vertices; light_estimates += o.light_estimates;
        bsdf_path_samples += o.bsdf_path_samples; path_iterations += o.path_iterations; mis_bsdf_rays += o.mis_bsdf_rays;
        rr_draws += o.rr_draws; shadow_occluded += o.shadow_occluded;
    }
};

struct scene_t {
    std::vector<shape_t>     shapes;
    std::vector<ky_material> materials;
    std::vector<light_t>     lights;
    std::vector<ky_surface>  surfaces;
    int                      environment_light = -1;
    ky_camera                camera;

    explicit scene_t(const ky_scene& s) : environment_light(s.environment_light), camera(s.camera) {
        for (int i = 0; i < s.shape_count; ++i) shapes.emplace_back(s.shapes[i]);
        materials.assign(s.materials, s.materials + s.material_count);
        surfaces.assign(s.surfaces, s.surfaces + s.surface_count);
        for (int i = 0; i < s.light_count; ++i) {
            const ky_light& l = s.lights[i];
            lights.push_back(light_t{l.kind, l.shape, color_t{l.color[0], l.color[1], l.color[2]}, vec3_t(l.position),
                                     vec3_t(l.direction), l.world_radius});
        }
    }

    // material_t::scattering x4, ky.cpp:2587, 2604, 2628, 2661
    void scattering(isect_t* isect, float lobe_random) const {
        const ky_material& m = materials[surfaces[isect->surface].material];
        bsdf_t& b = isect->bsdf;
        b.shading_frame = frame_t(isect->normal);
        color_t c0{m.color0[0], m.color0[1], m.color0[2]}, c1{m.color1[0], m.color1[1], m.color1[2]};
        switch (m.kind) {
        case KY_MATERIAL_MATTE: b.lobe = LOBE_LAMBERT; b.a = c0; break;
        case KY_MATERIAL_MIRROR: b.lobe = LOBE_MIRROR; b.a = c0; break;
        case KY_MATERIAL_GLASS: b.lobe = LOBE_GLASS; b.eta_i = 1; b.eta_t = m.eta; b.a = c0; b.b = c1; break;  // 2630
        case KY_MATERIAL_PLASTIC:                                                          // 2663-2671
            if (lobe_random < m.specular_probability) { b.lobe = LOBE_PHONG; b.a = c1 / m.specular_probability; b.exponent = m.exponent; }
            else { b.lobe = LOBE_LAMBERT; b.a = c0 / m.diffuse_probability; }
            break;
        }
    }

    // area_light_t::areal_radiance, 2957-2960
    color_t areal_radiance(int light, vec3_t light_normal, vec3_t wo) const {
        return (dot(light_normal, wo) > 0) ? lights[light].color : color_t();
    }

    // scene_t::intersect (3172-3184) + surface_t::intersect (3077-3088).
    // with_bsdf=false is used where the reference discards the BSDF (shadow rays, MIS light lookups).
    // `sampler` non-null: build the BSDF of the nearest hit; plastic_material_t::scattering then draws its lobe number
    // (2663) from the path's own stream.  Null where the reference discards the BSDF (shadow rays, MIS light lookups).
    bool intersect(const ray_t& ray, isect_t* isect, counters_t* c, sampler_t* sampler = nullptr) const {
        bool is_hit = false;
        int surface_num = (int)surfaces.size();
        if (c) { c->traversals++; c->primitive_tests += surface_num; }
        for (int i = 0; i < surface_num; ++i) {
            if (shapes[surfaces[i].shape].intersect(ray, isect)) {
                isect->surface = i;
                is_hit = true;
            }
        }
        if (is_hit) {
            const ky_surface& s = surfaces[isect->surface];
            if (sampler) {
                const bool plastic = materials[s.material].kind == KY_MATERIAL_PLASTIC;
                scattering(isect, plastic ? sampler->get_float() : 0.f);
            }
            isect->emission = s.area_light >= 0 ? areal_radiance(s.area_light, isect->normal, isect->wo) : color_t{};  // 3084
        }
        return is_hit;
    }

    // scene_t::occluded, 3187-3201
    bool occluded(vec3_t position, vec3_t normal, vec3_t direction, float dist, counters_t* c) const {
        ray_t ray{offset_ray_origin(position, normal, direction), direction, dist - 2e-3f};
        isect_t unused;
        if (c) c->shadow_rays++;
        bool occ = intersect(ray, &unused, c);
        if (c && occ) c->shadow_occluded++;
        return occ;
    }
    bool occluded(const isect_t& isect1, vec3_t isect2, counters_t* c) const {
        return occluded(isect1.position, isect1.normal, normalize(isect2 - isect1.position), distance(isect1.position, isect2), c);
    }

    // scene_t::environment_lighting, 3231-3237
    color_t environment_lighting() const { return environment_light >= 0 ? lights[environment_light].color : color_t{}; }

    // light_t::sample_Li, 2825 (point) / 2891 (direction) / 2964 (area) / 3026 (environment)
    light_sample_t sample_Li(int li, const isect_t& isect, vec2_t random) const {
        const light_t& light = lights[li];
        light_sample_t sample;
        switch (light.kind) {
        case KY_LIGHT_POINT:
            sample.position = light.world_position;
            sample.wi  = normalize(light.world_position - isect.position);
            sample.pdf = 1.f;
            sample.Li  = light.color / distance_squared(light.world_position, isect.position);
            break;
        case KY_LIGHT_DIRECTION:
            sample.wi       = -light.world_direction;
            sample.position = isect.position + sample.wi * 2 * light.world_radius;
            sample.pdf      = 1;
            sample.Li       = light.color;
            break;
        case KY_LIGHT_AREA: {
            vec3_t lposition, lnormal;
            shapes[light.shape].sample_direction(isect, random, &lposition, &lnormal, &sample.pdf);
            sample.position = lposition;
            if (sample.pdf == 0 || (lposition - isect.position).magnitude_squared() == 0) {
                sample.Li = color_t{};
            } else {
                sample.wi = normalize(lposition - isect.position);
                sample.Li = areal_radiance(li, lnormal, -sample.wi);   // one-sided: stored normal (quirk 5)
            }
            break;
        }
        case KY_LIGHT_ENVIRONMENT: {
            sample.wi       = uniform_sphere_sample(random);
            sample.position = isect.position + sample.wi * 2 * light.world_radius;
            float theta     = spherical_theta(sample.wi);
            float sin_theta = std::sin(theta);
            sample.pdf      = 1 / (2 * k_pi * k_pi * sin_theta);       // quirk 4
            if (sin_theta == 0) sample.pdf = 0;
            sample.Li = light.color;
            break;
        }
        }
        return sample;
    }

    // light_t::pdf_Li, 2855 / 2903 / 2984 / 3043
    float pdf_Li(int li, const isect_t& isect, vec3_t world_wi) const {
        const light_t& light = lights[li];
        switch (light.kind) {
        case KY_LIGHT_AREA: return shapes[light.shape].pdf_direction(isect, world_wi);
        case KY_LIGHT_ENVIRONMENT: {
            float theta     = spherical_theta(world_wi);
            float sin_theta = std::sin(theta);
            if (sin_theta == 0) return 0;
            return 1 / (2 * k_pi * k_pi * sin_theta);
        }
        default: return 0;
        }
    }

    // light_t::environmental_radiance, 2788 (base: black) / 3020 (environment)
    color_t environmental_radiance(int li) const { return lights[li].kind == KY_LIGHT_ENVIRONMENT ? lights[li].color : color_t{}; }
};

// camera_t::generate_ray, ky.cpp:1884-1892
inline ray_t generate_ray(const ky_camera& cam, vec2_t p_film) {
    vec3_t front(cam.front), right(cam.right), up(cam.up);
    vec3_t direction = front + right * (float)(p_film.x / cam.resolution[0] - 0.5) + up * (float)(0.5 - p_film.y / cam.resolution[1]);
    return ray_t{vec3_t(cam.position), direction.normalize(), k_infinity};
}

// ============================================================================================
// integrators, ky.cpp:3679-4618
// ============================================================================================
struct integrator_t {
    const scene_t* scene;
    int kind, max_path_depth, direct_sample;
    // optional per-vertex trace (kyo_trace_li; the analogue of LOG_VAST, ky.cpp:4578): 26 floats per vertex
    mutable std::vector<float>* trace = nullptr;
    // which per-light estimates of the last sample_all_light call were non-black: bit li = the BSDF-sampling half,
    // bit 16 + li = the light-sampling half (the discrete outcome of the carrier / occlusion / zero-value tests)
    mutable unsigned decisions = 0;

    // estimate_direct_lighting_by_bsdf, 3889-3930
    color_t by_bsdf(const isect_t& isect, int li, sampler_t& sampler, counters_t* c) const {
        if (scene->lights[li].is_delta()) return {};
        if (isect.bsdf.is_delta()) return {};
        bsdf_sample_t bs = isect.bsdf.sample(isect.wo, sampler.get_float2());   // draws a THIRD float2 (3900)
        color_t f_cos = bs.f * abs_dot(bs.wi, isect.normal);
        if (f_cos.is_black() || bs.pdf == 0) return {};
        ray_t ray = isect.spawn_ray(bs.wi);
        isect_t light_isect;
        if (c) c->mis_bsdf_rays++;
        bool is_hit_light = scene->intersect(ray, &light_isect, c);
        color_t Li{};
        if (is_hit_light) {
            if (scene->surfaces[light_isect.surface].area_light == li) Li = light_isect.emission;
        } else {
            Li = scene->environmental_radiance(li);
        }
        if (Li.is_black()) return {};
        return f_cos * Li / bs.pdf;
    }

    // estimate_direct_lighting_by_emitter, 3933-3962
    color_t by_emitter(const isect_t& isect, int li, vec2_t random_light, counters_t* c) const {
        if (isect.bsdf.is_delta()) return {};
        light_sample_t ls = scene->sample_Li(li, isect, random_light);
        if (ls.Li.is_black() || ls.pdf == 0) return {};
        if (scene->occluded(isect, ls.position, c)) return {};
        color_t f_cos = isect.bsdf.eval(isect.wo, ls.wi) * abs_dot(ls.wi, isect.normal);
        if (f_cos.is_black()) return {};
        return f_cos * ls.Li / ls.pdf;
    }

    // estimate_direct_lighting_by_bsdf_mis, 3968-4033
    color_t by_bsdf_mis(const isect_t& isect, int li, vec2_t random_bsdf, counters_t* c) const {
        bool is_specular = isect.bsdf.is_delta();
        if (is_specular) return {};                  // skip_specular is always true on this path (4575)
        if (scene->lights[li].is_delta()) return {};
        bsdf_sample_t bs = isect.bsdf.sample(isect.wo, random_bsdf);
        color_t f_cos = bs.f * abs_dot(bs.wi, isect.normal);
        if (f_cos.is_black() || bs.pdf <= 0) return {};
        ray_t ray = isect.spawn_ray(bs.wi);
        isect_t light_isect;
        if (c) c->mis_bsdf_rays++;
        bool is_hit_light = scene->intersect(ray, &light_isect, c);
        color_t Li{};
        if (is_hit_light) {
            if (scene->surfaces[light_isect.surface].area_light == li) Li = light_isect.emission;   // 3994
        } else {
            Li = scene->environmental_radiance(li);                                                // 4000
        }
        if (Li.is_black()) return {};
        color_t Ld{};
        float light_pdf = scene->pdf_Li(li, isect, bs.wi);
        if (light_pdf > 0) Ld = 2.f * (f_cos * Li) / (bs.pdf + light_pdf);                          // 4028
        return Ld;
    }

    // estimate_direct_lighting_by_emitter_mis, 4035-4074
    color_t by_emitter_mis(const isect_t& isect, int li, vec2_t random_light, counters_t* c) const {
        if (isect.bsdf.is_delta()) return {};
        light_sample_t ls = scene->sample_Li(li, isect, random_light);
        if (ls.Li.is_black() || ls.pdf <= 0) return {};
        if (scene->occluded(isect, ls.position, c)) return {};
        color_t f_cos = isect.bsdf.eval(isect.wo, ls.wi) * abs_dot(ls.wi, isect.normal);
        if (f_cos.is_black()) return {};
        color_t Ld{};
        if (scene->lights[li].is_delta()) {
            Ld = f_cos * ls.Li / ls.pdf;
        } else {
            float bsdf_pdf = isect.bsdf.pdf(isect.wo, ls.wi);
            Ld = 2 * (f_cos * ls.Li) / (ls.pdf + bsdf_pdf);                                         // 4070
        }
        return Ld;
    }

    // sample_all_light, 3834-3872.  Returns false for an illegal strategy value (bad_function_call, quirk 10).
    color_t sample_all_light(const isect_t& isect, sampler_t& sampler, counters_t* c) const {
        color_t Ld;
        decisions = 0;
        if (c) c->nee_vertices++;
        const int n = (int)scene->lights.size();
        for (int li = 0; li < n; ++li) {
            // 3866-3868: both get_float2() calls are argument expressions; the GCC build of the
            // reference evaluates them right-to-left, so the FIRST draw is random_bsdf.
            vec2_t random_bsdf  = sampler.get_float2();
            vec2_t random_light = sampler.get_float2();
            if (c) c->light_estimates++;
            color_t Lb, Ll;
            switch (direct_sample) {
            case KY_DIRECT_IDLE: break;                                                             // 3880-3886
            case KY_DIRECT_BSDF: Lb = by_bsdf(isect, li, sampler, c); Ld += Lb; break;
            case KY_DIRECT_LIGHT: Ll = by_emitter(isect, li, random_light, c); Ld += Ll; break;
            case KY_DIRECT_BSDF_MIS: Lb = by_bsdf_mis(isect, li, random_bsdf, c); Ld += Lb; break;
            case KY_DIRECT_LIGHT_MIS: Ll = by_emitter_mis(isect, li, random_light, c); Ld += Ll; break;
            case KY_DIRECT_BOTH_MIS: {                                                              // 4076-4088
                Lb = by_bsdf_mis(isect, li, random_bsdf, c);
                Ll = by_emitter_mis(isect, li, random_light, c);
                Ld += 0.5f * Lb + 0.5f * Ll;
                break;
            }
            }
            if (li < 16) decisions |= (Lb.is_black() ? 0u : 1u << li) | (Ll.is_black() ? 0u : 1u << (16 + li));
        }
        return Ld;
    }

    // path_tracing_iteration_t::Li, 4529-4617
    color_t Li_path(ray_t ray, sampler_t& sampler, counters_t* c) const {
        color_t Lo{};
        color_t beta{1, 1, 1};
        bool is_prev_specular = false;
        for (int bounces = 0;; ++bounces) {
            isect_t isect;
            if (c) c->path_iterations++;
            bool hit = scene->intersect(ray, &isect, c, &sampler);
            if (bounces == 0 || is_prev_specular) {
                if (hit) Lo += beta * isect.emission;
                else Lo += beta * scene->environment_lighting();
            }
            if (!hit || bounces >= max_path_depth) break;
            decisions = 0;
            if (!isect.bsdf.is_delta()) {
                color_t Ld = beta * sample_all_light(isect, sampler, c);
                Lo += Ld;
            }
            if (c) c->bsdf_path_samples++;
            bsdf_sample_t bs = isect.bsdf.sample(isect.wo, sampler.get_float2());
            if (trace) {
                const float row[26] = {(float)bounces, (float)isect.surface, (float)isect.bsdf.lobe, isect.position.x, isect.position.y, isect.position.z,
                                       isect.normal.x, isect.normal.y, isect.normal.z, isect.wo.x, isect.wo.y, isect.wo.z, beta.r, beta.g, beta.b,
                                       Lo.r, Lo.g, Lo.b, bs.f.r, bs.f.g, bs.f.b, bs.pdf, abs_dot(bs.wi, isect.normal), (float)bs.bsdf_type,
                                       (float)(decisions & 0xffffu), (float)(decisions >> 16)};
                trace->insert(trace->end(), row, row + 26);
            }
            if (bs.f.is_black() || bs.pdf == 0.f) break;
            beta *= bs.f * abs_dot(bs.wi, isect.normal) / bs.pdf;
            is_prev_specular = bs.is_delta_bsdf();
            ray = isect.spawn_ray(bs.wi);
            if (bounces > 3) {
                float beta_max_component = beta.max_component_value();
                float q = std::max(0.05f, 1 - beta_max_component);
                if (c) c->rr_draws++;
                if (sampler.get_float() < q) break;
                else beta *= 1 / (1 - q);
            }
        }
        return Lo;
    }

    // debug_integrator_t::Li, 4105-4122
    color_t Li_debug(ray_t ray, sampler_t& sampler, counters_t* c) const {
        isect_t isect;
        if (scene->intersect(ray, &isect, c, &sampler)) {
            switch (kind) {
            case KY_INTEGRATOR_POSITION: { vec3_t v = isect.position.normalize(); return {v.x, v.y, v.z}; }
            case KY_INTEGRATOR_NORMAL: { vec3_t v = isect.normal.normalize(); return {v.x, v.y, v.z}; }
            case KY_INTEGRATOR_BASECOLOR: return isect.bsdf.eval(isect.wo, isect.normal);
            }
        }
        return color_t();
    }

    // direct_lighting_t::Li, 4136-4154
    color_t Li_direct(ray_t ray, sampler_t& sampler, counters_t* c) const {
        isect_t isect;
        bool hit = scene->intersect(ray, &isect, c, &sampler);
        if (!hit) return scene->environment_lighting();
        color_t Lo = isect.emission;
        if (!isect.bsdf.is_delta()) Lo += sample_all_light(isect, sampler, c);
        return Lo;
    }

    // simple_path_tracing_recursion_t::Li, 4201-4237 (BSDF sampling only; rays leave from the un-offset hit point)
    color_t Li_simple(ray_t ray, sampler_t& sampler, counters_t* c, int depth) const {
        isect_t isect;
        if (!scene->intersect(ray, &isect, c, &sampler)) return scene->environment_lighting();
        if (depth >= max_path_depth) return isect.emission;
        bsdf_sample_t bs = isect.bsdf.sample(isect.wo, sampler.get_float2());
        if (bs.f.is_black() || bs.pdf == 0.f) return isect.emission;
        if (++depth > 3) {                                                       // russian roulette, 4219-4226
            float bsdf_max_comp = bs.f.max_component_value();
            if (sampler.get_float() < bsdf_max_comp) bs.f *= (1 / bsdf_max_comp);
            else return isect.emission;
        }
        ray_t wi{isect.position, bs.wi, k_infinity};                             // 4232
        color_t Ls = bs.f * Li_simple(wi, sampler, c, depth) * abs_dot(bs.wi, isect.normal) / bs.pdf;
        return isect.emission + Ls;
    }

    // shared by the two NEE recursions: indirect_lighting, 4381-4401 / 4493-4513
    //   offset: path_tracing_recursion_t spawns from the offset origin (4399), the defered variant does not (4511)
    color_t indirect_lighting(const isect_t& isect, sampler_t& sampler, counters_t* c, int depth, bool defered) const {
        bsdf_sample_t bs = isect.bsdf.sample(isect.wo, sampler.get_float2());
        if (bs.f.is_black() || bs.pdf == 0.f) return color_t{};
        if (++depth > 3) {
            float bsdf_max_comp = bs.f.max_component_value();
            if (sampler.get_float() < bsdf_max_comp) bs.f *= 1 / bsdf_max_comp;
            else return color_t{};
        }
        if (defered) {
            ray_t wi_ray{isect.position, bs.wi, k_infinity};
            return bs.f * Li_defered(wi_ray, sampler, c, depth, isect.bsdf.is_delta()) * abs_dot(bs.wi, isect.normal) / bs.pdf;
        }
        ray_t wi_ray{offset_ray_origin(isect.position, isect.normal, bs.wi), bs.wi, k_infinity};
        return bs.f * Li_recursion(wi_ray, sampler, c, depth) * abs_dot(bs.wi, isect.normal) / bs.pdf;
    }

    // path_tracing_recursion_t::Li, 4321-4356
    color_t Li_recursion(ray_t ray, sampler_t& sampler, counters_t* c, int depth) const {
        color_t Lo;
        isect_t isect;
        bool hit = scene->intersect(ray, &isect, c, &sampler);
        if (depth == 0) Lo += hit ? isect.emission : scene->environment_lighting();      // emission_lighting, 4358-4372
        if (hit && depth < max_path_depth) {
            if (!isect.bsdf.is_delta()) {
                Lo += sample_all_light(isect, sampler, c);
            } else {  // specular vertex: look the emitter up along the sampled direction, 4341-4349
                bsdf_sample_t bs = isect.bsdf.sample(isect.wo, sampler.get_float2());
                ray_t wi_ray{isect.position, bs.wi, k_infinity};
                isect_t next_isect;
                bool next_hit = scene->intersect(wi_ray, &next_isect, c, &sampler);
                color_t Le = next_hit ? next_isect.emission : scene->environment_lighting();
                Lo += bs.f * Le * abs_dot(bs.wi, isect.normal) / bs.pdf;
            }
            Lo += indirect_lighting(isect, sampler, c, depth, false);
        }
        return Lo;
    }

    // path_tracing_recursion_defered_t::Li, 4440-4466
    color_t Li_defered(ray_t ray, sampler_t& sampler, counters_t* c, int depth, bool is_prev_specular) const {
        color_t Lo;
        isect_t isect;
        bool hit = scene->intersect(ray, &isect, c, &sampler);
        if (depth == 0 || is_prev_specular) Lo += hit ? isect.emission : scene->environment_lighting();
        if (hit && depth < max_path_depth) {
            if (!isect.bsdf.is_delta()) Lo += sample_all_light(isect, sampler, c);
            Lo += indirect_lighting(isect, sampler, c, depth, true);
        }
        return Lo;
    }

    color_t Li(ray_t ray, sampler_t& sampler, counters_t* c) const {
        switch (kind) {
        case KY_INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION: return Li_simple(ray, sampler, c, 0);
        case KY_INTEGRATOR_PATH_TRACING_RECURSION: return Li_recursion(ray, sampler, c, 0);
        case KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED: return Li_defered(ray, sampler, c, 0, false);
        case KY_INTEGRATOR_PATH_TRACING_ITERATION: return Li_path(ray, sampler, c);
        case KY_INTEGRATOR_DIRECT_LIGHTING: return Li_direct(ray, sampler, c);
        default: return Li_debug(ray, sampler, c);
        }
    }
};

inline float clamp01(float x) { return std::clamp(x, 0.f, 1.f); }                         // 1545

inline bool valid_direct_sample(int v) {
    return v == KY_DIRECT_IDLE || v == KY_DIRECT_BSDF || v == KY_DIRECT_LIGHT || v == KY_DIRECT_BSDF_MIS || v == KY_DIRECT_LIGHT_MIS ||
           v == KY_DIRECT_BOTH_MIS;
}
inline bool valid_integrator(int v) {
    return v == KY_INTEGRATOR_POSITION || v == KY_INTEGRATOR_NORMAL || v == KY_INTEGRATOR_BASECOLOR || v == KY_INTEGRATOR_DIRECT_LIGHTING ||
           (v >= KY_INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION && v <= KY_INTEGRATOR_PATH_TRACING_ITERATION);
}

}  // namespace kyo

// ============================================================================================
// C entry points (ctypes)
// ============================================================================================
using namespace kyo;

extern "C" {

// integrator_t::render, ky.cpp:3689-3729, restricted to the tile shard named by params.
// Adds clamp01(L) into film (film_t::add_color, 1586).  threads <= 0: all OpenMP threads.
// counters (optional): 11 uint64 in counters_t order.
int kyo_render(const ky_scene* cscene, const ky_render_params* p, float* film, size_t stride_px, int threads, uint64_t* counters_out) {
    if (!cscene || !p || !film) return KY_ERR_INVALID_VALUE;
    if (!valid_integrator(p->integrator) || !valid_direct_sample(p->direct_sample)) return KY_ERR_INVALID_VALUE;
    if (p->samples_per_pixel <= 0 || p->width <= 0 || p->height <= 0 || p->tile_w <= 0 || p->tile_h <= 0 || p->tile_step <= 0) return KY_ERR_INVALID_VALUE;
    scene_t scene(*cscene);
    integrator_t integrator{&scene, p->integrator, p->max_path_depth, p->direct_sample};
    const int width = p->width, height = p->height, spp = p->samples_per_pixel;
    const int tiles_x = (width + p->tile_w - 1) / p->tile_w;
    counters_t total;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        counters_t local;
        counters_t* c = counters_out ? &local : nullptr;
#pragma omp for schedule(dynamic, 1)
        for (int y = 0; y < height; y += 1) {
            for (int x = 0; x < width; x += 1) {
                const int trow = y / p->tile_h, tcol = x / p->tile_w;   // tile numbering of include/kyhip.h: rows rotated by their index
                int tile = trow * tiles_x + ((tcol - trow) % tiles_x + tiles_x) % tiles_x;
                if (tile < p->tile_first || (tile - p->tile_first) % p->tile_step != 0) continue;
                color_t L{};
                sampler_t sampler;
                sampler.kind = p->sampler;
                for (int s = 0; s < spp; ++s) {
                    sampler.start_sample(p->seed, (uint32_t)(y * width + x), (uint32_t)s);
                    vec2_t cs  = sampler.get_camera_sample({(float)x, (float)y});          // 3714
                    ray_t ray  = generate_ray(scene.camera, cs);                           // 3715
                    if (c) c->camera_samples++;
                    color_t dL = integrator.Li(ray, sampler, c) * (float)(1. / spp);       // 3717
                    L = L + dL;                                                            // 3721
                }
                float* px = film + ((size_t)y * stride_px + x) * 3;                        // 1574
                px[0] += clamp01(L.r); px[1] += clamp01(L.g); px[2] += clamp01(L.b);       // 3726, 1586-1590
            }
        }
#pragma omp critical
        total.add(local);
    }
    if (counters_out) std::memcpy(counters_out, &total, sizeof(total));
    return KY_OK;
}

// per-sample unclamped Li for pixel (x, y), samples [s0, s0+n)
int kyo_li(const ky_scene* cscene, const ky_render_params* p, int x, int y, int s0, int n, float* out3) {
    if (!valid_integrator(p->integrator) || !valid_direct_sample(p->direct_sample)) return KY_ERR_INVALID_VALUE;
    scene_t scene(*cscene);
    integrator_t integrator{&scene, p->integrator, p->max_path_depth, p->direct_sample};
    sampler_t sampler;
    sampler.kind = p->sampler;
    for (int i = 0; i < n; ++i) {
        sampler.start_sample(p->seed, (uint32_t)(y * p->width + x), (uint32_t)(s0 + i));
        vec2_t cs = sampler.get_camera_sample({(float)x, (float)y});
        ray_t ray = generate_ray(scene.camera, cs);
        color_t L = integrator.Li(ray, sampler, nullptr);
        out3[3 * i] = L.r; out3[3 * i + 1] = L.g; out3[3 * i + 2] = L.b;
    }
    return KY_OK;
}

// per-vertex trace of one camera sample (path_tracing_iteration_t): one row of 26 floats per vertex that reaches the
// continuation sample (4586): {bounce, surface, lobe, p[3], n[3], wo[3], beta[3] before the bounce, Lo[3] after this vertex's
// direct lighting, bs.f[3], bs.pdf, |cos|, bsdf flags, which per-light BSDF-half estimates were non-black (bits), which
// light-half estimates}; returns the row count.  The twin of kyhip_kat_li_trace.
int kyo_trace_li(const ky_scene* cscene, const ky_render_params* p, int x, int y, int s, float* rows, int max_rows) {
    scene_t scene(*cscene);
    integrator_t integrator{&scene, p->integrator, p->max_path_depth, p->direct_sample};
    std::vector<float> tr;
    integrator.trace = &tr;
    sampler_t sampler;
    sampler.kind = p->sampler;
    sampler.start_sample(p->seed, (uint32_t)(y * p->width + x), (uint32_t)s);
    vec2_t cs = sampler.get_camera_sample({(float)x, (float)y});
    integrator.Li(generate_ray(scene.camera, cs), sampler, nullptr);
    int n = std::min<int>((int)tr.size() / 26, max_rows);
    std::memcpy(rows, tr.data(), (size_t)n * 26 * sizeof(float));
    return n;
}

// One light's direct-lighting estimate at a given vertex with given random numbers (the halves of estimate_direct_lighting_*,
// 3889-4088).  in15: n x {p[3], normal[3], wo[3], surface (caller's index), lobe_u, random_bsdf[2], random_light[2]};
// out6: n x {BSDF-sampling half [3], light-sampling half [3]} -- by_bsdf / by_emitter for the plain strategies 4 / 8 (by_bsdf takes
// random_bsdf as the float2 it draws itself, 3900), the _mis variants for 16 / 32 / 48 (both halves, unweighted by the 0.5 of 4083).
int kyo_kat_nee(const ky_scene* cscene, int direct_sample, int light, const float* in15, int n, float* out6) {
    if (!valid_direct_sample(direct_sample) || light < 0 || light >= cscene->light_count) return KY_ERR_INVALID_VALUE;
    scene_t scene(*cscene);
    integrator_t integrator{&scene, KY_INTEGRATOR_PATH_TRACING_ITERATION, 5, direct_sample};
    for (int i = 0; i < n; ++i) {
        const float* r = in15 + 15 * i;
        isect_t isect;
        isect.position = vec3_t(r); isect.normal = vec3_t(r + 3); isect.wo = vec3_t(r + 6); isect.surface = (int)r[9];
        scene.scattering(&isect, r[10]);
        const vec2_t ub{r[11], r[12]}, ul{r[13], r[14]};
        color_t Lb, Ll;
        sampler_t sampler;
        sampler.tape = r + 11;
        switch (direct_sample) {
        case KY_DIRECT_BSDF: Lb = integrator.by_bsdf(isect, light, sampler, nullptr); break;
        case KY_DIRECT_LIGHT: Ll = integrator.by_emitter(isect, light, ul, nullptr); break;
        case KY_DIRECT_BSDF_MIS: Lb = integrator.by_bsdf_mis(isect, light, ub, nullptr); break;
        case KY_DIRECT_LIGHT_MIS: Ll = integrator.by_emitter_mis(isect, light, ul, nullptr); break;
        case KY_DIRECT_BOTH_MIS: Lb = integrator.by_bsdf_mis(isect, light, ub, nullptr); Ll = integrator.by_emitter_mis(isect, light, ul, nullptr); break;
        default: break;
        }
        float* o = out6 + 6 * i;
        o[0] = Lb.r; o[1] = Lb.g; o[2] = Lb.b; o[3] = Ll.r; o[4] = Ll.g; o[5] = Ll.b;
    }
    return KY_OK;
}

int kyo_kat_intersect(const ky_shape* shape, const float* rays7, int n, float* out8) {
    shape_t s(*shape);
    for (int i = 0; i < n; ++i) {
        const float* r = rays7 + 7 * i;
        ray_t ray{vec3_t(r), vec3_t(r + 3), r[6]};
        isect_t isect;
        bool hit = s.intersect(ray, &isect);
        float* o = out8 + 8 * i;
        o[0] = hit ? 1.f : 0.f; o[1] = hit ? ray.distance : 0.f;
        o[2] = isect.position.x; o[3] = isect.position.y; o[4] = isect.position.z;
        o[5] = isect.normal.x; o[6] = isect.normal.y; o[7] = isect.normal.z;
    }
    return KY_OK;
}

// frame_t (526-578): in n x {normal[3], v[3]}; out n x {s[3], t[3], n[3], to_local(v)[3], to_world(v)[3]}
int kyo_kat_frame(const float* in6, int n, float* out15) {
    for (int i = 0; i < n; ++i) {
        const float* r = in6 + 6 * i;
        const frame_t f{vec3_t(r)};
        const vec3_t v(r + 3), l = f.to_local(v), w = f.to_world(v);
        float* o = out15 + 15 * i;
        o[0] = f.s_.x; o[1] = f.s_.y; o[2] = f.s_.z; o[3] = f.t_.x; o[4] = f.t_.y; o[5] = f.t_.z; o[6] = f.n_.x; o[7] = f.n_.y; o[8] = f.n_.z;
        o[9] = l.x; o[10] = l.y; o[11] = l.z; o[12] = w.x; o[13] = w.y; o[14] = w.z;
    }
    return KY_OK;
}

int kyo_kat_camera(const ky_camera* camera, const float* p_film2, int n, float* out6) {
    for (int i = 0; i < n; ++i) {
        ray_t ray = generate_ray(*camera, vec2_t{p_film2[2 * i], p_film2[2 * i + 1]});
        float* o = out6 + 6 * i;
        o[0] = ray.origin.x; o[1] = ray.origin.y; o[2] = ray.origin.z;
        o[3] = ray.direction.x; o[4] = ray.direction.y; o[5] = ray.direction.z;
    }
    return KY_OK;
}

// in: n x {normal[3], wo[3], u[2], wi_eval[3], lobe_u}; out: n x {f[3], wi[3], pdf, flags, eval[3], pdf_eval, is_delta}
int kyo_kat_bsdf(const ky_material* material, const float* in12, int n, float* out13) {
    ky_scene cs{};
    ky_surface surf{0, 0, -1};
    cs.materials = material; cs.material_count = 1; cs.surfaces = &surf; cs.surface_count = 1; cs.environment_light = -1;
    scene_t scene(cs);
    for (int i = 0; i < n; ++i) {
        const float* r = in12 + 12 * i;
        isect_t isect;
        isect.normal = vec3_t(r); isect.wo = vec3_t(r + 3); isect.surface = 0;
        scene.scattering(&isect, r[11]);
        bsdf_sample_t bs = isect.bsdf.sample(isect.wo, vec2_t{r[6], r[7]});
        color_t ev = isect.bsdf.eval(isect.wo, vec3_t(r + 8));
        float pd   = isect.bsdf.pdf(isect.wo, vec3_t(r + 8));
        float* o = out13 + 13 * i;
        o[0] = bs.f.r; o[1] = bs.f.g; o[2] = bs.f.b; o[3] = bs.wi.x; o[4] = bs.wi.y; o[5] = bs.wi.z; o[6] = bs.pdf;
        o[7] = (float)bs.bsdf_type; o[8] = ev.r; o[9] = ev.g; o[10] = ev.b; o[11] = pd; o[12] = isect.bsdf.is_delta() ? 1.f : 0.f;
    }
    return KY_OK;
}

// in: n x {p[3], normal[3], u[2], wi[3]}; out: n x {position[3], wi[3], pdf, Li[3], pdf_Li}
int kyo_kat_light(const ky_scene* cscene, int light, const float* in11, int n, float* out11) {
    scene_t scene(*cscene);
    for (int i = 0; i < n; ++i) {
        const float* r = in11 + 11 * i;
        isect_t isect;
        isect.position = vec3_t(r); isect.normal = vec3_t(r + 3);
        light_sample_t ls = scene.sample_Li(light, isect, vec2_t{r[6], r[7]});
        float pdf = scene.pdf_Li(light, isect, vec3_t(r + 8));
        float* o = out11 + 11 * i;
        o[0] = ls.position.x; o[1] = ls.position.y; o[2] = ls.position.z; o[3] = ls.wi.x; o[4] = ls.wi.y; o[5] = ls.wi.z;
        o[6] = ls.pdf; o[7] = ls.Li.r; o[8] = ls.Li.g; o[9] = ls.Li.b; o[10] = pdf;
    }
    return KY_OK;
}

// out: n x {hit, t, p[3], n[3], surface}
int kyo_kat_scene_intersect(const ky_scene* cscene, const float* rays7, int n, float* out9) {
    scene_t scene(*cscene);
    for (int i = 0; i < n; ++i) {
        const float* r = rays7 + 7 * i;
        ray_t ray{vec3_t(r), vec3_t(r + 3), r[6]};
        isect_t isect;
        bool hit = scene.intersect(ray, &isect, nullptr);
        float* o = out9 + 9 * i;
        o[0] = hit ? 1.f : 0.f; o[1] = hit ? ray.distance : 0.f;
        o[2] = isect.position.x; o[3] = isect.position.y; o[4] = isect.position.z;
        o[5] = isect.normal.x; o[6] = isect.normal.y; o[7] = isect.normal.z; o[8] = (float)isect.surface;
    }
    return KY_OK;
}

// in: n x {p[3], normal[3], target[3]}; out: n x {0/1}
int kyo_kat_occluded(const ky_scene* cscene, const float* in9, int n, float* out1) {
    scene_t scene(*cscene);
    for (int i = 0; i < n; ++i) {
        const float* r = in9 + 9 * i;
        isect_t isect;
        isect.position = vec3_t(r); isect.normal = vec3_t(r + 3);
        out1[i] = scene.occluded(isect, vec3_t(r + 6), nullptr) ? 1.f : 0.f;
    }
    return KY_OK;
}

// scene bounding sphere as light_t::preprocess computes it (3555-3574): out = {cx, cy, cz, radius}
int kyo_world_bounding_sphere(const ky_scene* cscene, float* out4) {
    scene_t scene(*cscene);
    bounds3_t b;
    for (const ky_surface& s : scene.surfaces) b = b.join(scene.shapes[s.shape].world_bound());   // 3209-3219
    vec3_t c; float r;
    b.bounding_sphere(&c, &r);
    out4[0] = c.x; out4[1] = c.y; out4[2] = c.z; out4[3] = r;
    return KY_OK;
}

int kyo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
