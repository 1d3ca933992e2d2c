/*
 * ky_device.hpp -- gfx950 device code of the path-tracing hot path (one path vertex per call).
 *
 * Written for CDNA4: 64-lane wavefronts, one lane = one path.  The primitive list is read with
 * wave-uniform indices (scalar loads into SGPRs: no VGPR or LDS traffic for the traversal); the
 * tables that are looked up with a per-lane index AFTER the nearest hit is known (surface ->
 * normal / material / light, materials, light radiance) are staged in LDS once per workgroup.
 * No MFMA: there is no dense contraction on this path.
 *
 * Arithmetic is fp32 like the reference, but organised for the VALU rather than transcribed:
 *   - rectangles lying in an axis plane are tested with the ray's reciprocal direction (12 VALU), other planar
 *     parallelograms with a plane hit plus two precomputed dual-basis dot products (26 VALU) instead of the
 *     reference's four edge cross products; quads that are not parallelograms, triangles and disks keep the
 *     reference's formulation;
 *   - 1/x, 1/sqrt(x), sqrt(x), sin/cos(2 pi x), exp2 and log2 use the hardware instructions
 *     (v_rcp/v_rsq/v_sqrt/v_sin/v_cos/v_exp/v_log, about 1 ulp);
 *   - vectors that are unit by construction are not normalised again.
 * These change results at the 1e-6 relative level; parity with the CPU oracle is by tolerance
 * (tests/test_parity_gpu.py), never bit-exact.
 *
 * Every function names the reference function it implements (file = /root/reference/ky.cpp).
 * This file is independent of oracle/ (the CPU checker): nothing is shared between the two.
 */
#pragma once
#ifndef __HIPCC_RTC__   // hiprtc brings the device runtime and the fixed-width integer types itself (ky_render.hpp is also compiled at run time)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "ky_scene.hpp"   // DScene and its records, KY_FEAT_*, RenderConst: plain data shared with the host's packing code

#define KY_DEV __device__ __forceinline__

// KY_PROBE(k) / KY_CLK(k): lane-utilisation probes and phase clocks of measurement builds (ky_measure.hpp); nothing in product builds
#if defined(KY_PROFILE_LANES) || defined(KY_PROFILE_CLOCKS) || defined(KY_MARKS)
#include "ky_measure.hpp"
#else
#define KY_PROBE(k) do { } while (0)
#define KY_CLK(k) do { } while (0)
#endif

namespace kyd {

// ---------------------------------------------------------------------------------------------
// scalar helpers: single hardware instructions
// ---------------------------------------------------------------------------------------------
KY_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
KY_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
KY_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
KY_DEV float sin_rev(float x) { return __builtin_amdgcn_sinf(x); }  // sin(2 pi x)
KY_DEV float cos_rev(float x) { return __builtin_amdgcn_cosf(x); }  // cos(2 pi x)
KY_DEV float clamp01f(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 1.f); }
// A value that is never read: what a variable holds on the paths that do not assign it.  (An explicit default would be a v_mov per
// register at every level of nested divergent control flow; this is "any register content", which costs nothing.)
// (A register definition without an instruction.  __builtin_nondeterministic_value says the same to the optimiser, but where such a value merges
// with a real one at a join the compiler materialises it as v_mov 0: round 4 measured +0.35 % / +2.3 % for the empty asm on the two scenes.)
KY_DEV float any_f() { float x; asm volatile("" : "=v"(x)); return x; }
KY_DEV float any_reg() { return any_f(); }
KY_DEV unsigned any_reg_u() { unsigned x; asm volatile("" : "=v"(x)); return x; }
// (The compiler puts an `s_nop 0` next to every such statement -- its hazard recogniser cannot see inside an inline asm -- 44 per loop turn of the hot kernel.  They are
// free: round 5 measured one statement with thirteen outputs against thirteen statements at 42.92 against 42.95-42.99 ms on configs[1]; s_nop does not take a scalar-ALU slot.)
// v_mul_legacy_f32: the product with 0 x anything = 0 (inf and NaN included), otherwise the IEEE product bit for bit.  It turns two special cases into no case at all:
// pow(x, 0) = 1 for every x (0 x log2(x) = 0 also for log2(0) = -inf and for NaN, as std::pow has it), and the concentric map's 0 / 0 at the square's centre
// (0 x (1 / 0) = 0: the centre maps to the centre) -- a compare and a select, or two compares, a scalar and and two selects, per call otherwise.
// (the s_nop: an operand may come straight from a quarter-rate instruction -- v_rcp_f32, v_log_f32 -- whose result the next VALU instruction must not read without a wait
// state; the compiler's hazard recogniser does not look inside an asm statement, and without the wait the product was made from a stale register: measured, 0.3 % of a film's mean)
KY_DEV float mul_legacy(float a, float b) { float r; asm("s_nop 0\n\tv_mul_legacy_f32_e64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// x^n for x >= 0 (pow(0, n > 0) = 0, pow(x, 0) = 1)
KY_DEV float pow_nonneg(float x, float n) {
    return __builtin_amdgcn_exp2f(mul_legacy(n, __builtin_amdgcn_logf(x)));
}

// ---------------------------------------------------------------------------------------------
// vectors (ky.cpp:226-388)
// ---------------------------------------------------------------------------------------------
struct f3 {
    float x, y, z;
};
KY_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
KY_DEV f3 any3() { return f3{any_f(), any_f(), any_f()}; }
KY_DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
KY_DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
KY_DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
KY_DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
KY_DEV f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
KY_DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
KY_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
KY_DEV f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
KY_DEV float length_sq(f3 a) { return dot(a, a); }
KY_DEV f3 normalize(f3 a) { return a * rsq(dot(a, a)); }  // vec3_t::normalize, 314
KY_DEV float max3(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
KY_DEV bool is_black(f3 c) { return (c.x <= 0) && (c.y <= 0) && (c.z <= 0); }  // color_t::is_black, 258
// the same for a WAVE-UNIFORM colour in memory (a light's), decided on the scalar unit: x <= 0 for a float that is not a NaN is "sign bit set or zero", i.e. its
// bits as a signed integer are <= 0 (s_cmp_le_i32; there is no scalar float compare on this chip, and as floats the three tests are VALU compares of SGPRs)
KY_DEV bool is_black_bits(const float* c) {
    const int* b = (const int*)c;
    return (b[0] <= 0) & (b[1] <= 0) & (b[2] <= 0);
}

// One channel of a film sum -> what goes into the 32.32 fixed-point accumulator and into the pixel's flag word, without a branch:
// NaN, +inf and -inf (or beyond the accumulator's range) set flag bits 1 << ch, 8 << ch, 64 << ch and add nothing.
// rint(a * 2^32) as a two's-complement 64-bit number for |a| <= 2e9, in single precision: the integer part and the fraction are converted apart.
// floor(|a|) < 2^31 converts exactly; |a| - floor(|a|) is exact (a multiple of |a|'s ulp with fewer significant bits than |a|); times 2^32 it is exact
// too, an integer already when it is 2^24 or more and rounded to the nearest even integer by v_rndne_f32 when it is less, and at most 2^32 - 2^8.
// Bit for bit what `__double2ll_rn((double)a * 4294967296.0)` gives (checked on the host over 2.5e8 random bit patterns and the edge cases), in six
// full-rate instructions instead of eight double-precision ones.
KY_DEV unsigned long long fixed_from_float(float a) {
    const float aa = fabsf(a);
    const float hi = __builtin_floorf(aa);
    const float lo = __builtin_rintf((aa - hi) * 4294967296.0f);
    const unsigned long long v = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
    return a < 0 ? 0ull - v : v;
}
KY_DEV unsigned long long film_fixed(float a, int ch, unsigned& flags) {
    const bool nan = a != a, pos = a > 2.0e9f, neg = a < -2.0e9f;
    flags |= (nan ? 1u << ch : 0u) | (pos ? 8u << ch : 0u) | (neg ? 64u << ch : 0u);
    const float b = (nan | pos | neg) ? 0.f : a;
    return fixed_from_float(b);
}

// float -> 32.32 fixed point (|a| <= 2e9): exact for |a| >= 2^-8, truncated below
KY_DEV long long to_fixed32(float a) {
    const float aa = fabsf(a);
    const float hi = floorf(aa);
    const float fr = aa - hi;
    const unsigned long long v = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)(fr * 4294967296.0f);
    return a < 0 ? -(long long)v : (long long)v;
}

constexpr float K_PI = 3.14159265358979323846f;
constexpr float K_INV_PI = 0.318309886183790671538f;
constexpr float K_INV_2PI = K_INV_PI / 2.f;
constexpr float K_SHAPE_EPS = 1e-3f;   // shape_t::epsilon, 1093
constexpr float K_RAY_OFFSET = 1e-2f;  // offset_ray_origin, 616
constexpr float K_INF = __builtin_huge_valf();

// How device functions see the scene: the pointer plus the compile-time facts of the instantiation (ky_scene.hpp, KY_FEAT_*; `general`: the scene may
// hold shapes that need the reference's own formulations, `large`: more than KY_LDS_SURFACES surfaces).
struct SceneRef {
    const DScene* p;
    bool general;
    int feat;
    bool large;   // the scene may hold more than 64 surfaces (LARGE instantiations; everything that converts from the bare pointer)
    __device__ __forceinline__ SceneRef(const DScene* p_) : p(p_), general(true), feat(0), large(true) {}
    __device__ __forceinline__ SceneRef(const DScene* p_, bool general_, int feat_ = 0, bool large_ = false) : p(p_), general(general_), feat(feat_), large(large_) {}
    __device__ __forceinline__ const DScene* operator->() const { return p; }
    __device__ __forceinline__ bool single_area() const { return (feat & KY_FEAT_SINGLE_AREA) != 0; }
    __device__ __forceinline__ bool single_light() const { return (feat & KY_FEAT_SINGLE_LIGHT) != 0; }
    // light kinds as far as the instantiation's facts decide them (wave-uniform; the rest is read from the light)
    __device__ __forceinline__ bool is_area(int kind) const {
        return (feat & (KY_FEAT_SINGLE_AREA | KY_FEAT_SPHERE_LIGHTS)) ? true : ((feat & (KY_FEAT_SINGLE_DELTA | KY_FEAT_SINGLE_ENV)) ? false : kind == KY_LIGHT_AREA);
    }
    __device__ __forceinline__ bool is_delta(int kind) const {
        return (feat & KY_FEAT_SINGLE_DELTA) ? true : ((feat & (KY_FEAT_SINGLE_AREA | KY_FEAT_SINGLE_ENV | KY_FEAT_SPHERE_LIGHTS)) ? false : (kind == KY_LIGHT_POINT || kind == KY_LIGHT_DIRECTION));
    }
    __device__ __forceinline__ bool is_env(int kind) const {
        return (feat & KY_FEAT_SINGLE_ENV) ? true : ((feat & (KY_FEAT_SINGLE_AREA | KY_FEAT_SINGLE_DELTA | KY_FEAT_SPHERE_LIGHTS)) ? false : kind == KY_LIGHT_ENVIRONMENT);
    }
    __device__ __forceinline__ bool may_have_env() const { return (feat & (KY_FEAT_SINGLE_AREA | KY_FEAT_SINGLE_DELTA | KY_FEAT_SPHERE_LIGHTS)) == 0; }
    __device__ __forceinline__ bool no_carried_light() const { return (feat & (KY_FEAT_SINGLE_DELTA | KY_FEAT_SINGLE_ENV)) != 0; }
    __device__ __forceinline__ bool x_planks() const { return (feat & KY_FEAT_X_PLANKS) != 0; }
    __device__ __forceinline__ bool sphere_lights() const { return (feat & KY_FEAT_SPHERE_LIGHTS) != 0; }
    __device__ __forceinline__ bool boxes() const { return (feat & KY_FEAT_BOXES) != 0; }
    __device__ __forceinline__ bool flat_phong() const { return (feat & KY_FEAT_FLAT_PHONG) != 0; }   // every plastic surface is a rectangle (bsdf_sample_dir_nondelta)
    __device__ __forceinline__ bool no_par() const { return (feat & KY_FEAT_AXIS_ALIGNED) != 0; }   // every planar surface is a rectangle in an axis plane: no parallelogram loops   // nearest-hit traversals scan DScene::boxtrav
    // the light-sampling estimators work with the RECIPROCAL of the light's density (shape_sample_direction): where every light is a sphere lamp
    __device__ __forceinline__ bool ipdf() const { return (feat & KY_FEAT_SPHERE_LIGHTS) != 0; }   // (measured on the one-rectangle-lamp kernel too: configs[1] -0.3 %, not taken)
};

struct LdsScene {
    const DHit* hit;
    const DMat* mat;
    const float (*light_color)[4];
};
template <int NS, int NM>
struct LdsSceneStaticT {
    DHit hit[NS];
    DMat mat[NM];
    float light_color[KYHIP_MAX_LIGHTS][4];
};
using LdsSceneStatic = LdsSceneStaticT<KY_LDS_SURFACES, KY_LDS_MATERIALS>;
extern __shared__ __attribute__((aligned(16))) unsigned char g_lds_scene[];

KY_DEV f3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

// cooperative copy global -> LDS, whole workgroup; ends with a barrier
template <bool LARGE, bool SMALL = false>
KY_DEV LdsScene stage_scene(SceneRef S) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ns = S->n_surfaces, nm = S->n_materials, nl = S->n_lights;
    uint32_t *dst_h, *dst_m;
    float* dst_l;
    if (LARGE) {
        dst_h = reinterpret_cast<uint32_t*>(g_lds_scene);
        dst_m = reinterpret_cast<uint32_t*>(g_lds_scene + lds_scene_mat_offset(ns));
        dst_l = reinterpret_cast<float*>(g_lds_scene + lds_scene_light_offset(ns, nm));
    } else {
        __shared__ LdsSceneStaticT<SMALL ? KY_LDS_SURFACES_SMALL : KY_LDS_SURFACES, SMALL ? KY_LDS_MATERIALS_SMALL : KY_LDS_MATERIALS> L;
        dst_h = reinterpret_cast<uint32_t*>(L.hit);
        dst_m = reinterpret_cast<uint32_t*>(L.mat);
        dst_l = &L.light_color[0][0];
    }
    const uint32_t* src_h = reinterpret_cast<const uint32_t*>(S->hit);
    for (int i = tid; i < ns * (int)(sizeof(DHit) / 4); i += nt) dst_h[i] = src_h[i];
    const uint32_t* src_m = reinterpret_cast<const uint32_t*>(S->mat);
    for (int i = tid; i < nm * (int)(sizeof(DMat) / 4); i += nt) dst_m[i] = src_m[i];
    for (int i = tid; i < nl; i += nt) {
        dst_l[4 * i + 0] = S->light[i].color[0];
        dst_l[4 * i + 1] = S->light[i].color[1];
        dst_l[4 * i + 2] = S->light[i].color[2];
        dst_l[4 * i + 3] = 0.f;
    }
    __syncthreads();
    return LdsScene{reinterpret_cast<const DHit*>(dst_h), reinterpret_cast<const DMat*>(dst_m), reinterpret_cast<const float (*)[4]>(dst_l)};
}

// ---------------------------------------------------------------------------------------------
// random numbers: one xoroshiro64+ stream per camera sample, keyed (seed, pixel, sample) -- DESIGN.md section 5
// sampler_t semantics of ky.cpp:877-975
// ---------------------------------------------------------------------------------------------
KY_DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21f0aaadu;
    x ^= x >> 15; x *= 0x735a2d97u;
    x ^= x >> 15;
    return x;
}
struct Sampler {
    uint32_t s0, s1;
};
// the state of a sample's xoroshiro64+ stream from (seed, pixel, sample) (s1 is made odd: the all-zero state is excluded)
KY_DEV uint32_t sampler_pixel_key(uint32_t seed, uint32_t pixel_index) { return mix32(pixel_index ^ mix32(seed)); }  // constant per pixel
// One hash per sample (round 5; two before): the second state word is the first one rotated by half a word, xor the pixel's key, made odd -- three full-rate
// instructions where the second lowbias32 was nine, two of them slow integer multiplies (configs[1] +0.9 %).  Samples of one pixel differ in s0 by a hash, pixels in the
// key; two samples share a stream only if both s0 and the pixel key collide, as before.  The oracle defines the stream the same way (oracle/ky_oracle.cpp, start_sample).
KY_DEV void sampler_start(Sampler& s, uint32_t h, uint32_t sample_index) {
    s.s0 = mix32(h + sample_index * 0x9E3779B9u);
    s.s1 = (__builtin_amdgcn_alignbit(s.s0, s.s0, 16) ^ h) | 1u;
}
KY_DEV uint32_t rotl32(uint32_t x, int k) { return __builtin_amdgcn_alignbit(x, x, 32 - k); }   // v_alignbit_b32
// xoroshiro64+ (Blackman / Vigna; a = 26, b = 9, c = 13): eight full-rate VALU instructions per number with the conversion -- a xor, a three-way xor
// (v_bitop3_b32), two v_alignbit, a shift, an add; v_alignbit, subtract -- against eleven for rounds 1-3's PCG-RXS-M-XS-32 with its two slow integer multiplies
// (v_mul_lo_u32, v_mad_u64_u32).  The sum's weak low bits are among the nine the conversion drops.
template <bool DEBUG_SAMPLER>
KY_DEV float sampler_next(Sampler& s) {
    if (DEBUG_SAMPLER) return 0.5f;  // debug_sampler_t, 933-941
    const uint32_t r = s.s0 + s.s1;
    s.s1 ^= s.s0;
    s.s0 = __builtin_amdgcn_bitop3_b32(rotl32(s.s0, 26), s.s1, s.s1 << 9, 0x96);   // a ^ b ^ c in one v_bitop3_b32 (gfx950; the compiler writes two v_xor for the C expression)
    s.s1 = rotl32(s.s1, 13);
    // the sum's upper 23 bits as the mantissa of a float in [1, 2), minus one: v_alignbit_b32 (0x7f : r) >> 9, v_sub -- two instructions where
    // shift, convert, scale were three; the value is (r >> 9) 2^-23 exactly, which is how the oracle writes it
    return __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, r, 9)) - 1.0f;
}

// ---------------------------------------------------------------------------------------------
// camera_t::generate_ray, ky.cpp:1884-1892
// ---------------------------------------------------------------------------------------------
KY_DEV void generate_ray(SceneRef S, float px, float py, f3& o, f3& d) {
    const float sx = px * S->cam_inv_w - 0.5f;
    const float sy = 0.5f - py * S->cam_inv_h;
    const f3 dir = ld3(S->cam_front) + ld3(S->cam_right) * sx + ld3(S->cam_up) * sy;
    o = ld3(S->cam_position);
    d = normalize(dir);
}

// offset_ray_origin, ky.cpp:614-620
KY_DEV f3 offset_ray_origin(f3 position, f3 normal, f3 direction) {
    // the offset with the cosine's sign: one v_bfi_b32 where the comparison, two constants and a select were four instructions.  (A cosine of -0.0 would
    // take the other side than the reference's `< 0`: it takes three negative-zero products to make one, which no ray leaving a surface has.)
    const float s = __builtin_copysignf(K_RAY_OFFSET, dot(normal, direction));
    return position + normal * s;
}

// ---------------------------------------------------------------------------------------------
// shape_t::intersect -- distance only.  Returns true and sets t when eps < t < tmax.
// ---------------------------------------------------------------------------------------------
KY_DEV bool is_equal_zero(float x) {  // is_equal(x, 0.f), ky.cpp:212-220
    const float eps = 1.1920929e-07f;
    return fabsf(x) <= eps * fmaxf(1.f, fabsf(x));
}

// the reference's formulations, used for general quads, triangles and disks
KY_DEV bool full_shape_hit(const DShapeFull& S, f3 o, f3 d, float tmax, float& t_out) {
    if (S.kind == KY_SHAPE_RECTANGLE) {  // rectangle_t::intersect, 1261-1297
        const f3 oa = ld3(S.p[0]) - o, ob = ld3(S.p[1]) - o, oc = ld3(S.p[2]) - o, od = ld3(S.p[3]) - o;
        const float v0d = dot(cross(oc, ob), d), v1d = dot(cross(ob, oa), d), v2d = dot(cross(oa, od), d), v3d = dot(cross(od, oc), d);
        const bool neg = (v0d < 0.f) && (v1d < 0.f) && (v2d < 0.f) && (v3d < 0.f);
        const bool pos = (v0d >= 0.f) && (v1d >= 0.f) && (v2d >= 0.f) && (v3d >= 0.f);
        const f3 n = ld3(S.n);
        const float t = dot(n, oa) * rcp(dot(n, d));
        t_out = t;
        return (neg || pos) && (t > K_SHAPE_EPS) && (t < tmax);
    } else if (S.kind == KY_SHAPE_TRIANGLE) {  // triangle_t::intersect, 1179-1215
        const f3 oa = ld3(S.p[0]) - o, ob = ld3(S.p[1]) - o, oc = ld3(S.p[2]) - o;
        const float v0d = dot(cross(oc, ob), d), v1d = dot(cross(ob, oa), d), v2d = dot(cross(oa, oc), d);
        const bool neg = (v0d < 0.f) && (v1d < 0.f) && (v2d < 0.f);
        const bool pos = (v0d >= 0.f) && (v1d >= 0.f) && (v2d >= 0.f);
        const f3 n = ld3(S.n);
        const float t = dot(n, oa) * rcp(dot(n, d));
        t_out = t;
        return (neg || pos) && (t > K_SHAPE_EPS) && (t < tmax);
    } else {  // disk_t::intersect, 1111-1132
        const f3 n = ld3(S.n), c = ld3(S.p[0]);
        const float nd = dot(d, n);
        const float t = dot(n, c - o) * rcp(nd);
        const f3 r = c - (o + t * d);
        t_out = t;
        return !is_equal_zero(nd) && (t > K_SHAPE_EPS) && (t < tmax) && (fsqrt(dot(r, r)) <= S.radius);
    }
}

// rectangle_t::intersect (1261-1297) for a planar parallelogram: plane hit, then dual-basis coordinates of the hit
// point; |u - 0.5| <= 0.5 and |v - 0.5| <= 0.5 is "inside".  A zero denominator gives inf / NaN, which compare false.
KY_DEV bool par_hit(const float4 q0, const float4 q1, const float4 q2, f3 o, f3 d, float tmax, float& t_out) {
    const float den = q0.x * d.x + q0.y * d.y + q0.z * d.z;
    const float num = q0.w - (q0.x * o.x + q0.y * o.y + q0.z * o.z);   // n.(p0 - o)
    const float t = num * rcp(den);
    const f3 h = o + t * d;
    const float u = (h.x * q1.x + h.y * q1.y + h.z * q1.z) - q1.w;
    const float v = (h.x * q2.x + h.y * q2.y + h.z * q2.z) - q2.w;
    t_out = t;
    return (fabsf(u) <= 0.5f) & (fabsf(v) <= 0.5f) & (t > K_SHAPE_EPS) & (t < tmax);
}

// the same up to the range tests: distance and the hit point's dual-basis coordinates minus one half (hit_update_nearest / hit_update_any test them)
// ... for a plank about the x axis (KY_FEAT_X_PLANKS): q0.x = q1.x = q2.y = q2.z = 0, the terms they multiply are left out.  The roundings are written out --
// explicit fused multiply-adds, and the products that must round on their own as instructions the compiler cannot fuse (mul_sv) -- because the sums must round where the
// general form's do: the compiler fuses a three-term sum x a + y b + z c as fma(z, c, fma(y, b, x a)), which with x = 0 is fma(z, c, round(y b)); left to itself it fuses
// the two-term sum the other way round, and a plank's hit point then moves by an ulp between the instantiations with and without this fact, which the planks'
// exponent-5000 lobe turns into 1e-3 of a highlight (tests/test_parity_gpu.py, test_engines_agree).  (Not `#pragma clang fp contract(off)`: one such pragma anywhere
// in the translation unit changes how the compiler contracts every OTHER function of it -- tests/test_random_scenes_gpu.py, room 8, caught two kernels of the table
// drifting apart by a decision flip.)
KY_DEV float mul_sv(float record_field, float x) {   // round(record_field * x): the field from its SGPR (records are read with wave-uniform indices)
    float r;
    asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "s"(record_field), "v"(x));
    return r;
}
KY_DEV void par_coords_x_plank(const float4 q0, const float4 q1, const float4 q2, f3 o, f3 d, float& t, float& u, float& v) {
    const float den = __builtin_fmaf(q0.z, d.z, mul_sv(q0.y, d.y));
    const float num = q0.w - __builtin_fmaf(q0.z, o.z, mul_sv(q0.y, o.y));
    t = num * rcp(den);
    const f3 h = mk3(__builtin_fmaf(t, d.x, o.x), __builtin_fmaf(t, d.y, o.y), __builtin_fmaf(t, d.z, o.z));
    u = __builtin_fmaf(q1.z, h.z, mul_sv(q1.y, h.y)) - q1.w;
    v = mul_sv(q2.x, h.x) - q2.w;
}
// `x_plank` (a compile-time constant at every call: KY_FEAT_X_PLANKS)
KY_DEV void par_coords(const float4 q0, const float4 q1, const float4 q2, f3 o, f3 d, float& t, float& u, float& v, bool x_plank = false) {
    if (x_plank) { par_coords_x_plank(q0, q1, q2, o, d, t, u, v); return; }
    const float den = q0.x * d.x + q0.y * d.y + q0.z * d.z;
    const float num = q0.w - (q0.x * o.x + q0.y * o.y + q0.z * o.z);   // n.(p0 - o)
    t = num * rcp(den);
    const f3 h = o + t * d;
    u = (h.x * q1.x + h.y * q1.y + h.z * q1.z) - q1.w;
    v = (h.x * q2.x + h.y * q2.y + h.z * q2.z) - q2.w;
}

// rectangle_t::intersect (1261-1297) for a rectangle in an axis plane (aar_scan below, and the lamp's test in estimate_by_bsdf): the plane hit is one
// subtraction and one multiply by the ray's reciprocal direction, the inside test needs only the two in-plane coordinates (12 VALU instead of 26).
// A zero direction component gives inf / NaN, which compare false.

// Scene tables are addressed as S + (32-bit byte offset): the scalar loads then take the scene pointer as their base and the offset from
// one SGPR (s_load_dwordx4 s[..], s[S:S+1], s_off offset:16), so no table needs a 64-bit pointer of its own.  As `T.aar + i` the compiler
// kept one hoisted base per table and loop alive across the whole path loop -- two SGPRs each, most of them spilled to VGPR lanes at the hot
// kernels' budget of 80 SGPRs and read back with two v_readlane per use.  The empty asm keeps the offset an offset (it stops the loop
// optimiser from turning base + offset back into a pointer that advances).
KY_DEV unsigned scene_off(SceneRef S, const void* q) { return (unsigned)((const char*)q - (const char*)S.p); }
template <class T>
KY_DEV const T& scene_at(SceneRef S, unsigned off) { return *(const T*)((const char*)S.p + off); }
// a record of the scene whose fields are read at several places of the path loop: the offset is (re)made where the record is used -- one
// s_mov / s_mul -- instead of one hoisted pointer per FIELD ADDRESS living (spilled) across the loop
KY_DEV unsigned opaque_off(unsigned off) { asm volatile("" : "+s"(off)); return off; }
KY_DEV const DLight& scene_light(SceneRef S, int li) { return scene_at<DLight>(S, opaque_off((unsigned)__builtin_offsetof(DScene, light) + (unsigned)li * (unsigned)sizeof(DLight))); }
KY_DEV const DSurf& scene_surf(SceneRef S, int i) { return scene_at<DSurf>(S, opaque_off((unsigned)__builtin_offsetof(DScene, all) + (unsigned)i * (unsigned)sizeof(DSurf))); }

// The four range tests of a hit and the update they guard, as a chain of v_cmpx: each compare NARROWS the exec mask to the lanes that passed, the
// update is two plain moves under what is left (the surface index straight from its SGPR), one s_mov restores the mask.  As C++ (`ok = a & b & c & d;
// tmax = ok ? t : tmax; best = ok ? i : best`) the compiler writes four v_cmp into SGPR pairs, three s_and_b64, a v_mov for the index and two
// v_cndmask: one VALU and three SALU instructions more per surface, in loops that run at 0.85 SALU per VALU instruction (the scalar unit retires
// an instruction per 1.84 ns against 1.1 for the vector unit: ky_amd DESIGN 3, "what bounds the kernel").  `ex` is the mask on entry (wave-uniform control
// flow inside the scan loops: the same for every surface of a scan).

KY_DEV void hit_update_nearest(unsigned long long ex, float u, float ru, float v, float rv, float t, float& tmax, int& best, int i) {
    unsigned long long tmp;
    asm volatile(
        "v_cmpx_le_f32_e64 %[tmp], |%[u]|, %[ru]\n\t"
        "v_cmpx_le_f32_e64 %[tmp], |%[v]|, %[rv]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[eps], %[t]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[t], %[tmax]\n\t"
        "v_mov_b32_e32 %[tmax], %[t]\n\t"
        "v_mov_b32_e32 %[best], %[i]\n\t"
        "s_mov_b64 exec, %[ex]"
        : [tmax] "+v"(tmax), [best] "+v"(best), [tmp] "=&s"(tmp)
        : [u] "v"(u), [ru] "s"(ru), [v] "v"(v), [rv] "s"(rv), [t] "v"(t), [eps] "s"(K_SHAPE_EPS), [i] "s"(i), [ex] "s"(ex));
}
KY_DEV void hit_update_any(unsigned long long ex, float u, float ru, float v, float rv, float t, float tmax, unsigned& occ) {
    unsigned long long tmp;
    asm volatile(
        "v_cmpx_le_f32_e64 %[tmp], |%[u]|, %[ru]\n\t"
        "v_cmpx_le_f32_e64 %[tmp], |%[v]|, %[rv]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[eps], %[t]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[t], %[tmax]\n\t"
        "v_mov_b32_e32 %[occ], 1\n\t"
        "s_mov_b64 exec, %[ex]"
        : [occ] "+v"(occ), [tmp] "=&s"(tmp)
        : [u] "v"(u), [ru] "s"(ru), [v] "v"(v), [rv] "s"(rv), [t] "v"(t), [tmax] "v"(tmax), [eps] "s"(K_SHAPE_EPS), [ex] "s"(ex));
}
// ... for a ray without an end (tmax = inf: "does the ray leave the scene"): every finite distance lies before it, and a distance that is not finite has failed the chain
// already (its in-plane coordinates are NaN or out of range), so the fourth compare is gone
KY_DEV void hit_update_any_unbounded(unsigned long long ex, float u, float ru, float v, float rv, float t, unsigned& occ) {
    unsigned long long tmp;
    asm volatile(
        "v_cmpx_le_f32_e64 %[tmp], |%[u]|, %[ru]\n\t"
        "v_cmpx_le_f32_e64 %[tmp], |%[v]|, %[rv]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[eps], %[t]\n\t"
        "v_mov_b32_e32 %[occ], 1\n\t"
        "s_mov_b64 exec, %[ex]"
        : [occ] "+v"(occ), [tmp] "=&s"(tmp)
        : [u] "v"(u), [ru] "s"(ru), [v] "v"(v), [rv] "s"(rv), [t] "v"(t), [eps] "s"(K_SHAPE_EPS), [ex] "s"(ex));
}

// Which form the rectangle loops take (a compile-time choice per instantiation, made by measurement: docs/rounds/round5.md): on the record's byte offset alone
// (one scalar add per record less) in the sphere-light instantiations -- configs[2] +0.75 % -- and on a counter + offset elsewhere, where the shorter form measured
// 0.3-0.7 % SLOWER on configs[1] (its register allocation spills one more SGPR in the bookkeeping block).
KY_DEV bool aar_by_offset(SceneRef S) { return S.sphere_lights(); }

// the axis-aligned rectangles of one axis: records [first, first + n) of the table at byte offset `aar_off`, whose sorted surface indices are the same
template <int AXIS, bool NEAREST>
KY_DEV void aar_scan(SceneRef S, unsigned aar_off, int first, int n, f3 o, f3 d, f3 inv_d, float& tmax, int& best, unsigned& occ_v) {
    if (n <= 0) return;
    unsigned off = aar_off + (unsigned)first * (unsigned)sizeof(DAar);
    const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);   // the exec mask here
    const float oa = AXIS == 0 ? o.x : (AXIS == 1 ? o.y : o.z), ia = AXIS == 0 ? inv_d.x : (AXIS == 1 ? inv_d.y : inv_d.z);
    const float ou_ = AXIS == 0 ? o.y : (AXIS == 1 ? o.z : o.x), du_ = AXIS == 0 ? d.y : (AXIS == 1 ? d.z : d.x);
    const float ov_ = AXIS == 0 ? o.z : (AXIS == 1 ? o.x : o.y), dv_ = AXIS == 0 ? d.z : (AXIS == 1 ? d.x : d.y);
    if (aar_by_offset(S)) {
        // the loop runs on the record's byte offset alone
        const unsigned end = off + (unsigned)n * (unsigned)sizeof(DAar);
        do {
            asm volatile("" : "+s"(off));
            const DAar& r = scene_at<DAar>(S, off);
            const float4 q0 = r.q0;
            const float rv = r.q1.x;
            const float t = (q0.x - oa) * ia;                 // aar_hit
            const float u = (ou_ + t * du_) - q0.y;
            const float v = (ov_ + t * dv_) - q0.w;
            if (NEAREST) hit_update_nearest(ex, u, q0.z, v, rv, t, tmax, best, __float_as_int(r.q1.y));   // the surface's sorted index travels in its record
            else hit_update_any(ex, u, q0.z, v, rv, t, tmax, occ_v);
            off += (unsigned)sizeof(DAar);
        } while (off != end);
        return;
    }
    int i = first;
    for (; i < first + n; ++i) {
        asm volatile("" : "+s"(off));
        const DAar& r = scene_at<DAar>(S, off);
        const float4 q0 = r.q0;
        const float rv = r.q1.x;
        const float t = (q0.x - oa) * ia;                 // aar_hit
        const float u = (ou_ + t * du_) - q0.y;
        const float v = (ov_ + t * dv_) - q0.w;
        if (NEAREST) hit_update_nearest(ex, u, q0.z, v, rv, t, tmax, best, __float_as_int(r.q1.y));   // the surface's sorted index travels in its record
        else hit_update_any(ex, u, q0.z, v, rv, t, tmax, occ_v);
        off += (unsigned)sizeof(DAar);
    }
}

// sphere_t::intersect, 1336-1393.  sqrt of a negative discriminant is NaN, which fails both range tests.
// The second half -- root, two distances, four range tests -- runs only when some lane's LINE meets the sphere (a wave-uniform branch on the discriminants' signs):
// for a lamp that subtends a thousandth of the sphere of directions that is one wavefront in fifty, and a Veach vertex tests five such lamps three times over.
// `sparse` (a compile-time constant at every call): the scene's spheres are small lamps (KY_FEAT_SPHERE_LIGHTS).  With the two big spheres of a Cornell box some lane's
// line nearly always meets the sphere and the branch only costs (configs[1] -1.3 %); with Veach's lamps it pays (configs[2] +1.9 %).
KY_DEV bool sph_hit(const float4 c, f3 o, f3 d, float tmax, float& t_out, bool sparse = false) {
    const f3 oc = mk3(c.x, c.y, c.z) - o;
    const float neg_b = dot(oc, d);
    const float discr = neg_b * neg_b - dot(oc, oc) + c.w;
    bool hit = false;
    t_out = neg_b;   // (read by no caller without a hit)
    if (!sparse || __any(discr >= 0.f)) {
        const float sq = fsqrt(discr);
        const float t0 = neg_b - sq, t1 = neg_b + sq;
        const bool h0 = (t0 > K_SHAPE_EPS) & (t0 < tmax);
        const bool h1 = (t1 > K_SHAPE_EPS) & (t1 < tmax);
        t_out = h0 ? t0 : t1;
        hit = h0 | h1;
    }
    return hit;
}

// The second half of sph_hit with its update, for the scan loops: the candidate root is t0 if it lies beyond the epsilon and t1 otherwise (t0 <= t1:
// if t0 > eps is too far, so is t1; if t0 <= eps only t1 can count) -- one compare and one select; then the v_cmpx chain of the planar tests.
// 6 VALU + 1 SALU for the nearest-hit update where four compares, three selects and the index move were 8 + 3; 5 + 1 for the any-hit flag where the lane masks'
// bookkeeping was 4 + 7.  (A NaN root -- the line misses the sphere -- fails both range tests as before.  The two wait states between a VALU write of VCC and
// the select that reads it are written out: the compiler's hazard recogniser does not look inside an asm statement.)
KY_DEV void sph_update_nearest(unsigned long long ex, float neg_b, float discr, float& tmax, int& best, int i) {
    const float sq = fsqrt(discr);
    const float t0 = neg_b - sq, t1 = neg_b + sq;
    unsigned long long tmp;
    float t;
    asm volatile(
        "v_cmp_lt_f32_e32 vcc, %[eps], %[t0]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e32 %[t], %[t1], %[t0], vcc\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[eps], %[t]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[t], %[tmax]\n\t"
        "v_mov_b32_e32 %[tmax], %[t]\n\t"
        "v_mov_b32_e32 %[best], %[i]\n\t"
        "s_mov_b64 exec, %[ex]"
        : [tmax] "+v"(tmax), [best] "+v"(best), [tmp] "=&s"(tmp), [t] "=&v"(t)
        : [t0] "v"(t0), [t1] "v"(t1), [eps] "s"(K_SHAPE_EPS), [i] "s"(i), [ex] "s"(ex)
        : "vcc");
}
KY_DEV void sph_update_any(unsigned long long ex, float neg_b, float discr, float tmax, unsigned& occ) {
    const float sq = fsqrt(discr);
    const float t0 = neg_b - sq, t1 = neg_b + sq;
    unsigned long long tmp;
    float t;
    asm volatile(
        "v_cmp_lt_f32_e32 vcc, %[eps], %[t0]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e32 %[t], %[t1], %[t0], vcc\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[eps], %[t]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[t], %[tmax]\n\t"
        "v_mov_b32_e32 %[occ], 1\n\t"
        "s_mov_b64 exec, %[ex]"
        : [occ] "+v"(occ), [tmp] "=&s"(tmp), [t] "=&v"(t)
        : [t0] "v"(t0), [t1] "v"(t1), [tmax] "v"(tmax), [eps] "s"(K_SHAPE_EPS), [ex] "s"(ex)
        : "vcc");
}
// ... without an end: some crossing lies beyond the epsilon iff the farther one does (a NaN root -- the line misses the sphere -- fails the compare)
KY_DEV void sph_update_any_unbounded(unsigned long long ex, float neg_b, float discr, unsigned& occ) {
    const float t1 = neg_b + fsqrt(discr);
    unsigned long long tmp;
    asm volatile(
        "v_cmpx_lt_f32_e64 %[tmp], %[eps], %[t1]\n\t"
        "v_mov_b32_e32 %[occ], 1\n\t"
        "s_mov_b64 exec, %[ex]"
        : [occ] "+v"(occ), [tmp] "=&s"(tmp)
        : [t1] "v"(t1), [eps] "s"(K_SHAPE_EPS), [ex] "s"(ex));
}

// A box's faces in one slab test (DBox, ky_scene.hpp; host: find_boxes).  A ray meets the boundary of a convex box at the two ends of the segment it has inside
// it: where it enters -- the LARGEST of the three near-plane distances -- and where it leaves -- the smallest of the three far-plane distances --, and it meets
// the box at all iff enter <= leave.  So the nearest hit among up to six rectangles that are whole faces is: the entry point if its face is a surface and its
// distance lies in (eps, tmax), else the exit point under the same conditions (a ray from inside, or one that came in through an open side).
// Which SURFACE a distance belongs to travels IN the distance: its four lowest mantissa bits are replaced by the face's sorted surface index (15: an open side)
// before the min / max network, which costs the hit distance up to fifteen units in the last place (1.8e-6 relative) and saves carrying six indices through
// twelve selects and looking the winner's surface up.  The distances are (c - o) * (1 / d) like the rectangle test's, with 1 / d clamped to +-1e30 so that a
// ray parallel to a pair of planes gives +-huge, not inf - inf.
// min / max as asm: the compiler's fminf / fmaxf add a canonicalising v_max_f32 x, x per operand it cannot prove quiet (the tagged values are bit patterns to it).
KY_DEV float vmin(float a, float b) { float r; asm("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
KY_DEV float vmax(float a, float b) { float r; asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
KY_DEV float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
KY_DEV float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
KY_DEV float face_tag(float t, float surface_bits) { return __uint_as_float((__float_as_uint(t) & ~15u) | __float_as_uint(surface_bits)); }   // v_and_or_b32 (the surface from its SGPR)
// one candidate (the entry or the exit point) as a v_cmpx chain like hit_update_nearest's: the box is met, eps < t < tmax, the face is a surface
KY_DEV void box_candidate(unsigned long long ex, float t_enter, float t_leave, float t, float& tmax, int& best) {
    unsigned long long tmp;
    const unsigned surface = __float_as_uint(t) & 15u;
    asm volatile(
        "v_cmpx_le_f32_e64 %[tmp], %[te], %[tl]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[eps], %[t]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[t], %[tmax]\n\t"
        "v_cmpx_ne_u32_e64 %[tmp], %[none], %[s]\n\t"
        "v_mov_b32_e32 %[tmax], %[t]\n\t"
        "v_mov_b32_e32 %[best], %[s]\n\t"
        "s_mov_b64 exec, %[ex]"
        : [tmax] "+v"(tmax), [best] "+v"(best), [tmp] "=&s"(tmp)
        : [te] "v"(t_enter), [tl] "v"(t_leave), [t] "v"(t), [s] "v"(surface), [eps] "s"(K_SHAPE_EPS), [none] "n"(KY_BOX_NO_FACE), [ex] "s"(ex));
}
KY_DEV void box_update_nearest(unsigned long long ex, const float4 q0, const float4 q1, const float4 q2, f3 o, f3 inv_c, float& tmax, int& best) {
    const float xl = face_tag((q0.x - o.x) * inv_c.x, q0.w), xh = face_tag((q1.x - o.x) * inv_c.x, q1.w);
    const float yl = face_tag((q0.y - o.y) * inv_c.y, q2.x), yh = face_tag((q1.y - o.y) * inv_c.y, q2.y);
    const float zl = face_tag((q0.z - o.z) * inv_c.z, q2.z), zh = face_tag((q1.z - o.z) * inv_c.z, q2.w);
    const float t_enter = vmax3(vmin(xl, xh), vmin(yl, yh), vmin(zl, zh));
    const float t_leave = vmin3(vmax(xl, xh), vmax(yl, yh), vmax(zl, zh));
    box_candidate(ex, t_enter, t_leave, t_enter, tmax, best);
    box_candidate(ex, t_enter, t_leave, t_leave, tmax, best);   // (after a hit at the entry point tmax <= t_leave: the chain's third compare keeps the entry)
}

// The same for an any-hit query (scene_t::occluded's scan, 3193-3195; an environment light's "does the ray leave the scene"): some face that is a surface is met inside
// (eps, tmax) iff the entry point or the exit point is such a face's -- two chains that set a flag instead of noting distance and surface.
KY_DEV void box_candidate_any(unsigned long long ex, float t_enter, float t_leave, float t, float tmax, unsigned& occ) {
    unsigned long long tmp;
    const unsigned surface = __float_as_uint(t) & 15u;
    asm volatile(
        "v_cmpx_le_f32_e64 %[tmp], %[te], %[tl]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[eps], %[t]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[t], %[tmax]\n\t"
        "v_cmpx_ne_u32_e64 %[tmp], %[none], %[s]\n\t"
        "v_mov_b32_e32 %[occ], 1\n\t"
        "s_mov_b64 exec, %[ex]"
        : [occ] "+v"(occ), [tmp] "=&s"(tmp)
        : [te] "v"(t_enter), [tl] "v"(t_leave), [t] "v"(t), [tmax] "v"(tmax), [s] "v"(surface), [eps] "s"(K_SHAPE_EPS), [none] "n"(KY_BOX_NO_FACE), [ex] "s"(ex));
}
KY_DEV void box_update_any(unsigned long long ex, const float4 q0, const float4 q1, const float4 q2, f3 o, f3 inv_c, float tmax, unsigned& occ) {
    const float xl = face_tag((q0.x - o.x) * inv_c.x, q0.w), xh = face_tag((q1.x - o.x) * inv_c.x, q1.w);
    const float yl = face_tag((q0.y - o.y) * inv_c.y, q2.x), yh = face_tag((q1.y - o.y) * inv_c.y, q2.y);
    const float zl = face_tag((q0.z - o.z) * inv_c.z, q2.z), zh = face_tag((q1.z - o.z) * inv_c.z, q2.w);
    const float t_enter = vmax3(vmin(xl, xh), vmin(yl, yh), vmin(zl, zh));
    const float t_leave = vmin3(vmax(xl, xh), vmax(yl, yh), vmax(zl, zh));
    box_candidate_any(ex, t_enter, t_leave, t_enter, tmax, occ);
    box_candidate_any(ex, t_enter, t_leave, t_leave, tmax, occ);
}
KY_DEV void box_candidate_any_unbounded(unsigned long long ex, float t_enter, float t_leave, float t, unsigned& occ) {
    unsigned long long tmp;
    const unsigned surface = __float_as_uint(t) & 15u;
    asm volatile(
        "v_cmpx_le_f32_e64 %[tmp], %[te], %[tl]\n\t"
        "v_cmpx_lt_f32_e64 %[tmp], %[eps], %[t]\n\t"
        "v_cmpx_ne_u32_e64 %[tmp], %[none], %[s]\n\t"
        "v_mov_b32_e32 %[occ], 1\n\t"
        "s_mov_b64 exec, %[ex]"
        : [occ] "+v"(occ), [tmp] "=&s"(tmp)
        : [te] "v"(t_enter), [tl] "v"(t_leave), [t] "v"(t), [s] "v"(surface), [eps] "s"(K_SHAPE_EPS), [none] "n"(KY_BOX_NO_FACE), [ex] "s"(ex));
}
// (the distances are at most 1e30 x a coordinate difference: finite, so "before the end" always holds)
KY_DEV void box_update_any_unbounded(unsigned long long ex, const float4 q0, const float4 q1, const float4 q2, f3 o, f3 inv_c, unsigned& occ) {
    const float xl = face_tag((q0.x - o.x) * inv_c.x, q0.w), xh = face_tag((q1.x - o.x) * inv_c.x, q1.w);
    const float yl = face_tag((q0.y - o.y) * inv_c.y, q2.x), yh = face_tag((q1.y - o.y) * inv_c.y, q2.y);
    const float zl = face_tag((q0.z - o.z) * inv_c.z, q2.z), zh = face_tag((q1.z - o.z) * inv_c.z, q2.w);
    const float t_enter = vmax3(vmin(xl, xh), vmin(yl, yh), vmin(zl, zh));
    const float t_leave = vmin3(vmax(xl, xh), vmax(yl, yh), vmax(zl, zh));
    box_candidate_any_unbounded(ex, t_enter, t_leave, t_enter, occ);
    box_candidate_any_unbounded(ex, t_enter, t_leave, t_leave, occ);
}

// one shape given as a generic record (KAT entry point, light shapes re-intersected by pdf_direction)
// `general` false: the caller knows the record is a parallelogram or a sphere (SceneRef::general)
KY_DEV bool surf_hit(const DSurf& S, const DShapeFull* __restrict__ full, f3 o, f3 d, float tmax, float& t_out, bool general = true, bool sphere_only = false, bool sparse = true) {
    if (sphere_only) return sph_hit(make_float4(S.f[0], S.f[1], S.f[2], S.f[3]), o, d, tmax, t_out, sparse);   // the caller knows (KY_FEAT_SPHERE_LIGHTS): small lamps, rarely met (sparse)
    if (S.kind == TK_PARALLELOGRAM)
        return par_hit(make_float4(S.f[0], S.f[1], S.f[2], S.f[3]), make_float4(S.f[4], S.f[5], S.f[6], S.f[7]), make_float4(S.f[8], S.f[9], S.f[10], S.f[11]), o, d, tmax, t_out);
    if (!general || S.kind == TK_SPHERE) return sph_hit(make_float4(S.f[0], S.f[1], S.f[2], S.f[3]), o, d, tmax, t_out);
    return full_shape_hit(full[S.full], o, d, tmax, t_out);
}

// scene_t::intersect, ky.cpp:3172-3184: linear scan, tmax shrinks, first of equals wins.  Returns the SORTED surface index.
KY_DEV int trace_nearest(SceneRef S, f3 o, f3 d, float& tmax) {
    int best = -1;
    const unsigned t_off = opaque_off((unsigned)__builtin_offsetof(DScene, trav));
    const int4 head = scene_at<int4>(S, t_off), axis = scene_at<int4>(S, t_off + 16u);   // n_aar, n_par; n_aar_axis[3]
    const int n_aar = head.x, n_par = S.no_par() ? 0 : head.y, n_sph = S->n_sph, n_gen = S->n_gen;
    if (S.boxes()) {   // KY_FEAT_BOXES: the boxes whole, then the rectangles that are no box's face (DScene::boxtrav)
        const unsigned b_off = opaque_off((unsigned)__builtin_offsetof(DScene, boxtrav));
        const int4 bhead = scene_at<int4>(S, b_off), baxis = scene_at<int4>(S, b_off + 16u);   // n_box, n_aar; n_aar_axis[3]
        const f3 inv_d = mk3(rcp(d.x), rcp(d.y), rcp(d.z));
        const f3 inv_c = mk3(__builtin_amdgcn_fmed3f(inv_d.x, -1e30f, 1e30f), __builtin_amdgcn_fmed3f(inv_d.y, -1e30f, 1e30f), __builtin_amdgcn_fmed3f(inv_d.z, -1e30f, 1e30f));
        const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);
        unsigned off = b_off + (unsigned)__builtin_offsetof(DBoxTrav, box);
        for (int k = 0; k < bhead.x; ++k) {
            asm volatile("" : "+s"(off));
            const DBox& B = scene_at<DBox>(S, off);
            box_update_nearest(ex, B.q0, B.q1, B.q2, o, inv_c, tmax, best);
            off += (unsigned)sizeof(DBox);
        }
        if (bhead.y > 0) {
            unsigned unused = 0;
            const int n0 = baxis.x, n1 = baxis.y, n2 = baxis.z;
            const unsigned aar_off = b_off + (unsigned)__builtin_offsetof(DBoxTrav, aar);
            aar_scan<0, true>(S, aar_off, 0, n0, o, d, inv_d, tmax, best, unused);
            aar_scan<1, true>(S, aar_off, n0, n1, o, d, inv_d, tmax, best, unused);
            aar_scan<2, true>(S, aar_off, n0 + n1, n2, o, d, inv_d, tmax, best, unused);
        }
    } else
    if (n_aar > 0) {
        const f3 inv_d = mk3(rcp(d.x), rcp(d.y), rcp(d.z));
        unsigned unused = 0;
        const int n0 = axis.x, n1 = axis.y, n2 = axis.z;
        const unsigned aar_off = t_off + (unsigned)__builtin_offsetof(DTrav, aar);
        aar_scan<0, true>(S, aar_off, 0, n0, o, d, inv_d, tmax, best, unused);
        aar_scan<1, true>(S, aar_off, n0, n1, o, d, inv_d, tmax, best, unused);
        aar_scan<2, true>(S, aar_off, n0 + n1, n2, o, d, inv_d, tmax, best, unused);
    }
    if (n_par > 0) {
        unsigned off = t_off + (unsigned)__builtin_offsetof(DTrav, par);
        const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);
        for (int i = 0; i < n_par; ++i) {
            asm volatile("" : "+s"(off));
            const DPar& r = scene_at<DPar>(S, off);
            float t, u, v;
            // (the planks' short form, KY_FEAT_X_PLANKS, is taken by the any-hit scans only: here it is worth another 0.5 % of configs[2], but the sphere-lights kernel's
            // allocation then spills three registers around its five-light loop -- 60 -> 106 GB of memory-side traffic per 9.4e8 samples, docs/rounds/round6.md section 3)
            par_coords(r.q0, r.q1, r.q2, o, d, t, u, v);
            hit_update_nearest(ex, u, 0.5f, v, 0.5f, t, tmax, best, n_aar + i);
            off += (unsigned)sizeof(DPar);
        }
    }
    if (n_sph > 0) {
        unsigned off = scene_off(S, &S->sph[0]);
        const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);
        for (int i = 0; i < n_sph; ++i) {
            asm volatile("" : "+s"(off));
            const float4 c = scene_at<DSph>(S, off).c;
            const f3 oc = mk3(c.x, c.y, c.z) - o;
            const float neg_b = dot(oc, d);
            const float discr = neg_b * neg_b - dot(oc, oc) + c.w;
            if (!S.sphere_lights() || __any(discr >= 0.f)) sph_update_nearest(ex, neg_b, discr, tmax, best, n_aar + n_par + i);   // (sph_hit's `sparse` rule)
            off += (unsigned)sizeof(DSph);
        }
    }
    for (int i = 0; S.general && i < n_gen; ++i) {
        float t;
        if (full_shape_hit(S->full[S->gen[i].full], o, d, tmax, t)) {
            tmax = t;
            best = n_aar + n_par + n_sph + i;
        }
    }
    return best;
}

// scene_t::occluded's traversal (3193-3195): any hit inside (eps, tmax) occludes.  `T` (wave-uniform): the table of planar surfaces
// to test -- S->trav (all), or the occluder table when the ray qualifies for it (DScene::occ).
KY_DEV bool trace_any_planar(SceneRef S, const DTrav& T, f3 o, f3 d, float tmax) {
    unsigned occ = 0;   // a lane flag in a VGPR: the scans OR into it (hit_update_any)
    const unsigned t_off = opaque_off(scene_off(S, &T));
    const int4 head = scene_at<int4>(S, t_off), axis = scene_at<int4>(S, t_off + 16u);   // n_aar, n_par; n_aar_axis[3]
    const int n_aar = head.x, n_par = S.no_par() ? 0 : head.y;
    if (n_aar > 0) {
        const f3 inv_d = mk3(rcp(d.x), rcp(d.y), rcp(d.z));
        int unused = -1;
        const int n0 = axis.x, n1 = axis.y, n2 = axis.z;
        const unsigned aar_off = t_off + (unsigned)__builtin_offsetof(DTrav, aar);
        aar_scan<0, false>(S, aar_off, 0, n0, o, d, inv_d, tmax, unused, occ);
        aar_scan<1, false>(S, aar_off, n0, n1, o, d, inv_d, tmax, unused, occ);
        aar_scan<2, false>(S, aar_off, n0 + n1, n2, o, d, inv_d, tmax, unused, occ);
    }
    if (n_par > 0) {
        unsigned off = t_off + (unsigned)__builtin_offsetof(DTrav, par);
        const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);
        for (int i = 0; i < n_par; ++i) {
            asm volatile("" : "+s"(off));
            const DPar& r = scene_at<DPar>(S, off);
            float t, u, v;
            par_coords(r.q0, r.q1, r.q2, o, d, t, u, v, S.x_planks());
            hit_update_any(ex, u, 0.5f, v, 0.5f, t, tmax, occ);
            off += (unsigned)sizeof(DPar);
        }
    }
    return occ != 0;
}
// the part of an any-hit scan no table prunes: spheres and general shapes
KY_DEV bool trace_any_round(SceneRef S, bool occ, f3 o, f3 d, float tmax) {
    const int n_sph = S->n_sph, n_gen = S->n_gen;
    float t;
    if (n_sph > 0) {
        unsigned off = scene_off(S, &S->sph[0]);
        const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);
        unsigned occ_v = 0;
        for (int i = 0; i < n_sph; ++i) {
            asm volatile("" : "+s"(off));
            const float4 c = scene_at<DSph>(S, off).c;
            const f3 oc = mk3(c.x, c.y, c.z) - o;
            const float neg_b = dot(oc, d);
            const float discr = neg_b * neg_b - dot(oc, oc) + c.w;
            // sphere-light scenes: a shadow ray is AIMED at a lamp, so its line always meets that lamp -- 2e-3 beyond tmax (3187-3201) -- and with five lamps every sphere is
            // some lane's target.  The root is needed only if some lane's nearer crossing can lie before tmax: neg_b - sqrt(discr) < tmax <=> e < 0 or e^2 < discr, e = neg_b - tmax
            // (no square root; conservative: the far crossing and eps are left to the full test).
            const float e = neg_b - tmax;
            if (!S.sphere_lights() || __any((discr >= 0.f) & ((e < 0.f) | (e * e < discr)))) sph_update_any(ex, neg_b, discr, tmax, occ_v);
            off += (unsigned)sizeof(DSph);
        }
        occ = occ | (occ_v != 0);
    }
    for (int i = 0; S.general && i < n_gen; ++i) occ = occ || full_shape_hit(S->full[S->gen[i].full], o, d, tmax, t);
    return occ;
}
KY_DEV bool trace_any(SceneRef S, const DTrav& T, f3 o, f3 d, float tmax) {
    return trace_any_round(S, trace_any_planar(S, T, o, d, tmax), o, d, tmax);
}
// Two any-hit queries of the same lanes over EVERY surface in one scan (an environment light's both_mis estimate: the BSDF-sampled ray and the light-sampled
// ray both ask "does it leave the scene"): every record is loaded once and the loops' scalar bookkeeping runs once for the two rays -- in loops that run at one
// scalar instruction per two vector ones that is a quarter of their issue slots.  The tests are trace_any_all's, instruction for instruction (the reciprocal
// direction clamped to +-1e30 for the rectangles too: it differs from the unclamped one only for |d| < 1e-30, where both give "no hit").
struct AnyRay {
    f3 o, d;
    float tmax;
};
template <int AXIS>
KY_DEV void aar_scan_any_pair(SceneRef S, unsigned long long ex, unsigned aar_off, int first, int n, const AnyRay& A, f3 iA, unsigned& occA, const AnyRay& B, f3 iB, unsigned& occB) {
    if (n <= 0) return;
    unsigned off = aar_off + (unsigned)first * (unsigned)sizeof(DAar);
    const float oaA = AXIS == 0 ? A.o.x : (AXIS == 1 ? A.o.y : A.o.z), iaA = AXIS == 0 ? iA.x : (AXIS == 1 ? iA.y : iA.z);
    const float ouA = AXIS == 0 ? A.o.y : (AXIS == 1 ? A.o.z : A.o.x), duA = AXIS == 0 ? A.d.y : (AXIS == 1 ? A.d.z : A.d.x);
    const float ovA = AXIS == 0 ? A.o.z : (AXIS == 1 ? A.o.x : A.o.y), dvA = AXIS == 0 ? A.d.z : (AXIS == 1 ? A.d.x : A.d.y);
    const float oaB = AXIS == 0 ? B.o.x : (AXIS == 1 ? B.o.y : B.o.z), iaB = AXIS == 0 ? iB.x : (AXIS == 1 ? iB.y : iB.z);
    const float ouB = AXIS == 0 ? B.o.y : (AXIS == 1 ? B.o.z : B.o.x), duB = AXIS == 0 ? B.d.y : (AXIS == 1 ? B.d.z : B.d.x);
    const float ovB = AXIS == 0 ? B.o.z : (AXIS == 1 ? B.o.x : B.o.y), dvB = AXIS == 0 ? B.d.z : (AXIS == 1 ? B.d.x : B.d.y);
    for (int i = 0; i < n; ++i) {
        asm volatile("" : "+s"(off));
        const DAar& r = scene_at<DAar>(S, off);
        const float4 q0 = r.q0;
        const float rv = r.q1.x;
        const float tA = (q0.x - oaA) * iaA, tB = (q0.x - oaB) * iaB;
        const float uA = (ouA + tA * duA) - q0.y, uB = (ouB + tB * duB) - q0.y;
        const float vA = (ovA + tA * dvA) - q0.w, vB = (ovB + tB * dvB) - q0.w;
        hit_update_any_unbounded(ex, uA, q0.z, vA, rv, tA, occA);
        hit_update_any(ex, uB, q0.z, vB, rv, tB, B.tmax, occB);
        off += (unsigned)sizeof(DAar);
    }
}
// A: a ray without an end (its tmax is not read: the chains without the fourth compare); B: a ray that ends at B.tmax
KY_DEV void trace_any_pair(SceneRef S, const AnyRay& A, const AnyRay& B, bool& occ_a, bool& occ_b) {
    unsigned occA = 0, occB = 0;
    const f3 iA = mk3(__builtin_amdgcn_fmed3f(rcp(A.d.x), -1e30f, 1e30f), __builtin_amdgcn_fmed3f(rcp(A.d.y), -1e30f, 1e30f), __builtin_amdgcn_fmed3f(rcp(A.d.z), -1e30f, 1e30f));
    const f3 iB = mk3(__builtin_amdgcn_fmed3f(rcp(B.d.x), -1e30f, 1e30f), __builtin_amdgcn_fmed3f(rcp(B.d.y), -1e30f, 1e30f), __builtin_amdgcn_fmed3f(rcp(B.d.z), -1e30f, 1e30f));
    const unsigned long long ex = __builtin_amdgcn_ballot_w64(true);
    const unsigned t_off = opaque_off((unsigned)__builtin_offsetof(DScene, trav));
    if (S.boxes()) {
        const unsigned b_off = opaque_off((unsigned)__builtin_offsetof(DScene, boxtrav));
        const int4 bhead = scene_at<int4>(S, b_off), baxis = scene_at<int4>(S, b_off + 16u);   // n_box, n_aar; n_aar_axis[3]
        unsigned off = b_off + (unsigned)__builtin_offsetof(DBoxTrav, box);
        for (int k = 0; k < bhead.x; ++k) {
            asm volatile("" : "+s"(off));
            const DBox& Bx = scene_at<DBox>(S, off);
            const float4 q0 = Bx.q0, q1 = Bx.q1, q2 = Bx.q2;
            box_update_any_unbounded(ex, q0, q1, q2, A.o, iA, occA);
            box_update_any(ex, q0, q1, q2, B.o, iB, B.tmax, occB);
            off += (unsigned)sizeof(DBox);
        }
        if (bhead.y > 0) {
            const unsigned aar_off = b_off + (unsigned)__builtin_offsetof(DBoxTrav, aar);
            aar_scan_any_pair<0>(S, ex, aar_off, 0, baxis.x, A, iA, occA, B, iB, occB);
            aar_scan_any_pair<1>(S, ex, aar_off, baxis.x, baxis.y, A, iA, occA, B, iB, occB);
            aar_scan_any_pair<2>(S, ex, aar_off, baxis.x + baxis.y, baxis.z, A, iA, occA, B, iB, occB);
        }
    } else {
        const int4 head = scene_at<int4>(S, t_off), axis = scene_at<int4>(S, t_off + 16u);   // n_aar, n_par; n_aar_axis[3]
        if (head.x > 0) {
            const unsigned aar_off = t_off + (unsigned)__builtin_offsetof(DTrav, aar);
            aar_scan_any_pair<0>(S, ex, aar_off, 0, axis.x, A, iA, occA, B, iB, occB);
            aar_scan_any_pair<1>(S, ex, aar_off, axis.x, axis.y, A, iA, occA, B, iB, occB);
            aar_scan_any_pair<2>(S, ex, aar_off, axis.x + axis.y, axis.z, A, iA, occA, B, iB, occB);
        }
    }
    const int n_par = S.no_par() ? 0 : S->trav.n_par;
    if (n_par > 0) {
        unsigned off = t_off + (unsigned)__builtin_offsetof(DTrav, par);
        for (int i = 0; i < n_par; ++i) {
            asm volatile("" : "+s"(off));
            const DPar& r = scene_at<DPar>(S, off);
            float t, u, v;
            par_coords(r.q0, r.q1, r.q2, A.o, A.d, t, u, v, S.x_planks());
            hit_update_any_unbounded(ex, u, 0.5f, v, 0.5f, t, occA);
            par_coords(r.q0, r.q1, r.q2, B.o, B.d, t, u, v, S.x_planks());
            hit_update_any(ex, u, 0.5f, v, 0.5f, t, B.tmax, occB);
            off += (unsigned)sizeof(DPar);
        }
    }
    const int n_sph = S->n_sph, n_gen = S->n_gen;
    if (n_sph > 0) {
        unsigned off = scene_off(S, &S->sph[0]);
        for (int i = 0; i < n_sph; ++i) {
            asm volatile("" : "+s"(off));
            const float4 c = scene_at<DSph>(S, off).c;
            const f3 ocA = mk3(c.x, c.y, c.z) - A.o, ocB = mk3(c.x, c.y, c.z) - B.o;
            const float nbA = dot(ocA, A.d), nbB = dot(ocB, B.d);
            sph_update_any_unbounded(ex, nbA, nbA * nbA - dot(ocA, ocA) + c.w, occA);
            sph_update_any(ex, nbB, nbB * nbB - dot(ocB, ocB) + c.w, B.tmax, occB);
            off += (unsigned)sizeof(DSph);
        }
    }
    occ_a = occA != 0;
    occ_b = occB != 0;
    float t;
    for (int i = 0; S.general && i < n_gen; ++i) {
        occ_a = occ_a || full_shape_hit(S->full[S->gen[i].full], A.o, A.d, A.tmax, t);
        occ_b = occ_b || full_shape_hit(S->full[S->gen[i].full], B.o, B.d, B.tmax, t);
    }
}

// normal the shape reports for a hit (1125, 1208, 1289, 1389)
KY_DEV f3 hit_normal(const DHit& H, f3 position, f3 d) {
    const f3 n = ld3(H.n);
    if (H.kind == KY_SHAPE_SPHERE) return normalize(position - n);
    if (H.kind == KY_SHAPE_RECTANGLE) {   // the normal that faces the ray (1289): the sign bit of all three components flipped by one mask, no divergent region
        const unsigned flip = dot(n, d) <= 0 ? 0u : 0x80000000u;
        return mk3(__uint_as_float(__float_as_uint(n.x) ^ flip), __uint_as_float(__float_as_uint(n.y) ^ flip), __uint_as_float(__float_as_uint(n.z) ^ flip));
    }
    return n;
}

// ---------------------------------------------------------------------------------------------
// frame_t (ky.cpp:526-578).  `normal` is unit by construction, so frame_t's own normalize (538) is skipped;
// cross(n, X) = (0, n.z, -n.y) and cross(n, Y) = (-n.z, 0, n.x) are written out.
// ---------------------------------------------------------------------------------------------
struct Frame {
    f3 s, t, n;
};
KY_DEV Frame make_frame(f3 n) {
    Frame f;
    f.n = n;
    if (fabsf(n.x) > 0.99f) {
        const float k = rsq(n.z * n.z + n.x * n.x);
        f.t = mk3(-n.z * k, 0.f, n.x * k);
    } else {
        const float k = rsq(n.z * n.z + n.y * n.y);
        f.t = mk3(0.f, n.z * k, -n.y * k);
    }
    f.s = cross(f.t, n);  // unit: t is unit and perpendicular to n
    return f;
}
KY_DEV f3 to_local(const Frame& f, f3 w) { return {dot(f.s, w), dot(f.t, w), dot(f.n, w)}; }
KY_DEV f3 to_world(const Frame& f, f3 l) { return f.s * l.x + f.t * l.y + f.n * l.z; }

// ---------------------------------------------------------------------------------------------
// BSDFs (ky.cpp:2092-2555) -- evaluated in WORLD space.
//
// The reference carries wo and wi into the shading frame of the hit (bsdf_t::to_local, 2162-2186), works there, and
// carries the sampled wi back.  Nothing a lobe computes depends on the frame's tangents except the ORIENTATION of the
// sampled direction around the lobe's axis, so here a vertex keeps, once, the world-space basis its lobe samples in:
//   Lambert   (a, b, c) = (s, t, n) of frame_t(n):  wi = to_world(frame, cosine_hemisphere_sample(u))   (2176, 737-743)
//   Phong     the frame around the mirror direction wr (2533-2543), built in the shading frame exactly as the reference
//             builds it and then carried to world space: a = to_world(frame, fr.s), b = to_world(frame, fr.t),
//             c = to_world(frame, wr), fr = frame_t(wr)
// and every sample of either lobe is  wi = a x + b y + c z  with (x, y, z) from the lobe's own mapping: one piece of code for the
// lanes of both lobes instead of two run one after the other.  Values and pdfs need three cosines: n.wo, n.wi and -- for the Phong
// lobe -- wr.wi = c.wi (for the Lambert lobe c = n, so c.wi IS n.wi).  The two delta lobes need no basis at all: reflect
// and refract are frame-independent.  Results agree with the local-space formulation to rounding (tests/test_parity_gpu.py).
// ---------------------------------------------------------------------------------------------
enum : int { LOBE_LAMBERT = 0, LOBE_MIRROR = 1, LOBE_GLASS = 2, LOBE_PHONG = 3 };
enum : int { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16 };

// The BSDF a material builds at a hit (material_t::scattering x4: 2587, 2604, 2628, 2661) is a lobe plus the material's
// record: colours and exponents are read from the record (LDS) where they are used instead of living in registers.
struct Bsdf {
    int lobe;
    const DMat* m;
};
KY_DEV bool bsdf_is_delta(const Bsdf& B) { return B.lobe == LOBE_MIRROR || B.lobe == LOBE_GLASS; }

// material_t::scattering's choice of lobe (2587, 2604, 2628, 2661): the material kinds and the lobes are numbered alike (matte / Lambert 0, mirror 1,
// glass 2, plastic / Phong 3), so the lobe IS the kind unless a plastic material's draw falls on its diffuse side (2663): one compare and one select.
static_assert((int)KY_MATERIAL_MATTE == (int)LOBE_LAMBERT && (int)KY_MATERIAL_MIRROR == (int)LOBE_MIRROR && (int)KY_MATERIAL_GLASS == (int)LOBE_GLASS &&
              (int)KY_MATERIAL_PLASTIC == (int)LOBE_PHONG, "pick_lobe");
KY_DEV int pick_lobe(const DMat& M, float lobe_random, bool no_delta = false) {
    (void)no_delta;
    const bool diffuse_side = (M.kind == KY_MATERIAL_PLASTIC) & !(lobe_random < M.p_specular);
    return diffuse_side ? (int)LOBE_LAMBERT : M.kind;
}
KY_DEV Bsdf make_bsdf_for_lobe(const DMat& M, int lobe) { return Bsdf{lobe, &M}; }
KY_DEV Bsdf make_bsdf(const DMat& M, float lobe_random, bool no_delta = false) { return Bsdf{pick_lobe(M, lobe_random, no_delta), &M}; }

// std::pow(base, exponent) of the Phong lobe (2499): a negative base is legal for an integral exponent
KY_DEV float phong_pow(float base, float exponent, int exp_flags) {
    const float m = pow_nonneg(fabsf(base), exponent);
    if (base >= 0.f) return m;
    if (!(exp_flags & 1)) return __builtin_nanf("");   // pow(negative, non-integer)
    return (exp_flags & 2) ? -m : m;
}

// The same for the lanes of a wavefront that evaluate their Phong lobes together (bsdf_eval_pdf / bsdf_eval_parts): away from the lobe's axis the power underflows
// to exactly zero -- exp2(5000 x log2(0.98)) = 2^-146 -- and for a light sample that is nearly every evaluation.  The host leaves the material a bound F with
// pow(|x|, exponent) == 0 for |x| <= F in this very arithmetic (DMat::exp_flags, upper half); when NO lane of the vote lies beyond it -- nor holds a NaN, nor a negative
// base of a non-integral power, which is NaN -- the two quarter-rate instructions and the sign logic are skipped and every lane takes the zero it would have computed
// (+0 where the odd power of a negative base gives -0: no caller can tell them apart).  configs[2]: +0.5 % (46.48 -> 46.23 ms at 512 spp); images unchanged.
KY_DEV float phong_pow_lobe(float base, const DMat& M) {
    const int fl = M.exp_flags;
    const float zero_below = __uint_as_float((unsigned)fl & 0xffff0000u);
    const bool need = !(fabsf(base) <= zero_below) | ((base < 0.f) & !(fl & 1));
    float m = 0.f;
    if (__any(need)) m = phong_pow(base, M.exponent, fl);
    return m;
}

struct BsdfSample {
    f3 f, wi;   // wi in world space
    float pdf;
    int flags;
};

// concentric_disk_sample, 710-733 (angles in revolutions: theta / 2 pi)
KY_DEV void concentric_disk(float u0, float u1, float& px, float& py) {
    const float rx = 2.f * u0 - 1, ry = 2.f * u1 - 1;
    const bool xmajor = fabsf(rx) > fabsf(ry);
    const float radius = xmajor ? rx : ry;
    const float ratio = mul_legacy(xmajor ? ry : rx, rcp(radius));    // rx = ry = 0: 0 x (1 / 0) = 0, and the radius is 0
    const float rev = xmajor ? 0.125f * ratio : 0.25f - 0.125f * ratio;  // (pi/4) q, pi/2 - (pi/4) q
    px = cos_rev(rev) * radius;
    py = sin_rev(rev) * radius;
}

// ---------------------------------------------------------------------------------------------
// path vertex (isect_t, 642-690)
// ---------------------------------------------------------------------------------------------
struct LobeBasis {
    f3 a, b, c;
};
// The lane engine keeps a vertex's lobe basis in LDS (read back where a direction is sampled or a Phong value is needed)
// instead of in nine registers that would be live -- in practice: spilled to scratch memory -- across the whole lights loop.
struct VertexLds {
    float a[3][256], b[3][256], c[3][256];   // [component][thread of the workgroup]
};
__shared__ VertexLds g_vertex_lds;   // allocated only in kernels that set Vertex::in_lds
struct Vertex {
    float t;      // distance along the ray that found the vertex
    f3 position, normal;
    LobeBasis basis;   // used when !in_lds
    Bsdf bsdf;
    int surface;
    bool in_lds = false;
};
KY_DEV void vertex_set_basis(Vertex& v, const LobeBasis& L) {
    if (v.in_lds) {
        const int i = threadIdx.x;
        g_vertex_lds.a[0][i] = L.a.x; g_vertex_lds.a[1][i] = L.a.y; g_vertex_lds.a[2][i] = L.a.z;
        g_vertex_lds.b[0][i] = L.b.x; g_vertex_lds.b[1][i] = L.b.y; g_vertex_lds.b[2][i] = L.b.z;
        g_vertex_lds.c[0][i] = L.c.x; g_vertex_lds.c[1][i] = L.c.y; g_vertex_lds.c[2][i] = L.c.z;
    } else {
        v.basis = L;
    }
}
KY_DEV LobeBasis vertex_basis(const Vertex& v) {
    if (!v.in_lds) return v.basis;
    const int i = threadIdx.x;
    LobeBasis L;
    L.a = mk3(g_vertex_lds.a[0][i], g_vertex_lds.a[1][i], g_vertex_lds.a[2][i]);
    L.b = mk3(g_vertex_lds.b[0][i], g_vertex_lds.b[1][i], g_vertex_lds.b[2][i]);
    L.c = mk3(g_vertex_lds.c[0][i], g_vertex_lds.c[1][i], g_vertex_lds.c[2][i]);
    return L;
}
KY_DEV f3 vertex_basis_c(const Vertex& v) {
    if (!v.in_lds) return v.basis.c;
    const int i = threadIdx.x;
    return mk3(g_vertex_lds.c[0][i], g_vertex_lds.c[1][i], g_vertex_lds.c[2][i]);
}

// frame_t(isect.normal) (2100) of a vertex on the surface `H` describes.  A planar shape reports its stored normal or, a rectangle seen from behind, the
// opposite one (1289): frame_t(n) = (s, t, n) and frame_t(-n) = (s, -t, -n) exactly (t = normalize(n x axis) changes sign, s = t x n does not), so the host
// builds (s, t) once per surface and a vertex reads them -- no frame build, no rsq, at the planar hits that are all of a Cornell box's non-specular
// vertices.  Spheres (a normal per hit) and callers without a surface record (the BSDF KATs: any normal) build the frame here.
KY_DEV Frame surface_frame(const DHit* H, f3 n) {
    if (H != nullptr && H->kind != KY_SHAPE_SPHERE) {
        const bool turned = (n.x != H->n[0]) | (n.y != H->n[1]) | (n.z != H->n[2]);   // n is the stored normal or its negative, bit for bit
        const float sg = turned ? -1.f : 1.f;
        return Frame{ld3(H->fs), ld3(H->ft) * sg, n};
    }
    return make_frame(n);
}

// The basis of a non-delta vertex (see the section comment).  wo = -ray.direction (3125), unit.
KY_DEV LobeBasis make_lobe_basis(const Bsdf& B, f3 n, f3 wo, const DHit* H = nullptr) {
    const Frame fr = surface_frame(H, n);   // frame_t(isect.normal), 2100
    LobeBasis L{fr.s, fr.t, n};
    if (B.lobe == LOBE_PHONG) {       // 2533-2536: wr = reflect(wo, z) in the shading frame, frame_t(wr) there
        const f3 wo_l = to_local(fr, wo);
        const f3 wr = mk3(-wo_l.x, -wo_l.y, wo_l.z);   // unit because wo is
        // frame_t(wr): t = normalize(cross(wr, X or Y)), s = cross(t, wr).  t has a zero component, and the shading frame is a rotation
        // (s x t = n), so cross products may be taken in world space: b = to_world(t), c = to_world(wr), a = cross(b, c) -- one frame
        // transform of a vector with a zero component, one of wr, one cross product instead of a frame build and three transforms.
        L.c = (2.f * wo_l.z) * n - wo;   // to_world(fr, wr) = reflect(wo, n) (1923): the frame is a rotation
        if (fabsf(wr.x) > 0.99f) {
            const float k = rsq(wr.z * wr.z + wr.x * wr.x);
            L.b = fr.s * (-wr.z * k) + fr.n * (wr.x * k);       // t = (-wr.z, 0, wr.x) k
        } else {
            const float k = rsq(wr.z * wr.z + wr.y * wr.y);
            L.b = fr.t * (wr.z * k) + fr.n * (-wr.y * k);       // t = (0, wr.z, -wr.y) k
        }
        L.a = cross(L.b, L.c);
    }
    return L;
}
// scattering + the vertex's basis: what an (active) lane does once per vertex after v.bsdf is known
KY_DEV void vertex_prepare(Vertex& v, f3 wo, const DHit* H = nullptr) {
    if (!bsdf_is_delta(v.bsdf)) vertex_set_basis(v, make_lobe_basis(v.bsdf, v.normal, wo, H));
}

// eval_ and pdf_ of the two non-delta lobes at one (wo, wi) pair, world space (2227-2240, 2489-2508, 2545-2550); the delta
// lobes evaluate to 0 / 0 (2289-2290, 2352-2353).  abs_cos_i = |dot(wi, isect.normal)|, the factor every caller multiplies f by.
KY_DEV void bsdf_eval_pdf(const Vertex& v, f3 wo, f3 wi, f3& f, float& pdf, float& abs_cos_i) {
    const Bsdf& B = v.bsdf;
    const float cos_o = dot(v.normal, wo), cos_i = dot(v.normal, wi);   // wo.z, wi.z of the shading frame
    abs_cos_i = fabsf(cos_i);
    const bool same = cos_o * cos_i > 0;  // same_hemisphere, 1921
    // Two arms that both assign everything, then selects: a default of zero overwritten under nested conditions costs a v_mov per
    // register and nesting level in every caller (the listing of the deferred estimator had four groups of four).
    f3 col;
    float scale, p;
    if (B.lobe == LOBE_PHONG) {
        const float cos_alpha = dot(vertex_basis_c(v), wi);   // dot(wr, wi), 2497
        // eval: cos_alpha is not clamped (a negative base with an even integral exponent is positive);
        // pdf: clamped at 0, no hemisphere test (quirk 6)
        const float exponent = B.m->exponent;
        const float pe = phong_pow_lobe(cos_alpha, *B.m);
        const float p0 = exponent == 0.f ? 1.f : (exponent > 0.f ? 0.f : K_INF);
        col = ld3(B.m->cs) * B.m->inv_eta;                          // (exponent + 2) / 2 pi
        scale = same ? pe : 0.f;
        p = (cos_alpha > 0.f ? pe : p0) * B.m->phong_pdf_norm;      // (exponent + 1) / 2 pi
    } else {   // Lambert; the delta lobes (eval 0, pdf 0: 2289-2290, 2352-2353) pass through and are zeroed below
        col = ld3(B.m->c0);
        scale = same ? K_INV_PI : 0.f;
        p = same ? abs_cos_i * K_INV_PI : 0.f;
    }
    const bool nondelta = !bsdf_is_delta(B);
    scale = nondelta ? scale : 0.f;
    f = col * scale;
    pdf = nondelta ? p : 0.f;
}

// The same as two factors, f = col x scale, for callers that multiply the value by a chain of scalars (the light-sampling estimators: |cos|, the MIS
// factor, the strategy weight): the scalars are multiplied first and the three colour channels once.  col: the material's colour as stored.
KY_DEV void bsdf_eval_parts(const Vertex& v, f3 wo, f3 wi, f3& col, float& scale, float& pdf, float& abs_cos_i) {
    const Bsdf& B = v.bsdf;
    const float cos_o = dot(v.normal, wo), cos_i = dot(v.normal, wi);
    abs_cos_i = fabsf(cos_i);
    const bool same = cos_o * cos_i > 0;
    // the Lambert lobe's three values for every lane (two selects and a multiply), the Phong lobe's over them under ONE divergent region (round 5: two regions before)
    float p = same ? abs_cos_i * K_INV_PI : 0.f;
    scale = same ? K_INV_PI : 0.f;
    const bool phong = B.lobe == LOBE_PHONG;
    col = ld3(phong ? B.m->cs : B.m->c0);
    if (phong) {
        const float cos_alpha = dot(vertex_basis_c(v), wi);
        const float exponent = B.m->exponent;
        const float pe = phong_pow_lobe(cos_alpha, *B.m);
        const float p0 = exponent == 0.f ? 1.f : (exponent > 0.f ? 0.f : K_INF);
        scale = same ? pe * B.m->inv_eta : 0.f;
        p = (cos_alpha > 0.f ? pe : p0) * B.m->phong_pdf_norm;
    }
    const bool nondelta = !bsdf_is_delta(B);
    scale = nondelta ? scale : 0.f;
    pdf = nondelta ? p : 0.f;
}

// The direction half of sample_ for the two non-delta lobes, world space (their value and pdf are eval_ / pdf_ of that
// direction: 2253-2254, 2526-2527), so that a caller that rarely needs the value can defer it (estimate_by_bsdf).
// Both lobes place a point (rad cos, rad sin) on a circle and lift it: the cosine lobe by the concentric disk mapping (710-743), the
// Phong lobe by cos(theta) = u1^(1/(n+1)) around the mirror direction (2510-2524).  A wavefront holds vertices of both kinds: the angle,
// the radius and the height are computed per lobe, the two quarter-rate sin / cos and the combination with the basis are issued once.
// `back_dead` (bsdf_continue): set when a Phong sample taken from the BACK side of its surface (wo.z < 0: the flip of 2539 moves wi away from
// the lobe's axis) has pdf_ = 0 -- the reference clamps cos_alpha at 0 there (2549) and ends the path (4588); from the front side
// cos_alpha = z and the caller's u1 > 0 test decides.
KY_DEV f3 bsdf_sample_dir_nondelta(const Vertex& v, f3 wo, float u0, float u1, bool* back_dead = nullptr, bool flat_phong = false) {
    const bool phong = v.bsdf.lobe == LOBE_PHONG;
    const float cos_o = dot(v.normal, wo);
    float ang, rad, z = any_f();
    if (phong) {
        z = pow_nonneg(u1, v.bsdf.m->eta);                                 // 1 / (exponent + 1)
        rad = fsqrt(1.f - z * z);
        ang = u0;                                                          // phi = 2 pi u0, in revolutions
    } else {   // concentric_disk_sample, 710-733 (angles in revolutions: theta / 2 pi)
        const float rx = 2.f * u0 - 1, ry = 2.f * u1 - 1;
        const bool xmajor = fabsf(rx) > fabsf(ry);
        rad = xmajor ? rx : ry;
        const float ratio = mul_legacy(xmajor ? ry : rx, rcp(rad));        // rx = ry = 0: 0 x (1 / 0) = 0 (mul_legacy): the radius is 0 and the centre maps to the centre
        ang = xmajor ? 0.125f * ratio : 0.25f - 0.125f * ratio;            // (pi/4) q, pi/2 - (pi/4) q
    }
    float px = cos_rev(ang) * rad, py = sin_rev(ang) * rad;
    if (!phong) {   // cosine_hemisphere_sample 737-743, flipped into wo's hemisphere (2247-2249)
        z = __builtin_copysignf(fsqrt(fmaxf(0.f, 1 - px * px - py * py)), cos_o);   // `if (wo.z < 0) z = -z` as one v_bfi_b32 (no divergent region around a single negation)
    }
    const LobeBasis L = vertex_basis(v);
    f3 wi = L.a * px + L.b * py + L.c * z;
    // (`flat_phong`, KY_FEAT_FLAT_PHONG: every plastic surface is a rectangle, so wo is never below a Phong lobe's normal and this block is dead -- which matters because the
    // compiler turns it into a dot product, three multiply-adds and three selects that EVERY direction sample executes: ten instructions, twice per Cornell vertex, six times per Veach vertex)
    if (!flat_phong)
    if (phong && cos_o < 0) {   // `if (wo.z < 0) wi.z *= -1` (2539) in world space; rectangles face the ray (1289), so this is the inside of a plastic sphere
        const float nw = dot(v.normal, wi);
        wi = wi - (2.f * nw) * v.normal;
        if (back_dead) {
            const float cos_alpha = z - 2.f * nw * dot(v.normal, L.c);   // dot(wr, wi) after the flip
            const float pe = phong_pow(cos_alpha, v.bsdf.m->exponent, v.bsdf.m->exp_flags);
            *back_dead = v.bsdf.m->exponent > 0.f && !(cos_alpha > 0.f && pe > 0.f);
        }
    }
    return wi;
}

// The two delta lobes in one pass: perfect_specular_reflection_t (2292-2307) is the reflection branch of
// fresnel_specular_scattering_t (2355-2412) taken with probability 1, so the lanes of both run the same code.  Glass:
// fresnel_dielectric(wo.z, 1, eta) (1963-1996) and refract (1931-1957) in one pass -- both start from the same cos(theta_i),
// sin^2(theta_i) and index ratio, and the Fresnel term's cos(theta_t) is the refracted direction's (the reference computes it
// twice, as sqrt(1 - (r sin)^2) and as sqrt(1 - r^2 sin^2)).  reflect = 2 (n.wo) n - wo is (-wo.x, -wo.y, wo.z) of the shading frame.
struct DeltaSample {
    f3 wi;
    float percent;    // the branch's probability = its pdf: 1 (mirror), F or 1 - F (glass)
    float abs_cos;    // |wi.z| of the shading frame
    bool reflected;
};
KY_DEV DeltaSample bsdf_sample_delta(const Vertex& v, f3 wo, float u0) {
    const DMat& M = *v.bsdf.m;
    const bool glass = v.bsdf.lobe == LOBE_GLASS;
    const float cos_o = dot(v.normal, wo);
    const float eta_t = M.eta, inv_eta = M.inv_eta;                  // (both read before the selects: no load under a branch)
    const bool into = cos_o > 0;
    const float nz = into ? 1.f : -1.f;
    const float ratio = into ? inv_eta : eta_t;                      // eta_i / eta_t seen from wo's side
    const float ei = into ? 1.f : eta_t, et = into ? eta_t : 1.f;
    const float cos_theta_i = fminf(fabsf(cos_o), 1.f);
    const float sin_theta_i_sq = fmaxf(0.f, 1 - cos_theta_i * cos_theta_i);
    const float sin_theta_t = ratio * fsqrt(sin_theta_i_sq);
    const bool tir = sin_theta_t >= 1;
    const float cos_theta_t = fsqrt(fmaxf(0.f, 1 - sin_theta_t * sin_theta_t));
    const float r_para = ((et * cos_theta_i) - (ei * cos_theta_t)) * rcp((et * cos_theta_i) + (ei * cos_theta_t));
    const float r_perp = ((ei * cos_theta_i) - (et * cos_theta_t)) * rcp((ei * cos_theta_i) + (et * cos_theta_t));
    const float reflect_percent = (!glass || tir) ? 1.f : (r_para * r_para + r_perp * r_perp) * 0.5f;
    DeltaSample s;
    s.reflected = !glass || u0 < reflect_percent;   // (glass with u0 >= 1 cannot happen: total internal reflection always reflects)
    const float k = ratio * cos_theta_i - cos_theta_t;
    const float sw = s.reflected ? -1.f : -ratio, sn = s.reflected ? 2.f * cos_o : k * nz;
    s.wi = wo * sw + v.normal * sn;
    s.percent = s.reflected ? reflect_percent : 1 - reflect_percent;
    s.abs_cos = fabsf(s.reflected ? cos_o : -ratio * cos_o + k * nz);
    return s;
}

// bsdf sample_ x4 with the reference's own value and pdf (KAT entry points, vertex traces, the recursive integrators' roulette on
// the BSDF value, the BSDF-sampling estimators' slow path); the iterative integrator continues its path with bsdf_continue.
KY_DEV BsdfSample bsdf_sample(const Vertex& v, f3 wo, float u0, float u1) {
    BsdfSample s;
    if (bsdf_is_delta(v.bsdf)) {
        const DeltaSample d = bsdf_sample_delta(v, wo, u0);
        s.wi = d.wi;
        s.pdf = d.percent;
        s.f = (ld3(d.reflected ? v.bsdf.m->c0 : v.bsdf.m->c1) * d.percent) * rcp(d.abs_cos);   // 2301, 2384, 2402
        s.flags = (d.reflected ? BSDF_REFLECTION : BSDF_TRANSMISSION) | BSDF_SPECULAR;
    } else {
        float abs_cos_i;
        s.wi = bsdf_sample_dir_nondelta(v, wo, u0, u1);
        bsdf_eval_pdf(v, wo, s.wi, s.f, s.pdf, abs_cos_i);
        s.flags = BSDF_REFLECTION | (v.bsdf.lobe == LOBE_PHONG ? BSDF_GLOSSY : BSDF_DIFFUSE);
    }
    return s;
}

// What path_tracing_iteration_t does with its BSDF sample (4586-4597): the direction, whether the path goes on, and the factor
// beta is multiplied by, bs.f |dot(wi, n)| / bs.pdf -- written out per lobe, where most of it cancels:
//   Lambert  (albedo / pi) |cos| / (|cos| / pi) = albedo                     the path ends when wi leaves wo's hemisphere (f = 0, pdf = 0)
//   Phong    Ks' (n+2)/2pi cos_a^n |cos| / ((n+1)/2pi cos_a^n) = Ks' (n+2)/(n+1) |cos|   (DMat::c1); ends when wi leaves the hemisphere
//            (f = 0) or cos_a^n = 0: with cos_a = u1^(1/(n+1)) that is u1 = 0
//   mirror   (R / |cos|) |cos| / 1 = R;   glass  (R F / |cos|) |cos| / F = R,  (T (1 - F) / |cos|) |cos| / (1 - F) = T
// and a black factor ends the path like the reference's is_black(bs.f) (4588).  The quotient is exact here where the reference's
// rounds twice (1e-7 relative); its inf x 0 = NaN at exactly grazing mirror incidence (a local cosine of 0 against a world one that is
// not, 2301 / 4592: about one sample in 5e7) has no counterpart -- the two cosines are one number here.
struct BsdfContinue {
    f3 wi, weight;
    bool ok, specular;
};
KY_DEV BsdfContinue bsdf_continue(const Vertex& v, f3 wo, float u0, float u1, bool flat_phong = false) {
    BsdfContinue c;
    c.specular = bsdf_is_delta(v.bsdf);
    if (c.specular) {
        const DeltaSample d = bsdf_sample_delta(v, wo, u0);
        c.wi = d.wi;
        c.weight = ld3(d.reflected ? v.bsdf.m->c0 : v.bsdf.m->c1);
        c.ok = ((v.bsdf.m->exp_flags & (d.reflected ? 4 : 8)) != 0) & (d.percent != 0.f);   // !is_black(weight): a constant of the material, decided by the host (DMat::exp_flags)
    } else {
        bool back_dead = false;
        c.wi = bsdf_sample_dir_nondelta(v, wo, u0, u1, &back_dead, flat_phong);
        const float cos_o = dot(v.normal, wo), cos_i = dot(v.normal, c.wi);
        const bool phong = v.bsdf.lobe == LOBE_PHONG;
        const f3 col = ld3(phong ? v.bsdf.m->c1 : v.bsdf.m->c0);
        c.weight = col * (phong ? fabsf(cos_i) : 1.f);   // (one select, three products: col x 1 is col)
        const bool dead_u1 = phong & !(u1 > 0.f) & (v.bsdf.m->exponent > 0.f);   // (bitwise: every operand is at hand, no short-circuit branches)
        c.ok = (cos_o * cos_i > 0) & ((v.bsdf.m->exp_flags & (phong ? 8 : 4)) != 0) & !dead_u1 & !back_dead;
    }
    return c;
}

// ---------------------------------------------------------------------------------------------
// lights (ky.cpp:2764-3062) and the shape sampling they call (1028-1090, 1404-1513)
// ---------------------------------------------------------------------------------------------
struct LightSample {
    f3 position, wi, Li;
    float pdf;
    bool lit;   // area lights: the sample sees the light's emitting side, i.e. Li is the light's colour (a wave-uniform value) and not black by position
    // area lights: the unit vector towards the sample and the distance, as sample_Li made them on the way to `wi` (before `wi` is zeroed for a sample that does not
    // count).  The shadow ray of scene_t::occluded (3187-3201) is the same subtraction, the same squared length, the same reciprocal root: the estimators take them
    // from here -- written out a second time after the sampler's branches had joined, the compiler computed them a second time (3 sub, 3 fma, v_rsq_f32, 3 mul per light sample).
    f3 dir;
    float dist;
};

KY_DEV f3 uniform_sphere_sample(float u0, float u1) {  // 761-769
    const float z = 1 - 2 * u0;
    const float radius = fsqrt(fmaxf(0.f, 1.f - z * z));
    return mk3(radius * cos_rev(u1), radius * sin_rev(u1), z);
}

// shape_t::sample_position x4 (1144, 1225, 1307, 1404); normals are the stored (unit) ones
KY_DEV void shape_sample_position(const DLight& L, float u0, float u1, f3& position, f3& normal, int feat = 0) {
    if ((feat & KY_FEAT_RECT_LIGHTS) || (!(feat & KY_FEAT_SPHERE_LIGHTS) && L.shape_kind == KY_SHAPE_RECTANGLE)) {
        // p1, e0, e1 and the normal as four 16-byte scalar loads (each is three floats and a fourth word of the record: DLight is laid out in 16-byte groups);
        // read as twelve floats they were eight loads, x2 + x1 each
        const float4 P = *reinterpret_cast<const float4*>(L.p1), E0 = *reinterpret_cast<const float4*>(L.e0), E1 = *reinterpret_cast<const float4*>(L.e1),
                     N = *reinterpret_cast<const float4*>(L.n);
        position = mk3(P.x, P.y, P.z) + mk3(E0.x, E0.y, E0.z) * u0 + mk3(E1.x, E1.y, E1.z) * u1;
        normal = mk3(N.x, N.y, N.z);
    } else if ((feat & KY_FEAT_SPHERE_LIGHTS) || L.shape_kind == KY_SHAPE_SPHERE) {
        const f3 dir = uniform_sphere_sample(u0, u1);
        position = ld3(L.p1) + L.radius * dir;
        normal = dir;
    } else if (L.shape_kind == KY_SHAPE_TRIANGLE) {
        const float su0 = fsqrt(u0);
        const float bx = 1 - su0, by = u1 * su0;
        position = bx * ld3(L.p1) + by * ld3(L.e0) + (1 - bx - by) * ld3(L.e1);   // p0, p1, p2
        normal = ld3(L.n);
    } else {  // disk
        const Frame fr = make_frame(ld3(L.n));
        float px, py;
        concentric_disk(u0, u1, px, py);
        position = ld3(L.p1) + L.radius * (fr.s * px + fr.t * py);
        normal = ld3(L.n);
    }
}

// shape_t::sample_direction (1028-1051) and sphere_t::sample_direction (1419-1501)
// `ipdf` (a compile-time constant at every call; the callers that pass true: SceneRef::ipdf): `pdf` receives the RECIPROCAL of the density -- for the cone that is
// 2 pi (1 - cos theta_max) itself, and the estimators' weights 2 / (p_l + p_b) become 2 x / (1 + p_b x): one quarter-rate reciprocal per light sample instead of two.
// A density of zero is an infinite reciprocal.
KY_DEV void shape_sample_direction(const DLight& L, f3 p, f3 p_normal, float u0, float u1, f3& lposition, f3& lnormal, float& pdf, int feat = 0, bool ipdf = false) {
    const bool sphere = (feat & KY_FEAT_SPHERE_LIGHTS) ? true : ((feat & KY_FEAT_RECT_LIGHTS) == 0 && L.shape_kind == KY_SHAPE_SPHERE);
    const f3 c = ld3(L.p1);
    const float dc2 = length_sq(p - c);
    if (sphere && !(dc2 <= L.radius * L.radius)) {
        // outside the sphere: uniform cone sampling, 1458-1500
        const float inv_dist = rsq(dc2);
        const float sin_theta_max = L.radius * inv_dist;
        const float sin_theta_max_sq = sin_theta_max * sin_theta_max;
        const float inv_sin_theta_max = (dc2 * inv_dist) * L.e0[0];   // distance x 1 / radius (the host's, DLight::e0[0] of a sphere light): two multiplies for a quarter-rate reciprocal
        const float cos_theta_max = fsqrt(fmaxf(0.f, 1 - sin_theta_max_sq));
        float cos_theta = (cos_theta_max - 1) * u0 + 1;
        float sin_theta_sq = 1 - cos_theta * cos_theta;
        if (sin_theta_max_sq < 0.00068523f) {
            sin_theta_sq = sin_theta_max_sq * u0;
            cos_theta = fsqrt(1 - sin_theta_sq);
        }
        const float cos_alpha = sin_theta_sq * inv_sin_theta_max +
                                cos_theta * fsqrt(fmaxf(0.f, 1.f - sin_theta_sq * inv_sin_theta_max * inv_sin_theta_max));
        const float sin_alpha = fsqrt(fmaxf(0.f, 1.f - cos_alpha * cos_alpha));
        const Frame fr = make_frame((c - p) * inv_dist);
        const f3 world_normal = (sin_alpha * cos_rev(u1)) * (-fr.s) + (sin_alpha * sin_rev(u1)) * (-fr.t) + cos_alpha * (-fr.n);  // 431-439
        lposition = c + L.radius * world_normal;
        lnormal = world_normal;
        pdf = ipdf ? 2 * K_PI * (1 - cos_theta_max) : rcp(2 * K_PI * (1 - cos_theta_max));
        return;
    }
    shape_sample_position(L, u0, u1, lposition, lnormal, feat);
    const f3 wv = lposition - p;
    const float d2 = length_sq(wv);
    // Straight-line (round 5; two nested branches before): a coincident sample (d2 = 0: rsq gives inf, the direction and the quotient NaN) and an infinite quotient (the
    // direction lies in the light's plane) both mean pdf = 0 (1041-1049) -- one class test on the quotient instead of exec-mask bookkeeping around three instructions.
    const f3 wi = wv * rsq(d2);
    // inside-sphere case divides by the SHADE POINT's normal (quirk, 1436); the base class by the light's (1044)
    const f3 nn = sphere ? p_normal : lnormal;
    if (ipdf) {
        // the reciprocal density area |cos| / d^2 from the reciprocal root the direction was made with: no further quarter-rate instruction.  A coincident sample
        // (r = inf, the cosine NaN) and a direction in the light's plane (cosine 0) are not positive: "density zero"
        const float r = rsq(d2);
        const float x = (L.area * fabsf(dot(nn, wi))) * (r * r);
        pdf = x > 0.f ? x : K_INF;
        return;
    }
    const float q = L.inv_area * d2 * rcp(fabsf(dot(nn, wi)));
    // q is a product of non-negative factors: "d2 == 0 (q is NaN then: 0 x rcp(|NaN|)), infinite or NaN -> 0" is "keep it iff it is a positive finite number": one
    // v_cmp_class_f32 (+normal | +denormal) instead of three compares and two scalar ors
    pdf = __builtin_amdgcn_classf(q, 0x180) ? q : 0.f;
}

// shape_t::pdf_direction (1055-1090) and sphere_t::pdf_direction (1503-1513)
KY_DEV float shape_pdf_direction(const DLight& L, const DShapeFull* __restrict__ full, f3 p, f3 p_normal, f3 wi, bool general = true, int feat = 0, bool ipdf = false) {
    const bool sphere = (feat & KY_FEAT_SPHERE_LIGHTS) ? true : ((feat & KY_FEAT_RECT_LIGHTS) == 0 && L.shape_kind == KY_SHAPE_SPHERE);
    const f3 c = ld3(L.p1);
    const float dc2 = length_sq(p - c);
    if (sphere && !(dc2 <= L.radius * L.radius)) {
        const float sin_theta_max_sq = L.radius * L.radius * rcp(dc2);
        const float cos_theta_max = fsqrt(fmaxf(0.f, 1 - sin_theta_max_sq));
        const float x = 2 * K_PI * (1 - cos_theta_max);
        return ipdf ? x : rcp(x);  // uniform_cone_pdf, 798; never tests the hit (quirk 13)
    }

    // base class: re-intersect the light's OWN shape with isect.spawn_ray(wi)
    const f3 o = offset_ray_origin(p, p_normal, wi);
    float t;
    if (!surf_hit(L.isect, full, o, wi, K_INF, t, general)) return ipdf ? K_INF : 0.f;
    const f3 hp = o + t * wi;
    f3 ln = ld3(L.n);
    if (sphere) ln = normalize(hp - c);
    float pdf = length_sq(p - hp) * rcp(fabsf(dot(ln, wi)) * L.area);   // |dot| makes the ray-facing flip (1289) irrelevant
    if (isinf(pdf)) pdf = 0.f;
    return ipdf ? rcp(pdf) : pdf;
}

// environment_light_t's pdf (3032-3036, 3046-3052): 1 / (2 pi^2 sin(theta)), theta = acos(clamp(wi.z)):
// sin(acos(z)) = sqrt(1 - z^2)
KY_DEV float env_pdf(float wz) {
    const float z = fminf(fmaxf(wz, -1.f), 1.f);
    const float sin_theta = fsqrt(fmaxf(0.f, 1.f - z * z));
    return sin_theta == 0 ? 0.f : rcp(2 * K_PI * K_PI * sin_theta);
}

// light_t::sample_Li x4 (2825, 2891, 2964, 3026)
KY_DEV LightSample light_sample_Li(const DLight& L, f3 p, f3 p_normal, float u0, float u1, int feat = 0, bool ipdf = false) {
    LightSample s;   // every kind (wave-uniform) assigns every field
    s.lit = true;
    s.dir = any3();   // (read for area lights only)
    s.dist = any_f();
    const SceneRef K{nullptr, false, feat};   // the kind predicates only
    if (K.is_area(L.kind)) {
        f3 lposition, lnormal;
        shape_sample_direction(L, p, p_normal, u0, u1, lposition, lnormal, s.pdf, feat, ipdf);
        s.position = lposition;
        const f3 dv = lposition - p;
        const float d2 = length_sq(dv);
        // (a coincident sample of a planar shape has a NaN direction and therefore density 0 already: only the cone sampler's needs the second test)
        const bool ok = !((ipdf ? __builtin_isinf(s.pdf) : s.pdf == 0) || ((feat & KY_FEAT_RECT_LIGHTS) ? false : d2 == 0));
        const float inv_d = rsq(d2);
        const f3 wi = dv * inv_d;
        s.dir = wi;
        s.dist = d2 * inv_d;
        s.wi = mk3(ok ? wi.x : 0.f, ok ? wi.y : 0.f, ok ? wi.z : 0.f);
        // areal_radiance(light_isect, -wi) with the SAMPLED (stored) normal: one-sided (quirk 5), 2957-2960
        // (a light whose colour is black -- or not positive -- emits nothing: is_black(Li) ends the estimate in the reference, 3940 / 4045.  The test is on a
        // wave-uniform value, a scalar compare; without it such a sample would trace its shadow ray and add colour x 0 x k, NaN when k overflows.)
        // (bitwise: the three tests are at hand -- short-circuit evaluation put the colour's scalar loads and their wait inside two nested divergent regions)
        const bool lit = ok & (dot(lnormal, wi) < 0) & !is_black_bits(L.color);
        s.lit = lit;
        s.Li = mk3(lit ? L.color[0] : 0.f, lit ? L.color[1] : 0.f, lit ? L.color[2] : 0.f);
    } else if (K.is_delta(L.kind) && L.kind == KY_LIGHT_POINT) {
        const f3 lp = ld3(L.position);
        const f3 dv = lp - p;
        const float inv_d2 = rcp(length_sq(dv));
        s.position = lp;
        s.wi = dv * fsqrt(inv_d2);
        s.pdf = 1.f;
        s.Li = ld3(L.color) * inv_d2;
    } else if (K.is_delta(L.kind)) {   // directional
        s.wi = -ld3(L.direction);
        s.position = p + s.wi * (2 * L.world_radius);
        s.pdf = 1;
        s.Li = ld3(L.color);
    } else {  // environment: uniform sphere direction with the pdf of quirk 4, 3026-3041
        s.wi = uniform_sphere_sample(u0, u1);
        s.position = p + s.wi * (2 * L.world_radius);
        s.pdf = env_pdf(s.wi.z);
        s.Li = ld3(L.color);
    }
    return s;
}

// light_t::pdf_Li x4 (2855, 2903, 2984, 3043)
KY_DEV float light_pdf_Li(const DLight& L, const DShapeFull* __restrict__ full, f3 p, f3 p_normal, f3 wi, bool general = true, int feat = 0, bool ipdf = false) {
    const SceneRef K{nullptr, false, feat};
    if (K.is_area(L.kind)) return shape_pdf_direction(L, full, p, p_normal, wi, general, feat, ipdf);
    if (K.is_env(L.kind)) return env_pdf(wi.z);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// direct lighting (ky.cpp:3834-4088)
// ---------------------------------------------------------------------------------------------

// emission seen along a ray that hit `surface` (surface_t::intersect 3084 + areal_radiance 2957)
KY_DEV f3 surface_emission(const LdsScene& Lds, int surface, f3 normal, f3 wo) {
    const int al = Lds.hit[surface].area_light;
    f3 e = mk3(0, 0, 0);
    if (al >= 0 && dot(normal, wo) > 0) e = mk3(Lds.light_color[al][0], Lds.light_color[al][1], Lds.light_color[al][2]);
    return e;
}

// BSDF-sampling half of an estimator: by_bsdf (3889-3930, MIS=false) and by_bsdf_mis (3968-4033, MIS=true).
//
// MUST be called in wave-uniform control flow (`active` says whether this lane takes part): the reference asks "is the
// nearest hit along the sampled direction a surface that carries this light?" (3989-3995) with a full scene traversal
// per ray, but only the few rays that reach a carrier surface at all need an answer.  So each lane first tests just
// the carrier surfaces; the rare lanes that hit one then ask "is anything in front of it?", and those few queries
// are resolved SURFACE-parallel: the query ray is broadcast (v_readlane) and lane j tests surface j (records in LDS),
// one ballot per query -- however many queries the wave has (rounds 1-3 switched to a traversal by the querying lanes beyond four
// queries; without that second way out the hot kernel needs 72 VGPRs instead of 80, which is a seventh wavefront per SIMD: +0.8 % by
// itself, +4.3 % with the wavefront).  Scenes the fast path does not cover (general quads / triangles / disks, environment lights, many
// carriers, more than 64 surfaces) take the ordinary traversal.
// path_tracing_recursion_t's emitter look-up at a specular vertex (4341-4349) needs the nearest hit along one more ray -- for exactly the lanes
// that take no part in the vertex's direct lighting.  In that integrator's own instantiations the look-up RIDES ALONG with the first
// nearest-hit or shadow traversal the light loop runs for the other lanes: one traversal with fuller lanes instead of two sparse ones.
struct RideAlong {
    bool want;   // this lane has a ray that nobody has traced yet
    f3 o, d;
    float t;     // out: distance of the nearest hit
    int hs;      // out: its (sorted) surface index, -1 on a miss
};
struct ShadowQueue;
struct SqRay;
KY_DEV void sq_push_bsdf_query(SceneRef S, ShadowQueue& q, bool push, f3 o, f3 d, float tmax, f3 c, unsigned tag);
// `sq` (QUEUE instantiations): the occlusion queries of (b) are not resolved here but join the wave's stack of deferred shadow rays, with
// the estimate x beta x weight as the contribution; the function then returns black.
// The estimate is ADDED, times `w`, to `acc` by the lanes that have one, under their own predicate (in place: no default value travels through the
// nesting levels, no select afterwards): the render kernels pass the path's radiance sum and throughput x strategy weight, the KAT entry a zeroed
// value and w = 1.
template <bool MIS>
KY_DEV void estimate_by_bsdf(SceneRef S, const LdsScene& Lds, const Vertex& v, f3 wo, int li, float u0, float u1, bool active, f3& acc, f3 w,
                             ShadowQueue* sq = nullptr, f3 beta = f3{0, 0, 0}, float weight = 0.f, unsigned tag = 0, RideAlong* ra = nullptr) {
    const DLight& L = scene_light(S, li);
    if (S.is_delta(L.kind)) return;  // light.is_delta(), 3894 / 3977 (wave-uniform)
    BsdfSample bs;
    // what `live` guards: read at the end only for lanes whose sample counts (and, in the query loop, through __shfl from such lanes)
    f3 f_cos = any3(), o = any3(), Li = any3();
    bs.wi = any3();
    bs.f = any3();
    bs.pdf = any_f();
    bool live = false, lit = false;   // live: the sample's value and pdf count; lit: it sees light (Li is set and not black)
    float t_l = any_f();              // fast path: distance of the nearest carrier hit
    const bool fast = S.is_area(L.kind) && ((S.feat & KY_FEAT_CARRIERS) || (L.n_carriers >= 0 && S->n_gen == 0));  // wave-uniform
    if (fast) {
        // Only the DIRECTION is sampled up front; the BSDF value and pdf (a pow for the Phong lobe) are evaluated for the few
        // lanes whose ray reaches a carrier that emits towards it -- for all other lanes Li = 0 decides the estimate (3996-4003).
        // (Delta lobes never get here: sample_all_light runs for non-delta vertices only, 4571.)
        const bool act = active && !bsdf_is_delta(v.bsdf);
        if (act) {
            bs.wi = bsdf_sample_dir_nondelta(v, wo, u0, u1, nullptr, S.flat_phong());
            o = offset_ray_origin(v.position, v.normal, bs.wi);  // isect.spawn_ray, 665-668
        }
        // (a) nearest carrier surface along the ray, and what it emits towards the ray (3084, 2957-2960)
        t_l = K_INF;
        int c = -1;
        bool pending;
        // every carrier carries THIS light (surface_t::area_light == &light, 3994): what it emits is the light's radiance, on the side its
        // normal faces (areal_radiance, 2957-2960) -- a wave-uniform colour, no per-lane look-up
        Li = ld3(L.color);
        if (S.feat & KY_FEAT_OWN_CARRIER) {
            // the one carrier IS the sampled parallelogram: its traversal record sits in the light's own record.  A rectangle reports the normal that faces the ray
            // (1289), so it emits towards every ray that hits it (dot(n, wi) = 0 is no hit: the plane test divides by it).
            const DSurf& R = L.isect;
            float t;
            bool hit;
            if (S.no_par()) {
                // ... and that parallelogram is a rectangle in an axis plane (KY_FEAT_AXIS_ALIGNED: every planar surface is): the rectangle test with the ray's reciprocal
                // direction, 12 instructions for 26 (the axis is the light's, a wave-uniform value: one of three copies runs)
                const float4 q0 = make_float4(L.aar[0], L.aar[1], L.aar[2], L.aar[3]);
                const float rv = L.aar[4];
                const int ax = L.aar_axis;
                // (each copy ends in an asm statement of its own: without them the compiler merges the three bodies into one behind six to eighteen register moves that
                // permute its operands -- more than the test's own arithmetic)
                float u_, v_;
                if (ax == 0) {
                    t = (q0.x - o.x) * rcp(bs.wi.x); u_ = (o.y + t * bs.wi.y) - q0.y; v_ = (o.z + t * bs.wi.z) - q0.w;
                    asm volatile("; lamp in an x plane" : "+v"(t), "+v"(u_), "+v"(v_));
                } else if (ax == 1) {
                    t = (q0.x - o.y) * rcp(bs.wi.y); u_ = (o.z + t * bs.wi.z) - q0.y; v_ = (o.x + t * bs.wi.x) - q0.w;
                    asm volatile("; lamp in a y plane" : "+v"(t), "+v"(u_), "+v"(v_));
                } else {
                    t = (q0.x - o.z) * rcp(bs.wi.z); u_ = (o.x + t * bs.wi.x) - q0.y; v_ = (o.y + t * bs.wi.y) - q0.w;
                    asm volatile("; lamp in a z plane" : "+v"(t), "+v"(u_), "+v"(v_));
                }
                hit = (fabsf(u_) <= q0.z) & (fabsf(v_) <= rv) & (t > K_SHAPE_EPS);   // aar_hit with tmax = inf
            } else {
                hit = par_hit(make_float4(R.f[0], R.f[1], R.f[2], R.f[3]), make_float4(R.f[4], R.f[5], R.f[6], R.f[7]), make_float4(R.f[8], R.f[9], R.f[10], R.f[11]), o, bs.wi, K_INF, t);
            }
            const bool ok = act & hit;
            t_l = ok ? t : t_l;
            c = ok ? L.carrier[0] : c;
            pending = ok && !is_black_bits(L.color);
        } else {
        for (int k = 0; k < L.n_carriers; ++k) {
            float t;
            bool hit;
            if (S.sphere_lights() && k == 0)   // the first carrier's sphere from the light's own record (DLight::aar): one load, not index -> table
                hit = sph_hit(make_float4(L.aar[0], L.aar[1], L.aar[2], L.aar[3]), o, bs.wi, t_l, t, true);
            else
                hit = surf_hit(scene_surf(S, L.carrier[k]), S->full, o, bs.wi, t_l, t, S.general, S.sphere_lights());
            const bool ok = act & hit;
            t_l = ok ? t : t_l;
            c = ok ? L.carrier[k] : c;
        }
        pending = c >= 0 && !is_black(Li);
        if (pending) {
            const f3 hp = o + t_l * bs.wi;
            pending = dot(hit_normal(Lds.hit[c], hp, bs.wi), bs.wi) < 0;
        }
        }
        if (pending) {  // rare: now the sample's value and pdf (3979-3987)
            float abs_cos_i;
            bsdf_eval_pdf(v, wo, bs.wi, bs.f, bs.pdf, abs_cos_i);
            f_cos = bs.f * abs_cos_i;
            live = !(is_black(f_cos) || (MIS ? (bs.pdf <= 0) : (bs.pdf == 0)));
            pending = live;
        }
        if (sq) {   // (b) deferred: a ray that ends just short of the carrier (its own hit, a few ulp around t_l, stays out of the interval)
            f3 c = any3();
            if (pending) {
                f3 Lq = (f_cos * Li) * rcp(bs.pdf);  // 3924
                if (MIS) {
                    float light_pdf;
                    if (L.pdf_from_carrier) {   // (wave-uniform) pdf_direction from the carrier hit itself, see below
                        const f3 hp = o + t_l * bs.wi;
                        light_pdf = length_sq(v.position - hp) * rcp(fabsf(dot(ld3(L.n), bs.wi)) * L.area);
                        if (isinf(light_pdf)) light_pdf = 0.f;
                    } else {
                        light_pdf = light_pdf_Li(L, S->full, v.position, v.normal, bs.wi, S.general, S.feat);
                    }
                    Lq = light_pdf > 0 ? (f_cos * Li) * (2.f * rcp(bs.pdf + light_pdf)) : mk3(0, 0, 0);  // 4028
                }
                c = (Lq * beta) * weight;
                pending = !(c.x == 0.f && c.y == 0.f && c.z == 0.f);
            }
            sq_push_bsdf_query(S, *sq, pending, o, bs.wi, t_l * (1.f - 1e-6f), c, tag);
            return;
        }
        // (b) is any surface in front of the carrier?  (the carrier itself reproduces t_l exactly, and t < t_l is strict)
        unsigned long long queries = __ballot(pending);
        bool blocked = false;
        if (S.large && S->n_surfaces > 64) {   // (lane j tests surface j: scenes of up to 64 surfaces)
            KY_PROBE(2);
            // the traversal may test the carrier with another formulation than (a) did (aar_hit vs par_hit): keep its own
            // hit, a few ulp around t_l, out of the interval
            if (pending) blocked = trace_any(S, S->occ, o, bs.wi, t_l * (1.f - 1e-6f));   // from a surface point to a point of a carrier surface
        } else {
            const int lane = (int)__lane_id();
            while (queries) {
                const int src = __ffsll((long long)queries) - 1;
                queries &= queries - 1;
                const f3 qo = mk3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
                const f3 qd = mk3(__shfl(bs.wi.x, src), __shfl(bs.wi.y, src), __shfl(bs.wi.z, src));
                const float qt = __shfl(t_l, src);
                // this lane's surface record is read INSIDE the loop (the empty asm makes the index opaque, so the loads
                // cannot be hoisted): queries are rare and the record would otherwise pin 14 VGPRs across the estimator
                int idx = lane < S->n_surfaces ? lane : 0;
                asm volatile("" : "+v"(idx));
                const DSurf& mine = S->all[idx];   // per-lane record from L2: queries are rare, LDS is worth more as wave capacity
                float t;
                const bool ok = (lane < S->n_surfaces) && surf_hit(mine, S->full, qo, qd, qt, t, S.general);
                const bool any = __any(ok);
                if (lane == src) blocked = any;
            }
        }
        lit = pending && !blocked;   // (pending implies live here)
    } else {
        if (active) {
            bs = bsdf_sample(v, wo, u0, u1);
            f_cos = bs.f * fabsf(dot(bs.wi, v.normal));
            live = !(is_black(f_cos) || (MIS ? (bs.pdf <= 0) : (bs.pdf == 0)));
            o = offset_ray_origin(v.position, v.normal, bs.wi);  // isect.spawn_ray, 665-668
        }
        float t = K_INF;
        KY_PROBE(2);
        int hs = -1;
        if (ra) {   // (no lane is both live and riding: riders hold specular vertices, which sample no light)
            const bool ride = ra->want;
            f3 to = o, td = bs.wi;
            if (ride) { to = ra->o; td = ra->d; }
            const int h = (live || ride) ? trace_nearest(S, to, td, t) : -1;
            if (ride) { ra->hs = h; ra->t = t; ra->want = false; }
            else hs = h;
        } else {
            hs = live ? trace_nearest(S, o, bs.wi, t) : -1;
        }
        Li = mk3(0, 0, 0);
        if (hs >= 0) {
            if (Lds.hit[hs].area_light == li) {  // 3912 / 3994
                const f3 hp = o + t * bs.wi;
                Li = surface_emission(Lds, hs, hit_normal(Lds.hit[hs], hp, bs.wi), -bs.wi);
            }
        } else if (live && S.is_env(L.kind)) {
            Li = ld3(L.color);  // light.environmental_radiance(ray), 3918 / 4000
        }
        lit = live && !is_black(Li);
    }
    if (lit) {
        if (MIS) {
            float light_pdf;
            if (fast && L.pdf_from_carrier) {   // (wave-uniform) shape_t::pdf_direction, 1055-1090, from the hit the carrier test found
                const f3 hp = o + t_l * bs.wi;
                if (S.ipdf()) {   // its reciprocal (an infinite density -- cosine zero -- counts as zero, 1086-1088: both are "not positive" here)
                    light_pdf = (fabsf(dot(ld3(L.n), bs.wi)) * L.area) * rcp(length_sq(v.position - hp));
                    if (!(light_pdf > 0.f)) light_pdf = K_INF;
                } else {
                    light_pdf = length_sq(v.position - hp) * rcp(fabsf(dot(ld3(L.n), bs.wi)) * L.area);
                    if (isinf(light_pdf)) light_pdf = 0.f;
                }
            } else {
                light_pdf = light_pdf_Li(L, S->full, v.position, v.normal, bs.wi, S.general, S.feat, S.ipdf());
            }
            if (S.ipdf()) {   // light_pdf is the density's reciprocal x: 2 / (p_b + 1 / x) = 2 x / (p_b x + 1); a density of zero is x = inf
                if (light_pdf < K_INF) acc = acc + w * ((f_cos * Li) * ((2.f * light_pdf) * rcp(bs.pdf * light_pdf + 1.f)));
            } else
            if (light_pdf > 0) acc = acc + w * ((f_cos * Li) * (2.f * rcp(bs.pdf + light_pdf)));  // 4028
        } else {
            acc = acc + w * ((f_cos * Li) * rcp(bs.pdf));  // 3924
        }
    }
}

// ---------------------------------------------------------------------------------------------
// deferred shadow rays (the lane engine's QUEUE instantiation, ky_launch.hip)
//
// scene_t::occluded (3187-3201) is the most expensive step of the light-sampling estimators and the one with the fewest
// lanes that need it: many light samples are dead before it (back side of the light, zero BSDF value -- a Phong lobe away
// from its peak) or die on the sampled light's own shape (quirk 1).  Instead of tracing each light's shadow rays at once with
// whatever lanes have one, a lane that has a live sample evaluates everything else first (BSDF value, MIS weight) and hands
// {ray, contribution, destination pixel} to sq_push.  Rays wait on a small per-wavefront stack in global memory until a push
// brings the total to 64; that push does not store its rays at all: as many of the new rays as are needed stay in their lanes'
// registers, the other lanes pop the stack, and the wave traces 64 rays with every lane busy, then adds the unoccluded
// contributions to their pixels' sums.  The stack is private to the wave (no synchronisation), an array of 48-byte entries so
// that a push of k rays writes one contiguous run of 48 k bytes (stores leave the L2 on this chip whatever one does: full lines cost
// what they carry, partial ones a whole line each), and at most 127 entries deep (6 KB per wave: the stacks of an XCD's 768
// wavefronts stay in its 4 MB L2, so pops are L2 hits).  Contributions are added in 32.32 fixed point, so the image does not
// depend on when a ray is resolved.
// ---------------------------------------------------------------------------------------------
#ifndef KY_SQ_FLUSH_AT
#define KY_SQ_FLUSH_AT 64   // rays at hand (waiting + new) that trigger a traversal: 64 fills every lane; less keeps the stack shallower and the lanes emptier (measurements)
#endif
constexpr int KY_SQ_ENTRY = 3;                                // float4 per ray: (origin, tmax) (direction, tag) (contribution, -)
#ifndef KY_SQ_CAP_ENTRIES
#define KY_SQ_CAP_ENTRIES 64
#endif
constexpr int KY_SQ_CAP = KY_SQ_CAP_ENTRIES;                  // entries per wavefront's block: < 64 waiting (round 5: a flush stores nothing; rounds 3-4: + < 64 of a push that did not fit the wave being traced)
struct ShadowQueue {
    float4* base;   // this wave's block: KY_SQ_CAP entries
    int n;          // rays on the stack (wave-uniform), < 64 between calls
    // where contributions go
    const int* c_pix;               // LDS: the pixel every lane of the workgroup is working on (-1: none)
    unsigned long long* c_def;      // LDS: [3][256] fixed-point sums of resolved contributions, per lane
    unsigned long long* accum;      // global fixed-point accumulators of the shard
    unsigned* flags;                // global NaN / inf flags of the shard
};
struct SqRay {
    f3 o, d, c;
    float tmax;
    unsigned tag;   // destination pixel << 6 | lane that pushed it
};
#define KY_SQ_SLOT(q, i) ((q).base + (i) * KY_SQ_ENTRY)
KY_DEV void sq_store(float4* e, const SqRay& r) {
    e[0] = make_float4(r.o.x, r.o.y, r.o.z, r.tmax);
    e[1] = make_float4(r.d.x, r.d.y, r.d.z, __uint_as_float(r.tag));
    e[2] = make_float4(r.c.x, r.c.y, r.c.z, 0.f);
}
KY_DEV SqRay sq_load(const float4* e) {
    const float4 a = e[0], b = e[1], c = e[2];
    return SqRay{mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), a.w, __float_as_uint(b.w)};
}
// traces this lane's ray (any hit occludes) and adds an unoccluded contribution to its pixel
KY_DEV void sq_trace(SceneRef S, const ShadowQueue& q, const SqRay& r) {
    if (trace_any(S, S->occ_deferred_ok ? S->occ : S->trav, r.o, r.d, r.tmax)) return;   // rays of all lights share the stack
    const float c[3] = {r.c.x, r.c.y, r.c.z};
    const int pix = (int)(r.tag >> 6), owner = (int)((threadIdx.x & ~63u) | (r.tag & 63u));
    const bool local = q.c_pix[owner] == pix;   // the lane that pushed the ray is still on that pixel: its LDS sum takes it
    unsigned fl = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const unsigned long long fx = film_fixed(c[ch], ch, fl);   // NaN / +-inf: flag bits, nothing added
        if (fx != 0) {
            if (local) atomicAdd(&q.c_def[ch * 256 + owner], fx);
            else atomicAdd(&q.accum[(size_t)pix * 3 + ch], fx);
        }
    }
    if (fl) atomicOr(&q.flags[pix], fl);
}
// wave-uniform call: lanes with `push` hand over the ray r
KY_DEV void sq_push(SceneRef S, ShadowQueue& q, bool push, SqRay r) {
    const unsigned long long m = __ballot(push);
    if (!m) return;
    const int k = __popcll(m);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));   // pushing lanes below this one
    if (q.n + k < KY_SQ_FLUSH_AT) {   // not yet a wavefront's worth: the rays wait
        if (push) sq_store(KY_SQ_SLOT(q, q.n + rank), r);
        q.n += k;
        return;
    }
    // 64 or more rays at hand: EVERY new ray is traced from the registers it is in, and the 64 - k lanes without one pop the top of the stack (q.n >= 64 - k).
    // Round 5: rounds 3-4 kept only the first 64 - q.n new rays in registers and stored the others on top of the stack -- from where they were popped again a
    // moment later by the very lanes that had stored them: k - (64 - q.n) entries written and read for nothing per flush, and a stack that grew to 2 q.n + k - 65
    // entries for a moment (which is why its block held 128).  Now nothing is stored at a flush, the stack never holds more than 63 entries, and q.n + k - 64 of
    // the OLD entries simply stay where they are.
    asm volatile("" ::: "memory");   // the compiler must keep the order of earlier pushes' stores and these loads; the hardware keeps a wavefront's accesses to one address in order
    const unsigned long long idle = ~m;
    const int idx = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));   // lanes without a new ray below this one
    const bool pop = !push && idx < q.n;   // (KY_SQ_FLUSH_AT < 64: there may be fewer old rays than idle lanes)
    if (pop) r = sq_load(KY_SQ_SLOT(q, q.n - 1 - idx));
    q.n -= min(q.n, 64 - k);
    if (KY_SQ_FLUSH_AT == 64 || push || pop) sq_trace(S, q, r);
}
KY_DEV void sq_push_bsdf_query(SceneRef S, ShadowQueue& q, bool push, f3 o, f3 d, float tmax, f3 c, unsigned tag) {
    sq_push(S, q, push, SqRay{o, d, c, tmax, tag});
}
// end of the kernel: what is left on the stack (wave-uniform call)
KY_DEV void sq_drain(SceneRef S, ShadowQueue& q) {
    const int lane = (int)__lane_id();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (lane < q.n) sq_trace(S, q, sq_load(KY_SQ_SLOT(q, lane)));
    q.n = 0;
}

// light-sampling half with the occlusion test deferred: by_emitter (3933-3962, MIS = false) / by_emitter_mis (4035-4074) in the order sample -> BSDF value ->
// weight -> [sampled shape's own hit] -> push; the reference's order (occlusion before the BSDF value) gives the same sum
// because every factor is computed from the same inputs and a zero factor zeroes the term either way.
// Wave-uniform call.  beta x weight = throughput x strategy weight x 1 / spp: what multiplies this estimate in the pixel's sum.
template <bool MIS>
KY_DEV void estimate_by_emitter_deferred(SceneRef S, const Vertex& v, f3 wo, int li, float u0, float u1, bool active, f3 beta, float weight,
                                         unsigned tag, ShadowQueue& q) {
    const DLight& L = scene_light(S, li);
    bool push = false;
    SqRay r{any3(), any3(), any3(), any_f(), tag};   // read by sq_push only for lanes that push
    if (active) {
        // Straight-line from here: every lane runs every instruction and `push` says whether its values mean anything.  (Skipping the
        // work of a dead sample would need ALL lanes of the wave dead; the nested version paid for its structure with defaults and
        // exec-mask bookkeeping at every level instead.)
        const bool ip = S.ipdf();   // (compile-time) ls.pdf is the density's reciprocal: shape_sample_direction
        const LightSample ls = light_sample_Li(L, v.position, v.normal, u0, u1, S.feat, ip);
        // scene_t::occluded(isect, ls.position), 3187-3201: the ray
        if (S.is_area(L.kind)) {   // (compile-time in the kernels that defer: every light an area light)
            r.d = ls.dir;
            r.tmax = ls.dist - 2e-3f;
        } else {
            const f3 to = ls.position - v.position;
            const float d2 = length_sq(to);
            const float inv_d = rsq(d2);
            r.d = to * inv_d;
            r.tmax = d2 * inv_d - 2e-3f;
        }
        r.o = offset_ray_origin(v.position, v.normal, r.d);
        f3 col;
        float scale, bsdf_pdf, abs_cos_i;
        // an area light's ls.wi IS r.d for a sample that counts (the same expression of the same operands, 1075-1078): its cosine is the one the origin was offset by
        bsdf_eval_parts(v, wo, S.is_area(L.kind) ? r.d : ls.wi, col, scale, bsdf_pdf, abs_cos_i);
        const bool delta_light = S.is_delta(L.kind);
        // f |cos| Li / pdf (3956) or 2 f |cos| Li / (p_l + p_b) (4057 / 4070), times throughput and strategy weight: the scalar factors first, the three channels once;
        // an area light's Li is its colour where the sample is lit (a wave-uniform value: scalar operands) and the predicate below says whether it is
        const float fc = scale * abs_cos_i;   // f |cos| = col x fc
        const float k = fc * (ip ? ((!MIS || delta_light) ? ls.pdf : (2.f * ls.pdf) * rcp(1.f + bsdf_pdf * ls.pdf))
                                 : ((!MIS || delta_light) ? rcp(ls.pdf) : 2.f * rcp(ls.pdf + bsdf_pdf))) * weight;
        const bool area = S.is_area(L.kind);
        const f3 Li = area ? ld3(L.color) : ls.Li;
        r.c = ((col * Li) * beta) * k;
        // !is_black(f |cos|) (3952 / 4052) without the product: a positive factor (a Phong lobe's odd power of a negative cosine is negative: black) and a material that is
        // not black; a NaN factor counts.  (A light whose colour is not finite must not meet a zero here: 0 x inf.)
        // (with reciprocal densities every light is an area light, whose `lit` already says that the density is not zero)
        push = (area ? ls.lit : !is_black(ls.Li)) && (ip || !(MIS ? ls.pdf <= 0 : ls.pdf == 0)) && !(fc <= 0.f) && !is_black(col);
        // the sampled shape itself is the likeliest occluder (quirk 1): one test here saves the ray a full traversal
        if (S.is_area(L.kind) && L.sampled_is_surface) {   // wave-uniform
            float t;
            if (surf_hit(L.isect, S->full, r.o, r.d, r.tmax, t, S.general, S.sphere_lights(), false)) push = false;   // (every one of these rays is aimed at the sphere: not sparse)
        }
    }
    sq_push(S, q, push, r);
}

// light-sampling half: by_emitter (3933-3962, MIS=false) and by_emitter_mis (4035-4074, MIS=true)
// scene_t::occluded for a shadow ray towards a sample of light li (wave-uniform), through the occluder tables that apply to that light
KY_DEV bool light_sample_occluded(SceneRef S, int li, f3 o, f3 dir, float tmax) {
    const unsigned tab = (unsigned)scene_light(S, li).shadow_table;   // which table, decided by the host (DLight::shadow_table)
    const bool two = (tab & 1u) != 0;
    bool occ = trace_any(S, scene_at<DTrav>(S, tab & ~1u), o, dir, tmax);
    if (two) {   // what is mounted behind the lamp: only a ray with an end in that half-space can meet it (DScene::occ_behind)
        const float4 plane = scene_at<float4>(S, opaque_off((unsigned)__builtin_offsetof(DScene, ts_plane)));
        const f3 pn = mk3(plane.x, plane.y, plane.z);
        const f3 e = o + dir * tmax;
        const bool need = !occ && (fminf(dot(pn, e), dot(pn, o)) <= plane.w);
        if (__any(need)) {
            if (need) occ = trace_any_planar(S, S->occ_behind, o, dir, tmax);
        }
    }
    return occ;
}

template <bool MIS>
KY_DEV void estimate_by_emitter(SceneRef S, const LdsScene& Lds, const Vertex& v, f3 wo, int li, float u0, float u1, f3& acc, f3 w) {
    const DLight& L = scene_light(S, li);
    KY_PROBE(3);
    const bool ip = S.ipdf();   // (compile-time) ls.pdf is the density's reciprocal: shape_sample_direction
    const LightSample ls = light_sample_Li(L, v.position, v.normal, u0, u1, S.feat, ip);
    const bool area = S.is_area(L.kind);   // (wave-uniform)
    const bool dead = (area ? !ls.lit : is_black(ls.Li)) || (!(ip || area) && (MIS ? (ls.pdf <= 0) : (ls.pdf == 0)));   // (an area light's `lit` says that the density is not zero -- and it is never negative)
    KY_CLK(5);
    if (!dead) {
        // scene_t::occluded(isect, ls.position), 3187-3201 (an area light's direction and distance: LightSample::dir)
        f3 dir;
        float dist;
        if (area) {
            dir = ls.dir;
            dist = ls.dist;
        } else {
            const f3 to = ls.position - v.position;
            const float d2 = length_sq(to);
            const float inv_d = rsq(d2);
            dir = to * inv_d;
            dist = d2 * inv_d;
        }
        const f3 o = offset_ray_origin(v.position, v.normal, dir);
        KY_PROBE(4);
        const bool occ = light_sample_occluded(S, li, o, dir, dist - 2e-3f);
        KY_CLK(6);
        if (!occ) {
            KY_PROBE(5);
            f3 col;
            float scale, bsdf_pdf, abs_cos_i;
            // an area light's ls.wi IS dir for a live sample (the same expression of the same operands, 1075-1078): its cosine is the one the ray's origin was offset by
            bsdf_eval_parts(v, wo, area ? dir : ls.wi, col, scale, bsdf_pdf, abs_cos_i);
            const float fc = scale * abs_cos_i;   // f |cos| = col x fc
            if (!(fc <= 0.f) && !is_black(col)) {    // !is_black(f |cos|), 3952 / 4052: some channel positive (colours are not negative; a NaN factor counts)
                const bool delta_light = S.is_delta(L.kind);
                // 3956 / 4057 / 4070, the scalar factors first; an area light's Li is its colour where the sample is lit (wave-uniform: scalar operands)
                const float k = fc * (ip ? ((!MIS || delta_light) ? ls.pdf : (2.f * ls.pdf) * rcp(1.f + bsdf_pdf * ls.pdf))
                                         : ((!MIS || delta_light) ? rcp(ls.pdf) : 2.f * rcp(ls.pdf + bsdf_pdf)));
                acc = acc + ((col * (area ? ld3(L.color) : ls.Li)) * w) * k;
            }
            KY_CLK(7);
        }
    }
}

// the same estimator as a WAVE-UNIFORM call whose shadow traversal also serves the lanes of `ra` (RideAlong) that still wait: when there are any,
// shadow rays and look-up rays go through one nearest-hit scan of the whole scene (a shadow ray is occluded iff that scan finds a hit
// inside its interval: the occluder tables are subsets that decide the same, tests/test_occluders.py).
template <bool MIS>
KY_DEV void estimate_by_emitter_ride(SceneRef S, const LdsScene& Lds, const Vertex& v, f3 wo, int li, float u0, float u1, bool active, RideAlong& ra, f3& acc, f3 w) {
    const DLight& L = scene_light(S, li);
    LightSample ls{any3(), any3(), any3(), any_f(), false, any3(), any_f()};
    const bool area = S.is_area(L.kind);   // (wave-uniform)
    bool dead = true;
    if (active) {
        ls = light_sample_Li(L, v.position, v.normal, u0, u1, S.feat);
        dead = (area ? !ls.lit : is_black(ls.Li)) || (MIS ? (ls.pdf <= 0) : (ls.pdf == 0));
    }
    f3 o = any3(), dir = any3();
    float tmax = any_f();
    if (!dead) {  // scene_t::occluded(isect, ls.position), 3187-3201
        if (area) {
            dir = ls.dir;
            tmax = ls.dist - 2e-3f;
        } else {
            const f3 to = ls.position - v.position;
            const float d2 = length_sq(to);
            const float inv_d = rsq(d2);
            dir = to * inv_d;
            tmax = d2 * inv_d - 2e-3f;
        }
        o = offset_ray_origin(v.position, v.normal, dir);
    }
    bool occ = true;
    const bool ride = ra.want;
    if (__any(ride)) {
        if (ride) { o = ra.o; dir = ra.d; tmax = K_INF; }
        int hs = -1;
        if (ride || !dead) hs = trace_nearest(S, o, dir, tmax);
        occ = hs >= 0;
        if (ride) { ra.hs = hs; ra.t = tmax; ra.want = false; }
    } else if (!dead) {
        occ = light_sample_occluded(S, li, o, dir, tmax);
    }
    if (!dead && !occ) {
        f3 col;
        float scale, bsdf_pdf, abs_cos_i;
        bsdf_eval_parts(v, wo, area ? dir : ls.wi, col, scale, bsdf_pdf, abs_cos_i);   // (an area light's ls.wi IS dir for a live sample; no lane is both live and riding)
        const float fc = scale * abs_cos_i;
        if (!(fc <= 0.f) && !is_black(col)) {    // !is_black(f |cos|), 3952 / 4052: some channel positive (colours are not negative; a NaN factor counts)
            const bool delta_light = S.is_delta(L.kind);
            const float k = fc * ((!MIS || delta_light) ? rcp(ls.pdf) : 2.f * rcp(ls.pdf + bsdf_pdf));   // 3956 / 4057 / 4070: scalar factors first (estimate_by_emitter)
            acc = acc + ((col * (area ? ld3(L.color) : ls.Li)) * w) * k;
        }
    }
}

// estimate_direct_lighting_both_mis (4076-4088) for a scene whose ONLY light is its environment (KY_FEAT_SINGLE_ENV): the two halves above in one pass.  Both halves' rays ask
// the same question of the same surfaces -- the BSDF-sampled ray counts iff it leaves the scene (no surface carries a light: 3994 never holds, 4000 decides), the
// light-sampled ray is unoccluded iff it does (its sample lies a scene diameter away, 3039) -- so both go through ONE any-hit scan (trace_any_pair).
// Called by the lanes that hold a non-delta vertex; `acc_b` / `acc_l` receive w x the halves (the same sum unless a trace keeps them apart).
KY_DEV void estimate_env_both(SceneRef S, const Vertex& v, f3 wo, float ub0, float ub1, float ul0, float ul1, f3& acc_b, f3& acc_l, f3 w) {
    const DLight& L = scene_light(S, 0);
    const bool dark = is_black_bits(L.color);   // (wave-uniform)
    // BSDF-sampling half up to its ray (3979-3990): only the direction now -- the sample's value and pdf (a pow for the Phong lobe) are evaluated for the rays that
    // leave the scene (a sample whose value is black traces its ray for nothing, and adds nothing)
    AnyRay A;
    A.d = bsdf_sample_dir_nondelta(v, wo, ub0, ub1, nullptr, S.flat_phong());
    A.o = offset_ray_origin(v.position, v.normal, A.d);
    A.tmax = K_INF;
    const bool live_b = !dark;
    // light-sampling half up to its ray (4041-4050): a uniform direction with quirk 4's density (3026-3041); the ray towards p + wi 2R
    // (uniform_sphere_sample's radius IS the sine of the density: sqrt(1 - z^2) of the same z, which lies in [-1, 1] without the clamp)
    const float z = 1 - 2 * ul0;
    const float sin_theta = fsqrt(fmaxf(0.f, 1.f - z * z));
    const f3 wl = mk3(sin_theta * cos_rev(ul1), sin_theta * sin_rev(ul1), z);
    const float pl = sin_theta == 0 ? 0.f : rcp(2 * K_PI * K_PI * sin_theta);
    const bool live_l = !dark && !(pl <= 0);
    AnyRay B{offset_ray_origin(v.position, v.normal, wl), wl, 2 * L.world_radius - 2e-3f};
    bool occ_b = true, occ_l = true;
    if (live_b | live_l) trace_any_pair(S, A, B, occ_b, occ_l);
    const f3 Li = ld3(L.color);
    if (live_b && !occ_b) {
        f3 f;
        float pdf, abs_cos_i;
        bsdf_eval_pdf(v, wo, A.d, f, pdf, abs_cos_i);   // the sample's own value and pdf (2253-2254, 2526-2527)
        const f3 f_cos = f * abs_cos_i;
        const float light_pdf = env_pdf(A.d.z);
        if (!(is_black(f_cos) || pdf <= 0) && light_pdf > 0) acc_b = acc_b + w * ((f_cos * Li) * (2.f * rcp(pdf + light_pdf)));   // 3987, 4014, 4028
    }
    if (live_l && !occ_l) {
        f3 col;
        float scale, bsdf_pdf, abs_cos_i;
        bsdf_eval_parts(v, wo, wl, col, scale, bsdf_pdf, abs_cos_i);
        const float fc = scale * abs_cos_i;
        if (!(fc <= 0.f) && !is_black(col)) acc_l = acc_l + ((col * Li) * w) * (fc * (2.f * rcp(pl + bsdf_pdf)));   // 4052, 4070
    }
}

// sample_all_light, 3834-3872.  Wave-uniform call; `active` lanes draw 4 numbers per light (+2 for the plain bsdf
// strategy, 3900) and ADD beta x (the estimators' sum) to Lo -- each estimator adds its own term in place (estimate_by_bsdf), so the 0.5 of
// both_mis (4083) is part of the weight the terms are multiplied by, not a pass over their sum.
// `decisions` (KAT tracing only; a null constant everywhere else, which removes the code): bit li = the BSDF half of light li's
// estimate was non-black, bit 16 + li = its light half.
// `sq` (the QUEUE instantiations of the lane engine: strategies both_mis, light_mis, light): the light-sampling halves are not added but
// pushed on the wave's shadow-ray stack with beta x weight (x 0.5 under both_mis) as their weight in the pixel's sum.
template <bool DEBUG_SAMPLER>
KY_DEV void sample_all_light(SceneRef S, const LdsScene& Lds, const Vertex& v, f3 wo, Sampler& smp, int strategy, bool active, f3& Lo, f3 beta,
                             unsigned* decisions = nullptr, ShadowQueue* sq = nullptr, float weight = 0.f, unsigned tag = 0, RideAlong* ra = nullptr) {
    const int nl = S.single_light() ? 1 : S->n_lights;
    const f3 w = strategy == KY_DIRECT_BOTH_MIS ? beta * 0.5f : beta;
    for (int li = 0; li < nl; ++li) {
        // the reference's GCC build draws random_bsdf first, then random_light (3866-3868), for every light and strategy
        float ub0 = any_f(), ub1 = any_f(), ul0 = any_f(), ul1 = any_f();   // drawn, and read, by the active lanes only
        if (active) { ub0 = sampler_next<DEBUG_SAMPLER>(smp); ub1 = sampler_next<DEBUG_SAMPLER>(smp); }
        f3 Lb = mk3(0, 0, 0), Ll = mk3(0, 0, 0);      // tracing only: the two halves on their own
        f3& ab = decisions ? Lb : Lo;
        f3& al = decisions ? Ll : Lo;
        const f3 wt = decisions ? mk3(1, 1, 1) : w;
        if (strategy == KY_DIRECT_BOTH_MIS && (S.feat & KY_FEAT_SINGLE_ENV) && !sq && !ra) {   // (compile-time) one environment light: both halves in one pass
            if (active) {
                ul0 = sampler_next<DEBUG_SAMPLER>(smp); ul1 = sampler_next<DEBUG_SAMPLER>(smp);
                estimate_env_both(S, v, wo, ub0, ub1, ul0, ul1, ab, al, wt);
            }
        } else if (strategy == KY_DIRECT_BOTH_MIS) {  // 4076-4088
            KY_CLK(3);
            estimate_by_bsdf<true>(S, Lds, v, wo, li, ub0, ub1, active, ab, wt, sq, beta, weight * 0.5f, tag, ra);   // draws nothing itself
            KY_CLK(4);
            // random_light is drawn here, after the BSDF half: same stream position, two registers fewer across it
            if (active) { ul0 = sampler_next<DEBUG_SAMPLER>(smp); ul1 = sampler_next<DEBUG_SAMPLER>(smp); }
            if (sq) estimate_by_emitter_deferred<true>(S, v, wo, li, ul0, ul1, active, beta, weight * 0.5f, tag, *sq);
            else if (ra) estimate_by_emitter_ride<true>(S, Lds, v, wo, li, ul0, ul1, active, *ra, al, wt);
            else if (active) estimate_by_emitter<true>(S, Lds, v, wo, li, ul0, ul1, al, wt);
        } else {
            if (active) { ul0 = sampler_next<DEBUG_SAMPLER>(smp); ul1 = sampler_next<DEBUG_SAMPLER>(smp); }
            KY_CLK(3);
            if (strategy == KY_DIRECT_BSDF_MIS) {
                estimate_by_bsdf<true>(S, Lds, v, wo, li, ub0, ub1, active, ab, wt);
            } else if (strategy == KY_DIRECT_LIGHT_MIS) {
                if (sq) estimate_by_emitter_deferred<true>(S, v, wo, li, ul0, ul1, active, beta, weight, tag, *sq);
                else if (active) estimate_by_emitter<true>(S, Lds, v, wo, li, ul0, ul1, al, wt);
            } else if (strategy == KY_DIRECT_LIGHT) {
                if (sq) estimate_by_emitter_deferred<false>(S, v, wo, li, ul0, ul1, active, beta, weight, tag, *sq);
                else if (active) estimate_by_emitter<false>(S, Lds, v, wo, li, ul0, ul1, al, wt);
            } else if (strategy == KY_DIRECT_BSDF) {
                const int lk = S->light[li].kind;
                if (!S.is_delta(lk)) {  // the third float2 is drawn after the delta test (3894-3900)
                    float u0 = any_f(), u1 = any_f();
                    if (active) { u0 = sampler_next<DEBUG_SAMPLER>(smp); u1 = sampler_next<DEBUG_SAMPLER>(smp); }
                    estimate_by_bsdf<false>(S, Lds, v, wo, li, u0, u1, active, ab, wt);
                }
            }
            // KY_DIRECT_IDLE: estimate_direct_lighting_idle, 3880-3886
        }
        if (decisions) {
            if (active) Lo = Lo + w * (Lb + Ll);
            if (li < 16) *decisions |= (is_black(Lb) ? 0u : 1u << li) | (is_black(Ll) ? 0u : 1u << (16 + li));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// one path = one camera sample.  path_tracing_iteration_t::Li (4529-4617), direct_lighting_t::Li
// (4136-4154) and debug_integrator_t::Li (4105-4122) share this state machine: step() advances
// the path by one vertex and returns false when the path has ended (radiance complete in Lo).
// ---------------------------------------------------------------------------------------------
struct PathState {
    f3 o, d;  // current ray
    f3 beta, Lo;
    Sampler smp;
    int bounces;
    bool prev_specular;
};

// everything of a path's state but Lo becomes "any register content" (callers: lanes that hold no path)
KY_DEV void path_state_dead(PathState& ps) {
    ps.o = mk3(any_reg(), any_reg(), any_reg()); ps.d = mk3(any_reg(), any_reg(), any_reg()); ps.beta = mk3(any_reg(), any_reg(), any_reg());
    ps.smp.s0 = any_reg_u(); ps.smp.s1 = any_reg_u(); ps.bounces = (int)any_reg_u(); ps.prev_specular = any_reg_u() != 0;
}


// KEEP_LO (render_kernel): ps.Lo is not the sample's radiance but the running sum of the lane's pixel chunk -- every contribution of every
// path of the chunk is added to it, the kernel scales and flushes it once per chunk -- so a new path leaves it alone.
template <bool DEBUG_SAMPLER, bool KEEP_LO = false>
KY_DEV void path_begin(PathState& ps, SceneRef S, uint32_t pixel_key, int x, int y, int sample) {
    sampler_start(ps.smp, pixel_key, (uint32_t)sample);
    // get_camera_sample, 943-946 / 971-974
    const float u0 = sampler_next<DEBUG_SAMPLER>(ps.smp), u1 = sampler_next<DEBUG_SAMPLER>(ps.smp);
    generate_ray(S, (float)x + u0, (float)y + u1, ps.o, ps.d);
    ps.beta = mk3(1, 1, 1);
    if (!KEEP_LO) ps.Lo = mk3(0, 0, 0);
    ps.bounces = 0;
    ps.prev_specular = false;
}

// First half of a path vertex: trace the current ray and account for what the hit (or miss) itself contributes.
// Returns false when the path has ended (radiance complete in ps.Lo); true when `v` holds a vertex to shade.
template <bool DEBUG_SAMPLER>
KY_DEV bool path_intersect(PathState& ps, Vertex& v, SceneRef S, const LdsScene& Lds, const RenderConst& rc) {
    float t = K_INF;
    KY_PROBE(0);
    const int hs = trace_nearest(S, ps.o, ps.d, t);  // scene->intersect, 4542
    const bool hit = hs >= 0;

    if (hit) {
        v.t = t;
        v.position = ps.o + t * ps.d;
        v.normal = hit_normal(Lds.hit[hs], v.position, ps.d);
        v.surface = hs;
    }
    // What the hit (or the miss) itself adds: the hit surface's emission towards the ray (surface_t::intersect 3084 + areal_radiance 2957) or the
    // environment's radiance (environment_lighting, 3231) -- for the lanes whose integrator counts it at this vertex, added in place under
    // their predicate.  (Rounds 1-3 computed an `emission` value for every lane first: a default of zero at each of three nesting levels, nine
    // v_mov and six selects per loop turn for a term one vertex in three or four takes.)
    bool see;
    if (rc.integrator == KY_INTEGRATOR_PATH_TRACING_ITERATION || rc.integrator == KY_INTEGRATOR_DIRECT_LIGHTING ||
        rc.integrator == KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED)
        see = ps.bounces == 0 || ps.prev_specular;             // 4548-4559 / 4449-4452
    else if (rc.integrator == KY_INTEGRATOR_PATH_TRACING_RECURSION)
        see = ps.bounces == 0;                                 // 4328-4331
    else
        see = rc.integrator == KY_INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION;   // every way out of 4201-4237 returns it; the debug integrators: never
    if (see) {
        if (hit) {
            if (!S.no_carried_light()) {   // (a scene lit by one delta or environment light: no surface carries a light)
            const int al = Lds.hit[hs].area_light;
            if (al >= 0 && dot(v.normal, ps.d) < 0)            // areal_radiance: the side the (ray-facing) normal looks at, wo = -d
                ps.Lo = ps.Lo + ps.beta * mk3(Lds.light_color[al][0], Lds.light_color[al][1], Lds.light_color[al][2]);
            }
        } else if (S.may_have_env() && S->env_light >= 0) {
            ps.Lo = ps.Lo + ps.beta * ld3(S->light[S->env_light].color);
        }
    }
    if (!hit) return false;  // 4563 / 4141 / 4454 / 4333 / 4204; the debug integrators return black on a miss (4121)
    if (rc.integrator == KY_INTEGRATOR_PATH_TRACING_ITERATION || rc.integrator == KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED ||
        rc.integrator == KY_INTEGRATOR_PATH_TRACING_RECURSION || rc.integrator == KY_INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION)
        if (ps.bounces >= rc.max_path_depth) return false;     // 4563 / 4454 / 4333 / 4204-4210
    return true;
}

// material->scattering's lobe decision on its own (3083, 2663): draws the plastic lobe number from the path's stream
template <bool DEBUG_SAMPLER>
KY_DEV int path_pick_lobe(PathState& ps, int surface, const LdsScene& Lds) {
    const DMat& M = Lds.mat[Lds.hit[surface].material];
    float lobe_u = 0.f;
    if (M.kind == KY_MATERIAL_PLASTIC) lobe_u = sampler_next<DEBUG_SAMPLER>(ps.smp);
    return pick_lobe(M, lobe_u);
}

// Second half: material, direct lighting, continuation.  WAVE-UNIFORM call: every lane of the wave calls it, `active`
// says whether this lane holds a vertex.  Returns true when the (active) lane's path continues.
// per-vertex trace of one path (kyhip_kat_li_trace): rows of 26 floats, the format of the oracle's kyo_trace_li
struct VertexTrace {
    float* rows;
    int max_rows, n;
};

// `tr` is a null constant everywhere but in the trace KAT kernel, which removes the tracing code.
template <bool DEBUG_SAMPLER>
KY_DEV bool path_shade(PathState& ps, Vertex& v, SceneRef S, const LdsScene& Lds, const RenderConst& rc, bool active,
                       int lobe = -1, VertexTrace* tr = nullptr, ShadowQueue* sq = nullptr, unsigned tag = 0, bool ride_along = false, bool free_state = false) {
    if (active) {
        // material->scattering(isect) for the nearest hit (3083); only plastic draws a lobe number (2663).
        // lobe >= 0: the caller has already made that draw (path_pick_lobe).
        const DMat& M = Lds.mat[Lds.hit[v.surface].material];
        if (lobe < 0) {
            float lobe_u = 0.f;
            if (M.kind == KY_MATERIAL_PLASTIC) lobe_u = sampler_next<DEBUG_SAMPLER>(ps.smp);
            v.bsdf = make_bsdf(M, lobe_u, (S.feat & KY_FEAT_NO_DELTA) != 0);
        } else {
            v.bsdf = make_bsdf_for_lobe(M, lobe);
        }
        vertex_prepare(v, -ps.d, &Lds.hit[v.surface]);   // ps.d still holds the direction of the ray that found this vertex
    }
    const f3 wo = -ps.d;   // isect.wo, 3125

    if (rc.integrator < KY_INTEGRATOR_DIRECT_LIGHTING) {  // debug_integrator_t, 4110-4118 (wave-uniform)
        if (active) {
            // (added, not assigned: in the render kernel Lo is the running sum of the pixel chunk; everywhere else it is zero here)
            if (rc.integrator == KY_INTEGRATOR_POSITION) ps.Lo = ps.Lo + normalize(v.position);
            else if (rc.integrator == KY_INTEGRATOR_NORMAL) ps.Lo = ps.Lo + v.normal;
            else {
                float pdf, abs_cos_i;
                f3 base;
                bsdf_eval_pdf(v, wo, v.normal, base, pdf, abs_cos_i);
                ps.Lo = ps.Lo + base;
            }
        }
        return false;
    }

    KY_CLK(2);
    const bool recursion = rc.integrator == KY_INTEGRATOR_PATH_TRACING_RECURSION;
    const bool defered = rc.integrator == KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED;
    const bool simple = rc.integrator == KY_INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION;
    const bool delta = bsdf_is_delta(v.bsdf);
    const bool nee = active && !delta;  // 4571
    KY_PROBE(6);
    unsigned decisions = 0;
    // path_tracing_recursion_t, specular vertex (4341-4349): look the emitter up along a sampled direction, from the un-offset hit point
    // (the continuation below draws a NEW sample).  `ride_along` (that integrator's own instantiations): the specular lanes -- which draw
    // nothing in the light loop -- sample the direction here and let the loop's first traversal carry the ray (RideAlong).
    const bool riding = recursion && ride_along;
    RideAlong ra{false, any3(), any3(), K_INF, -1};
    f3 es_f = any3();
    float es_k = any_f();
    if (riding && active && delta) {
        const float e0 = sampler_next<DEBUG_SAMPLER>(ps.smp), e1 = sampler_next<DEBUG_SAMPLER>(ps.smp);
        const BsdfSample es = bsdf_sample(v, wo, e0, e1);
        ra.want = true;
        ra.o = v.position;
        ra.d = es.wi;
        es_f = es.f;
        es_k = fabsf(dot(es.wi, v.normal)) / es.pdf;
    }
    if (!simple) {  // simple_path_tracing_recursion_t samples the BSDF only
        sample_all_light<DEBUG_SAMPLER>(S, Lds, v, wo, ps.smp, rc.strategy, nee, ps.Lo, ps.beta, tr ? &decisions : nullptr, sq, rc.inv_spp, tag,
                                        riding ? &ra : nullptr);  // 4575 / 4337 / 4458
    }
    KY_CLK(8);
    if (rc.integrator == KY_INTEGRATOR_DIRECT_LIGHTING) return false;  // 4153
    if (!active) {
        // A lane without a vertex holds no path: its ray, throughput, sampler and depth are dead until path_begin rewrites them.  Saying so -- they
        // become "any register content" here -- lets the values the active lanes compute below BE the loop-carried state instead of being
        // copied into it (sixteen v_mov per loop turn in the round-3 listing).
        if (free_state) path_state_dead(ps);
        return false;
    }

    if (recursion && delta) {
        if (!riding) {
            const float e0 = sampler_next<DEBUG_SAMPLER>(ps.smp), e1 = sampler_next<DEBUG_SAMPLER>(ps.smp);
            const BsdfSample es = bsdf_sample(v, wo, e0, e1);
            ra.o = v.position;
            ra.d = es.wi;
            ra.want = true;
            es_f = es.f;
            es_k = fabsf(dot(es.wi, v.normal)) / es.pdf;
        }
        if (ra.want) {   // nobody carried the ray (no light loop, or none of its traversals ran)
            ra.t = K_INF;
            ra.hs = trace_nearest(S, ra.o, ra.d, ra.t);
        }
        f3 Le = (S.may_have_env() && S->env_light >= 0) ? ld3(S->light[S->env_light].color) : mk3(0, 0, 0);
        if (ra.hs >= 0) {
            const f3 hp = ra.o + ra.t * ra.d;
            Le = surface_emission(Lds, ra.hs, hit_normal(Lds.hit[ra.hs], hp, ra.d), -ra.d);
            // scene->intersect builds the hit's BSDF: a plastic surface draws its lobe number (2663)
            if (Lds.mat[Lds.hit[ra.hs].material].kind == KY_MATERIAL_PLASTIC) (void)sampler_next<DEBUG_SAMPLER>(ps.smp);
        }
        ps.Lo = ps.Lo + ps.beta * ((es_f * Le) * es_k);  // 0/0 = NaN on total internal reflection, as in 4349
    }

    // sample BSDF to get the new path direction, 4586 / 4213 / 4383 / 4495
    const float u0 = sampler_next<DEBUG_SAMPLER>(ps.smp), u1 = sampler_next<DEBUG_SAMPLER>(ps.smp);
    KY_PROBE(7);
    if (rc.integrator == KY_INTEGRATOR_PATH_TRACING_ITERATION && !tr) {
        // path_tracing_iteration_t, 4586-4616, with the sample's value, cosine and pdf folded into one factor (bsdf_continue).
        // Straight-line: a path that ends here never reads its ray or throughput again, so every lane updates them and `cont` alone says
        // whether the path goes on -- no copies of the old state on the ways out, no exec-mask bookkeeping around them.
        const BsdfContinue c = bsdf_continue(v, wo, u0, u1, S.flat_phong());
        bool cont = c.ok;  // 4588
        ps.beta = ps.beta * c.weight;  // 4592
        ps.prev_specular = c.specular;  // 4596
        ps.o = offset_ray_origin(v.position, v.normal, c.wi);  // 4597
        ps.d = c.wi;
        if (ps.bounces > 3) {  // Russian roulette, 4601-4612 (the number is drawn by the paths that get here only: the stream of the others must not move)
            const float q = fmaxf(0.05f, 1 - max3(ps.beta));
            const float u = sampler_next<DEBUG_SAMPLER>(ps.smp);
            cont = cont && !(u < q);
            ps.beta = ps.beta * rcp(1 - q);
        }
        ps.bounces += 1;
        // The vertex at bounces == max_depth can only add emission after a delta bounce (4548, 4563):
        // when the previous bounce was not specular that last traversal cannot change Lo, so skip it.
        return cont && !(ps.bounces >= rc.max_path_depth && !ps.prev_specular);
    }
    // the reference's own value and pdf: the vertex trace records them, the recursive integrators play roulette on the value
    BsdfSample bs = bsdf_sample(v, wo, u0, u1);
    if (tr && tr->n < tr->max_rows) {
        float* r = tr->rows + 26 * tr->n++;
        r[0] = (float)ps.bounces; r[1] = (float)S->orig[v.surface]; r[2] = (float)v.bsdf.lobe;
        r[3] = v.position.x; r[4] = v.position.y; r[5] = v.position.z; r[6] = v.normal.x; r[7] = v.normal.y; r[8] = v.normal.z;
        r[9] = wo.x; r[10] = wo.y; r[11] = wo.z; r[12] = ps.beta.x; r[13] = ps.beta.y; r[14] = ps.beta.z;
        r[15] = ps.Lo.x; r[16] = ps.Lo.y; r[17] = ps.Lo.z; r[18] = bs.f.x; r[19] = bs.f.y; r[20] = bs.f.z; r[21] = bs.pdf;
        r[22] = fabsf(dot(bs.wi, v.normal)); r[23] = (float)bs.flags; r[24] = (float)(decisions & 0xffffu); r[25] = (float)(decisions >> 16);
    }
    if (is_black(bs.f) || bs.pdf == 0.f) return false;  // 4588 / 4215 / 4385 / 4497
    if (rc.integrator == KY_INTEGRATOR_PATH_TRACING_ITERATION) {
        ps.beta = ps.beta * (bs.f * (fabsf(dot(bs.wi, v.normal)) * rcp(bs.pdf)));  // 4592
        ps.prev_specular = (bs.flags & BSDF_SPECULAR) != 0;  // 4596
        ps.o = offset_ray_origin(v.position, v.normal, bs.wi);  // 4597
        ps.d = bs.wi;
        if (ps.bounces > 3) {  // Russian roulette, 4601-4612
            const float q = fmaxf(0.05f, 1 - max3(ps.beta));
            const float u = sampler_next<DEBUG_SAMPLER>(ps.smp);
            if (u < q) return false;
            ps.beta = ps.beta * rcp(1 - q);
        }
        ps.bounces += 1;
        // The vertex at bounces == max_depth can only add emission after a delta bounce (4548, 4563):
        // when the previous bounce was not specular that last traversal cannot change Lo, so skip it.
        if (ps.bounces >= rc.max_path_depth && !ps.prev_specular) return false;
        return true;
    }
    // the recursive integrators: roulette on the BSDF value once ++depth > 3 (4219-4226, 4389-4397, 4501-4509)
    if (ps.bounces + 1 > 3) {
        const float m = max3(bs.f);
        const float u = sampler_next<DEBUG_SAMPLER>(ps.smp);
        if (!(u < m)) return false;
        bs.f = bs.f * (1 / m);
    }
    ps.beta = ps.beta * (bs.f * (fabsf(dot(bs.wi, v.normal)) * rcp(bs.pdf)));  // 4233 / 4400 / 4512
    ps.prev_specular = delta;  // isect.bsdf()->is_delta(), 4512 (only the defered variant reads it)
    ps.o = recursion ? offset_ray_origin(v.position, v.normal, bs.wi) : v.position;  // 4399 vs 4232 / 4511
    ps.d = bs.wi;
    ps.bounces += 1;
    // a traversal at depth == max_depth adds nothing for the two NEE recursions unless (defered) the bounce was specular
    if (ps.bounces >= rc.max_path_depth && (recursion || (defered && !ps.prev_specular))) return false;
    return true;
}

}  // namespace kyd
