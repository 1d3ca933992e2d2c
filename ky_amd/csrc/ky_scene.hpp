/*
 * ky_scene.hpp -- the packed scene as the device reads it (DScene and its records), the scene facts an instantiation may assume (KY_FEAT_*)
 * and the launch constants: plain data, shared by the host code that builds it (ky_pack.cpp: plain C++, also built by g++ with sanitizers,
 * `make sanitize`) and the device code that reads it (ky_device.hpp).  Nothing here is device code; KY_HD marks the few inline helpers both sides call.
 * Every record names the reference lines (file = /root/reference/ky.cpp) it restates data of.
 */
#pragma once
#ifndef __HIPCC_RTC__   // hiprtc brings the vector and the fixed-width integer types itself
#include <hip/hip_vector_types.h>
#include <stdint.h>
#endif

#include "../../include/kyhip.h"

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define KY_HD __host__ __device__
#else
#define KY_HD
#endif

namespace kyd {

// ---------------------------------------------------------------------------------------------
// device scene layout (HBM, read-only)
// ---------------------------------------------------------------------------------------------
enum : int {  // traversal kinds
    TK_DISK = KY_SHAPE_DISK,
    TK_TRIANGLE = KY_SHAPE_TRIANGLE,
    TK_QUAD = KY_SHAPE_RECTANGLE,  // general (non-parallelogram / non-planar) quad: the reference's edge tests
    TK_SPHERE = KY_SHAPE_SPHERE,
    TK_PARALLELOGRAM = 4           // planar parallelogram: plane hit + dual-basis coordinates
};

// 64-byte traversal record, read with a wave-uniform index (scalar loads):
//   TK_PARALLELOGRAM  f[0..2] = n, f[3] = n.p0, f[4..6] = a*, f[7] = a*.p1 + 0.5, f[8..10] = b*, f[11] = b*.p1 + 0.5
//                     where a = p0 - p1, b = p2 - p1 and (a*, b*) is the dual basis in the plane, so that for a point
//                     h of the plane  h.a* - f[7] = u - 0.5,  h.b* - f[11] = v - 0.5  with h = p1 + u a + v b
//   TK_SPHERE         f[0..2] = centre, f[3] = radius^2
//   other kinds       full = index of the DShapeFull record
struct DSurf {
    float f[12];
    int32_t kind;
    int32_t full;
    int32_t pad[2];
};

struct DShapeFull {  // the reference's own shape data (ky_shape)
    float p[4][3];
    float n[3];
    float radius;
    float radius_sq;
    int32_t kind;
    int32_t pad[2];
};  // 80 B

struct DHit {  // what is needed once the nearest surface is known; gathered per lane from LDS
    float n[3];  // stored normal, or the sphere centre
    int32_t kind;
    int32_t material;
    int32_t area_light;
    float fs[3], ft[3];   // planar shapes: frame_t(stored normal)'s s and t (537-541), made once by the host instead of at every vertex (surface_frame below)
    int32_t pad_h[4];     // 64 B: a per-lane index becomes an LDS address by a shift (24 B cost a v_mul_lo_u32 at each of three look-ups per loop turn)
};
static_assert(sizeof(DHit) == 64, "DHit");

struct DMat {  // ky_material, gathered per lane from LDS
    float c0[3];        // lambert albedo | mirror R | glass R; plastic: Kd / P_diff, the Lambert lobe's albedo (2667)
    int32_t kind;
    float c1[3];        // glass T; plastic: cs (n + 2) / (n + 1), the Phong lobe's value / pdf per unit |cos| (bsdf_continue)
    float eta;          // glass: eta; plastic: 1 / (exponent + 1), the power of the Phong lobe's cos(theta) = u^(1/(n+1)) (2515)
    float exponent, phong_pdf_norm, p_specular;   // phong_pdf_norm = (exponent + 1) / 2 pi (2549)
    int32_t exp_flags;  // bit 0: exponent is integral, bit 1: it is odd (sign of pow(negative, n)); bit 2 / 3: c0 / c1 is not black (some channel positive, or NaN);
                        // bits 16-31: plastic: the upper half of a float F such that pow(|x|, exponent) underflows to 0 for |x| <= F (phong_pow_lobe), else 0
    float cs[3];        // plastic: Ks / P_spec, the Phong lobe's colour (2665)
    float inv_eta;      // glass: 1 / eta (`eta_i / eta_t` entering, 1977 / 2388: the same float division, done once on the host);
                        // plastic: (exponent + 2) / 2 pi, the Phong lobe's normalisation (2505)
};  // 64 B

struct DLight {  // light_t + the shape an area light samples; wave-uniform index
    float color[3];
    int32_t kind;
    float position[3];
    float world_radius;
    float direction[3];
    int32_t shape_kind;   // ky_shape_kind of the sampled shape
    float p1[3];          // rectangle: p1, e0 = p0 - p1, e1 = p2 - p1 (1310); triangle: p0, p1, p2; sphere / disk: centre
    float radius;
    float e0[3];          // (a sphere light: e0[0] = 1 / radius)
    float area;
    float e1[3];
    float inv_area;
    float n[3];           // stored normal
    int32_t n_carriers;   // surfaces whose surface_t::area_light is this light (sorted indices); -1: more than KY_MAX_CARRIERS
    int32_t carrier[4];
    int32_t sampled_is_surface;   // the shape this light samples is also the shape of some surface of the scene (so it occludes)
    int32_t occ_ok;               // shadow rays towards samples of this light may use DScene::occ
    int32_t pdf_from_carrier;     // the light samples a planar shape and its ONE carrier surface has that very shape: pdf_direction's re-intersection of the light's
                                  // shape with isect.spawn_ray(wi) (1057-1061) IS the BSDF-sampling estimator's carrier hit -- same ray, same record, same arithmetic
    int32_t shadow_table;         // byte offset (from the scene's base) of the planar table a shadow ray towards a sample of this light scans first -- DScene::occ_front for the
                                  // two-stage light (bit 0 set: occ_behind follows for rays with an end behind its plane), DScene::occ when occ_ok, DScene::trav otherwise --
                                  // decided once by the host instead of by three scalar loads and two compares per light sample
    float aar[5];                 // the sampled shape as a rectangle in an axis plane (DAar's c, mu, ru, mv, rv) when it is one (a lamp that is no rectangle: aar[0..3] = its first
                                  // carrier's sphere record, centre and radius^2, when that carrier is a sphere) ...
    int32_t aar_axis;             // ... in the plane x_axis = c; -1: it is not (KY_FEAT_AXIS_ALIGNED kernels test the lamp with it: estimate_by_bsdf)
    int32_t pad_a[2];
    DSurf isect;          // traversal record of the sampled shape (pdf_direction re-intersects it, 1057-1061)
};
constexpr int KY_MAX_CARRIERS = 4;
static_assert(sizeof(DLight) % 16 == 0 && __builtin_offsetof(DLight, p1) % 16 == 0 && __builtin_offsetof(DLight, n) % 16 == 0, "DLight is read in 16-byte groups (shape_sample_position)");

struct DPar {  // planar parallelogram, 48 B: q0 = (n, n.p0), q1 = (a*, a*.p1 + 0.5), q2 = (b*, b*.p1 + 0.5)
    float4 q0, q1, q2;
};
struct DSph {  // sphere, 16 B: centre, radius^2
    float4 c;
};
// axis-aligned rectangle in the plane x_axis = c, 32 B.  With (u, v) = the other two axes in cyclic order and [lo, hi] the
// rectangle's extent along them: q0 = (c, mu, ru, mv), q1.x = rv with m = (lo + hi) / 2, r = (hi - lo) / 2: a point h of
// the plane is inside iff |h_u - mu| <= ru and |h_v - mv| <= rv (borders inclusive, like the parallelogram test; each
// test is one subtract and one compare with a single scalar operand).  q1.y = the surface's SORTED index (an int's bits): the nearest-hit scans
// note it straight from the record, so a table's order carries no meaning beyond the order of exact ties (DBoxTrav::aar leaves rectangles out).
struct DAar {
    float4 q0, q1;
};

// An axis-aligned BOX some of whose six faces are surfaces of the scene, each exactly a whole face (host: find_boxes -- the room of a Cornell box, its lamp
// housing): 48 B.  q0 = (lo, s[0]), q1 = (hi, s[1]), q2 = (s[2], s[3], s[4], s[5]) with s[f] = the sorted surface index of face f = 2 axis + (0: the lo plane,
// 1: the hi plane) as an integer's bits, KY_BOX_NO_FACE = 15 for an open side -- so a box's faces have sorted indices below 15.  A ray meets the boundary of a
// convex box where it enters and where it leaves it, so the nearest hit among up to six rectangles is one slab test (box_update_nearest, ky_device.hpp:
// 40 VALU instructions against 12 per rectangle).
struct DBox {
    float4 q0, q1, q2;
};
constexpr int KY_MAX_BOXES = 4, KY_BOX_MIN_FACES = 4, KY_BOX_NO_FACE = 15;
// What a nearest-hit traversal of an instantiation with KY_FEAT_BOXES scans instead of DTrav::aar: the boxes, then the axis-aligned rectangles that are no
// box's face (grouped by axis like DTrav::aar); parallelograms, spheres and general shapes as ever.
struct DBoxTrav {
    int32_t n_box, n_aar, pad_b0, pad_b1;
    int32_t n_aar_axis[3], pad_b2;
    DBox box[KY_MAX_BOXES];
    DAar aar[KYHIP_MAX_SURFACES + 1];
};

// The surfaces are stored SORTED BY TRAVERSAL KIND -- axis-aligned rectangles (x, y, z planes), other parallelograms, then
// spheres, then everything else -- keeping the
// reference's surface order inside each group, so that each traversal loop is branch-free.  `hit[]` and every surface
// index used on the device are in this sorted order; orig[] maps back to the caller's surface index.  (Ties: the
// reference's "first surface in list order wins an exactly equal distance" (3177-3180) is preserved inside a group;
// an exact tie between shapes of different kinds has measure zero.)
struct DTrav {   // the planar part of a traversal table: axis-aligned rectangles grouped by axis (x, y, z planes), then other parallelograms
    int32_t n_aar, n_par, pad_t0, pad_t1;   // n_aar = n_aar_axis[0] + [1] + [2]
    int32_t n_aar_axis[3], pad_t2;
    DAar aar[KYHIP_MAX_SURFACES + 1];   // one readable record past the end: the traversal reads i + 1
    DPar par[KYHIP_MAX_SURFACES + 1];
};
struct DScene {
    int32_t n_surfaces, n_lights, n_materials, env_light;
    int32_t n_sph, n_gen, general, occ_deferred_ok;   // general: SceneRef::general; occ_deferred_ok: shadow rays towards every light may use `occ`
    float cam_position[3], cam_inv_w;
    float cam_front[3], cam_inv_h;
    float cam_right[3], pad0;
    float cam_up[3], pad1;
    DTrav trav;                         // every planar surface; its order is the sorted surface order
    // Occluder tables (host: find_non_occluders, which states the conditions): `trav` without surfaces that provably hold no point of a
    // shadow ray.
    //  occ            without the walls of a room: planar surfaces that have the whole scene in one closed half-space of their plane.
    //                 For rays that end at a scene point (the MIS rays' "is anything in front of the carrier" query) and, when
    //                 occ_deferred_ok, for the deferred shadow rays of all lights.
    //                 Shadow rays towards a SAMPLE of light li (by_emitter) use it when DLight::occ_ok: the light keeps clear of every
    //                 wall's plane by more than the ray origin's offset (never for directional / environment lights).
    //  Spheres are in none of these tables: they are always tested.
    DTrav occ;
    // Two-stage scan for shadow rays towards ONE planar area light (ts_light; -1: none): `occ` split by the plane n.x = k of the light's
    // sampled shape into occ_front (everything not entirely in n.x <= k, plus the light's own surfaces) and occ_behind (the planar
    // surfaces entirely in n.x <= k: what is mounted behind a lamp).  A segment has a point in that half-space only if one of its ends
    // has; such rays overshoot the lamp (quirk 1) and the lamp itself stops nearly all of them, so occ_behind is scanned only for the
    // few that are left, under a wave-uniform branch.
    int32_t ts_light, feat, ts_pad[2];   // feat: the KY_FEAT_* facts that hold for this scene (host: pack_scene)
    float ts_plane[4];
    DTrav occ_front, occ_behind;
    DBoxTrav boxtrav;                   // KY_FEAT_BOXES: the nearest-hit traversal's planar part with the boxes' faces taken out of the rectangle lists
    DSph sph[KYHIP_MAX_SURFACES + 1];
    DSurf gen[KYHIP_MAX_SURFACES];
    DSurf all[KYHIP_MAX_SURFACES];      // every surface as a generic record, sorted order (carrier tests, surface-parallel queries)
    DShapeFull full[KYHIP_MAX_SURFACES + KYHIP_MAX_LIGHTS];
    DHit hit[KYHIP_MAX_SURFACES];
    int32_t orig[KYHIP_MAX_SURFACES];
    DMat mat[KYHIP_MAX_MATERIALS];
    DLight light[KYHIP_MAX_LIGHTS];
};
static_assert(__builtin_offsetof(DScene, light) % 16 == 0 && __builtin_offsetof(DScene, trav) % 16 == 0 && __builtin_offsetof(DScene, boxtrav) % 16 == 0 &&
              __builtin_offsetof(DBoxTrav, box) % 16 == 0, "16-byte scalar loads of light, table and box records");


// How device functions see the scene: the pointer plus one compile-time fact.  `general` = the scene may hold shapes that
// need the reference's own formulations (quads that are not parallelograms, triangles, disks: full_shape_hit, ~150 VALU and
// the register peak of the whole kernel).  The hot instantiation of the render kernel is launched only for scenes without
// them (every scene ky ships) and passes `false`, which removes that code; everything else converts from the bare pointer.
// Compile-time facts about a scene (SceneRef::feat, a mask): what a render-kernel instantiation may assume, and so what code it does
// not carry.  Code a scene never executes still costs it registers and instruction-cache space; each of these was measured
// (DESIGN.md 3).  The host computes the scene's facts (pack_scene -> DScene::feat) and launches an instantiation whose assumptions
// are a subset of them; 0 assumes nothing.
enum : int {
    KY_FEAT_SINGLE_AREA = 1,     // the lights are exactly ONE area light, no environment light: no other light kind's code, no environment term, no lights loop
    KY_FEAT_RECT_LIGHTS = 2,     // every area light samples a rectangle (the Cornell lamp): no sphere / triangle / disk light sampling
    KY_FEAT_CARRIERS = 4,        // every area light is carried by at most KY_MAX_CARRIERS surfaces and the scene has no general shapes: the
                                 // BSDF-sampling estimators always take the carrier test, never the full traversal (estimate_by_bsdf)
    KY_FEAT_SINGLE_DELTA = 8,    // the lights are exactly ONE point or directional light, no environment light: the BSDF-sampling estimators
                                 // are gone (they return black for a delta light, 3894 / 3977), and with them every area / environment path
    KY_FEAT_SINGLE_ENV = 16,     // the lights are exactly ONE environment light (which is the scene's environment): no area / delta light code
    KY_FEAT_SPHERE_LIGHTS = 32,  // every light is an area light that samples a SPHERE and is carried by sphere surfaces only, no environment light (the
                                 // Veach scene's five): no other light kind's or light shape's code, no dispatch on either per light and vertex
    KY_FEAT_NO_DELTA = 64,       // no material is a mirror or glass: no delta lobe's code, prev_specular is never set
    KY_FEAT_OWN_CARRIER = 256,   // every area light samples a planar parallelogram and is carried by exactly ONE surface, which has that very shape (a lamp that is its own
                                 // emitting rectangle: DLight::pdf_from_carrier for every light): the BSDF-sampling estimators test DLight::isect directly -- no carrier
                                 // list, no dispatch on the carrier's kind, no look-up of its normal (a rectangle seen by a ray emits on both sides, 1289 / 2957)
    KY_FEAT_BOXES = 512,         // some axis-aligned rectangles are whole faces of common boxes (DBox; at least KY_BOX_MIN_FACES faces each): the nearest-hit traversal
                                 // tests a box with one slab test instead of its faces one by one (DScene::boxtrav)
    KY_FEAT_AXIS_ALIGNED = 1024, // every planar surface is a rectangle in an axis plane and there are no general shapes: no parallelogram loops in the traversals, and a lamp
                                 // that is its own carrier (KY_FEAT_OWN_CARRIER) is tested as such a rectangle
    KY_FEAT_FLAT_PHONG = 2048,   // every surface with a plastic material is a rectangle -- a shape that reports the normal facing the ray (1289) --, so no Phong lobe is ever
                                 // entered from below its normal: the `if (wo.z < 0) wi.z *= -1` of 2539 is dead code
    KY_FEAT_X_PLANKS = 4096,     // every planar parallelogram that is not an axis rectangle is a plank tilted about the x axis -- n.x = 0, one edge along x, the other in the
                                 // y-z plane (create_mis_scene's four, 3469-3479): its record's q0.x, q1.x, q2.y, q2.z are zeros, and the products with them are not computed
                                 // by the any-hit scans (five instructions per plank and ray; the sums round where the general form's do)
    KY_FEAT_SMALL_TABLES = 128   // at most KY_LDS_SURFACES_SMALL surfaces and KY_LDS_MATERIALS_SMALL materials: the per-lane tables' LDS block is 1.1 KB instead of
                                 // 3.8 (what lets the sphere-lights kernel with its deferred rays' sums fit a seventh workgroup per CU)
};
constexpr int KY_FEAT_SINGLE_LIGHT = KY_FEAT_SINGLE_AREA | KY_FEAT_SINGLE_DELTA | KY_FEAT_SINGLE_ENV;   // any of them: no lights loop
// (Measured and not kept: "every area light samples a sphere" + "no mirror or glass material" for the Veach scene: 11 fewer spilled
// registers in the instantiation with deferred shadow rays, no change in time.)

// The per-workgroup LDS copy of the tables that are indexed per lane: hit[n_surfaces], mat[n_materials], light_color[n_lights][4].
// Two homes.  The standard kernels keep a STATIC block for scenes of up to KY_LDS_SURFACES surfaces and KY_LDS_MATERIALS materials
// (every scene ky ships has 11-13 and 4-8): its addresses are compile-time constants that fold into the ds_read offsets.  Larger
// scenes (up to the ABI's KYHIP_MAX_*) run on the LARGE instantiations, which size the block by the scene in dynamic shared memory
// (lds_scene_bytes() at launch) and pay an add per table access for it -- measured on the kernels that do not need it: Veach -2.2 %,
// Cornell -0.5 %, which is why they keep the static block.
constexpr int KY_LDS_SURFACES = 64, KY_LDS_MATERIALS = 32;
constexpr int KY_LDS_SURFACES_SMALL = 16, KY_LDS_MATERIALS_SMALL = 8;
KY_HD inline int lds_scene_mat_offset(int n_surfaces) { return (n_surfaces * (int)sizeof(DHit) + 15) & ~15; }
KY_HD inline int lds_scene_light_offset(int n_surfaces, int n_materials) { return lds_scene_mat_offset(n_surfaces) + n_materials * (int)sizeof(DMat); }
KY_HD inline int lds_scene_bytes(int n_surfaces, int n_materials, int n_lights) { return lds_scene_light_offset(n_surfaces, n_materials) + n_lights * 16; }

struct RenderConst {  // wave-uniform launch constants
    int integrator, max_path_depth, strategy;
    uint32_t seed;
    int width, height, spp;
    float inv_spp;
};

}  // namespace kyd
