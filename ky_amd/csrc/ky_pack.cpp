/*
 * ky_pack.cpp -- the host code of libkyhip.so that needs no GPU: error state, parameter validation and shard geometry, packing a caller's ky_scene
 * into the device layout (DScene: kind-sorted traversal tables, occluder tables, scene facts), the proof of which surfaces a shadow ray never has to
 * test (find_non_occluders), the launch policies, and the CPU side of the host-film seam (HostPool, the banded add).
 * Plain C++17 with no HIP runtime call: hipcc compiles it into the library, and `make sanitize` compiles the very same file with
 * g++ -fsanitize=address,undefined and -fsanitize=thread (round 5; until round 4 all of this lived inside kyhip.hip, out of any sanitizer's reach).
 */
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <sched.h>
#include <unistd.h>

#include "ky_host.hpp"

namespace kyh {

// ------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}
const std::string& last_error() { return g_error; }


// ------------------------------------------------------------------------------------------------
// shard geometry (host + device)
// ------------------------------------------------------------------------------------------------
bool valid_params(const ky_render_params* p) {
    if (!p) return false;
    if (p->width <= 0 || p->height <= 0 || p->samples_per_pixel <= 0 || p->max_path_depth < 0 || p->max_path_depth > 250) return false;
    if (p->width > 32767 || p->height > 32767 || p->samples_per_pixel > (1 << 24)) return false;   // packed fields: x | y << 16, sample << 7
    if (p->tile_w <= 0 || p->tile_h <= 0 || (p->tile_w % 8) || (p->tile_h % 8)) return false;
    if (p->tile_first < 0 || p->tile_step <= 0) return false;
    switch (p->integrator) {
    case KY_INTEGRATOR_POSITION: case KY_INTEGRATOR_NORMAL: case KY_INTEGRATOR_BASECOLOR:
    case KY_INTEGRATOR_DIRECT_LIGHTING: case KY_INTEGRATOR_PATH_TRACING_ITERATION:
    case KY_INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION: case KY_INTEGRATOR_PATH_TRACING_RECURSION:
    case KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED: break;
    default: return false;   // create_integrator returns nullptr (ky.cpp:4638)
    }
    switch (p->direct_sample) {
    case KY_DIRECT_IDLE: case KY_DIRECT_BSDF: case KY_DIRECT_LIGHT: case KY_DIRECT_BSDF_MIS:
    case KY_DIRECT_LIGHT_MIS: case KY_DIRECT_BOTH_MIS: break;
    default: return false;   // empty std::function -> bad_function_call (ky.cpp:3860)
    }
    if (p->sampler != KY_SAMPLER_DEBUG && p->sampler != KY_SAMPLER_RANDOM) return false;
    return true;
}

// Index ranges of the device code: work items are counted in 32 bits, accumulator and tile indices are ints.
bool shard_in_range(const ky_render_params* p) {
    const long long tiles_x = (p->width + p->tile_w - 1) / p->tile_w, tiles_y = (p->height + p->tile_h - 1) / p->tile_h;
    const long long total = tiles_x * tiles_y;
    const long long n_tiles = p->tile_first >= total ? 0 : (total - p->tile_first + p->tile_step - 1) / p->tile_step;
    const long long n_pix = n_tiles * p->tile_w * p->tile_h;
    const long long n_blocks = n_tiles * (p->tile_w / 8) * (p->tile_h / 8);
    const long long n_chunks = chunk_count(chunk_plan(p->samples_per_pixel));
    // a wavefront's fetches run past the end of the queue by at most one per wave plus the first-item offset (4 x grid): keep
    // every id such a fetch can produce below 2^32, or it would wrap to a small number and a chunk would be rendered twice
    return n_pix * 3 <= 0x7fffffffLL && n_blocks * n_chunks < 0xffffffffLL - (1 << 20) && total <= 0x7fffffffLL;
}

ShardConst make_shard(const ky_render_params* p) {
    ShardConst s{};
    s.tile_w = p->tile_w; s.tile_h = p->tile_h; s.tile_first = p->tile_first; s.tile_step = p->tile_step;
    s.tiles_x = (p->width + p->tile_w - 1) / p->tile_w;
    s.tiles_y = (p->height + p->tile_h - 1) / p->tile_h;
    const int total = s.tiles_x * s.tiles_y;
    s.n_tiles = p->tile_first >= total ? 0 : (total - p->tile_first + p->tile_step - 1) / p->tile_step;
    s.blocks_w = p->tile_w / 8;
    s.blocks_per_tile = s.blocks_w * (p->tile_h / 8);
    s.n_blocks = s.n_tiles * s.blocks_per_tile;
    s.n_pix = s.n_tiles * p->tile_w * p->tile_h;
    s.n_chunks = chunk_count(chunk_plan(p->samples_per_pixel));
    s.n_items = (unsigned)s.n_blocks * (unsigned)s.n_chunks;
    return s;
}

RenderConst make_rc(const ky_render_params* p) {
    RenderConst rc{};
    rc.integrator = p->integrator; rc.max_path_depth = p->max_path_depth; rc.strategy = p->direct_sample; rc.seed = p->seed;
    rc.width = p->width; rc.height = p->height; rc.spp = p->samples_per_pixel;
    rc.inv_spp = (float)(1. / p->samples_per_pixel);  // ky.cpp:3717
    return rc;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
void cp3(float* d, const float* s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
// frame_t(n) for a unit n (ky.cpp:537-541, 566-571; ky_device.hpp make_frame): t = normalize(n x (|n.x| > 0.99 ? Y : X)), s = t x n
static void host_frame(const float* n, float* fs, float* ft) {
    if (std::fabs(n[0]) > 0.99f) {
        const float k = 1.0f / std::sqrt(n[2] * n[2] + n[0] * n[0]);
        ft[0] = -n[2] * k; ft[1] = 0.f; ft[2] = n[0] * k;
    } else {
        const float k = 1.0f / std::sqrt(n[2] * n[2] + n[1] * n[1]);
        ft[0] = 0.f; ft[1] = n[2] * k; ft[2] = -n[1] * k;
    }
    fs[0] = ft[1] * n[2] - ft[2] * n[1];
    fs[1] = ft[2] * n[0] - ft[0] * n[2];
    fs[2] = ft[0] * n[1] - ft[1] * n[0];
}

static float host_shape_area(const ky_shape& s) {  // shape_t::area x4 (1141, 1222, 1304, 1401), fp32
    auto sub = [](const float* a, const float* b, float* r) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; };
    auto crossmag = [](const float* a, const float* b) {
        const float cx = a[1] * b[2] - a[2] * b[1], cy = a[2] * b[0] - a[0] * b[2], cz = a[0] * b[1] - a[1] * b[0];
        return std::sqrt(cx * cx + cy * cy + cz * cz);
    };
    const float pi = 3.14159265358979323846f;
    float u[3], v[3];
    switch (s.kind) {
    case KY_SHAPE_DISK: return pi * s.radius * s.radius;
    case KY_SHAPE_TRIANGLE: sub(s.p[1], s.p[0], u); sub(s.p[2], s.p[0], v); return 0.5f * crossmag(u, v);
    case KY_SHAPE_RECTANGLE: sub(s.p[0], s.p[1], u); sub(s.p[2], s.p[1], v); return crossmag(u, v);
    default: return 4 * pi * (s.radius * s.radius);
    }
}

// Builds the traversal record of one shape.  A rectangle_t whose four points form a planar parallelogram gets the
// plane + dual-basis form (precomputed in double); every other shape keeps the reference's own data in `full`.
void pack_shape(const ky_shape& sh, int full_index, DSurf* surf, DShapeFull* full) {
    std::memset(surf, 0, sizeof *surf);
    std::memset(full, 0, sizeof *full);
    std::memcpy(full->p, sh.p, sizeof full->p);
    cp3(full->n, sh.normal);
    full->radius = sh.radius; full->radius_sq = sh.radius * sh.radius; full->kind = sh.kind;  // sphere_t::radius_sq_, 1332
    surf->kind = sh.kind;
    surf->full = full_index;
    if (sh.kind == KY_SHAPE_SPHERE) {
        cp3(surf->f, sh.p[0]);
        surf->f[3] = sh.radius * sh.radius;
        return;
    }
    if (sh.kind != KY_SHAPE_RECTANGLE) return;
    double a[3], b[3], e[3], nn[3], bxn[3], nxa[3];
    double la = 0, lb = 0, le = 0;
    for (int j = 0; j < 3; ++j) {
        a[j] = (double)sh.p[0][j] - sh.p[1][j];
        b[j] = (double)sh.p[2][j] - sh.p[1][j];
        e[j] = (double)sh.p[3][j] - ((double)sh.p[0][j] + sh.p[2][j] - sh.p[1][j]);
        la += a[j] * a[j]; lb += b[j] * b[j]; le += e[j] * e[j];
    }
    auto cross3 = [](const double* x, const double* y, double* r) {
        r[0] = x[1] * y[2] - x[2] * y[1]; r[1] = x[2] * y[0] - x[0] * y[2]; r[2] = x[0] * y[1] - x[1] * y[0];
    };
    auto dot3 = [](const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
    cross3(a, b, nn);
    const double area2 = dot3(nn, nn);
    if (!(area2 > 1e-24) || !(std::sqrt(le) <= 1e-5 * (std::sqrt(la) + std::sqrt(lb)))) return;  // not a parallelogram: TK_QUAD
    // the stored normal must be the plane's normal (it is, by construction: 1256); otherwise keep the general path
    const double sn[3] = {sh.normal[0], sh.normal[1], sh.normal[2]};
    if (std::fabs(std::fabs(dot3(sn, nn)) / std::sqrt(area2) - 1.0) > 1e-4) return;
    cross3(b, nn, bxn);
    cross3(nn, a, nxa);
    const double ka = 1.0 / dot3(a, bxn), kb = 1.0 / dot3(b, nxa);
    const double p0[3] = {sh.p[0][0], sh.p[0][1], sh.p[0][2]}, p1[3] = {sh.p[1][0], sh.p[1][1], sh.p[1][2]};
    double as[3], bs[3];
    for (int j = 0; j < 3; ++j) { as[j] = bxn[j] * ka; bs[j] = nxa[j] * kb; }
    surf->kind = TK_PARALLELOGRAM;
    cp3(surf->f, sh.normal);
    surf->f[3] = (float)dot3(sn, p0);
    for (int j = 0; j < 3; ++j) { surf->f[4 + j] = (float)as[j]; surf->f[8 + j] = (float)bs[j]; }
    surf->f[7] = (float)(dot3(as, p1) + 0.5);
    surf->f[11] = (float)(dot3(bs, p1) + 0.5);
}

// A parallelogram (pack_shape has checked that) lying in an axis plane with its edges along the other two axes:
// returns that axis and the DAar record, or -1.  Exact comparisons on the caller's floats: nothing is snapped.
static int axis_aligned_rectangle(const ky_shape& sh, DAar* out) {
    for (int axis = 0; axis < 3; ++axis) {
        const float c = sh.p[0][axis];
        if (!(sh.p[1][axis] == c && sh.p[2][axis] == c && sh.p[3][axis] == c)) continue;
        const int u = (axis + 1) % 3, v = (axis + 2) % 3;
        // edges p1->p0 and p1->p2 must each run along one in-plane axis
        const bool a_u = sh.p[0][v] == sh.p[1][v] && sh.p[0][u] != sh.p[1][u];   // a = p0 - p1 along u
        const bool a_v = sh.p[0][u] == sh.p[1][u] && sh.p[0][v] != sh.p[1][v];
        const bool b_u = sh.p[2][v] == sh.p[1][v] && sh.p[2][u] != sh.p[1][u];
        const bool b_v = sh.p[2][u] == sh.p[1][u] && sh.p[2][v] != sh.p[1][v];
        if (!((a_u && b_v) || (a_v && b_u))) continue;
        double lo[2], hi[2];
        const int ax[2] = {u, v};
        for (int k = 0; k < 2; ++k) {
            lo[k] = hi[k] = sh.p[0][ax[k]];
            for (int q = 1; q < 4; ++q) { lo[k] = std::min(lo[k], (double)sh.p[q][ax[k]]); hi[k] = std::max(hi[k], (double)sh.p[q][ax[k]]); }
        }
        // the fourth corner must complete the rectangle exactly
        if (!((sh.p[3][u] == lo[0] || sh.p[3][u] == hi[0]) && (sh.p[3][v] == lo[1] || sh.p[3][v] == hi[1]))) continue;
        out->q0 = make_float4(c, (float)(0.5 * (lo[0] + hi[0])), (float)(0.5 * (hi[0] - lo[0])), (float)(0.5 * (lo[1] + hi[1])));
        out->q1 = make_float4((float)(0.5 * (hi[1] - lo[1])), 0.f, 0.f, 0.f);
        return axis;
    }
    return -1;
}

void pack_material(const ky_material& m, DMat* d) {
    std::memset(d, 0, sizeof *d);
    cp3(d->c0, m.color0); cp3(d->c1, m.color1);
    d->kind = m.kind; d->eta = m.eta; d->exponent = m.exponent; d->phong_pdf_norm = 0.f; d->p_specular = m.specular_probability;
    d->inv_eta = 1.f / m.eta;   // eta_i / eta_t entering the glass (fresnel_dielectric 1977, fresnel_specular 2388), in float like the reference
    if (m.kind == KY_MATERIAL_PLASTIC) {   // the two lobes' colours, plastic_material_t::scattering 2665 / 2667
        for (int j = 0; j < 3; ++j) { d->c0[j] = m.color0[j] / m.diffuse_probability; d->cs[j] = m.color1[j] / m.specular_probability; }
        // the Phong lobe's constants, in float like the reference computes them per call (2505, 2515, 2549); eta / inv_eta are glass-only
        const float inv_2pi = 0.15915494309189535f;
        d->eta = 1.f / (m.exponent + 1.f);
        d->inv_eta = (m.exponent + 2.f) * inv_2pi;
        d->phong_pdf_norm = (m.exponent + 1.f) * inv_2pi;
        // what a path's throughput is multiplied by (per unit |cos|) when it continues through the Phong lobe: value / pdf, the pow cancels
        for (int j = 0; j < 3; ++j) d->c1[j] = (d->cs[j] * d->inv_eta) / d->phong_pdf_norm;
    }
    const float e = m.exponent;
    const bool integral = std::isfinite(e) && std::fabs(e) < 16777216.f && std::floor(e) == e;
    d->exp_flags = (integral ? 1 : 0) | ((integral && std::fmod(std::fabs(e), 2.f) == 1.f) ? 2 : 0);
    // bits 2 / 3: c0 / c1 as packed above is not black (color_t::is_black, 258: every channel <= 0) -- the test of 4588 on a colour that is a constant of the material
    if (!(d->c0[0] <= 0 && d->c0[1] <= 0 && d->c0[2] <= 0)) d->exp_flags |= 4;
    if (!(d->c1[0] <= 0 && d->c1[1] <= 0 && d->c1[2] <= 0)) d->exp_flags |= 8;
    // bits 16-31: the upper half of a float F with pow(|x|, exponent) == 0 in the device's arithmetic for every |x| <= F (phong_pow_lobe): exp2(exponent x log2|x|)
    // underflows to zero once the product is below -150; F = 2^(-151 / exponent) leaves the hardware logarithm's last bits a margin, and cutting a positive
    // float's lower half off only lowers it.  0 (nothing is skipped) for exponents that are not positive and finite.
    if (m.kind == KY_MATERIAL_PLASTIC && std::isfinite(e) && e > 0.f) {
        const float F = std::exp2(-151.f / e);
        uint32_t bits;
        std::memcpy(&bits, &F, 4);
        if (F > 0.f && F < 1.f) d->exp_flags |= (int32_t)(bits & 0xffff0000u);
    }
}

// the stored normal of a disk / triangle / rectangle must be unit length (the reference's constructors normalise it:
// 1105, 1174, 1256); the device code relies on it
bool shape_normal_ok(const ky_shape& sh) {
    if (sh.kind == KY_SHAPE_SPHERE) return true;
    const double n2 = (double)sh.normal[0] * sh.normal[0] + (double)sh.normal[1] * sh.normal[1] + (double)sh.normal[2] * sh.normal[2];
    return std::fabs(n2 - 1.0) < 1e-4;
}

// Which surfaces a shadow ray never has to test (DScene::occ).
//
// The rays in question (scene_t::occluded 3187-3201 and the carrier query of by_bsdf) start at o = p + w, w = +-1e-2 n_p
// (offset_ray_origin, 614-620: along the normal of p's surface, on the side the ray leaves to), and run along dir = (q - p) / |q - p|
// to t = |q - p| - 2e-3: the segment from p to just short of q, SHIFTED by w.  It does not pass through q, and it can end up to 8e-3
// beyond q's depth (a reference quirk the tables must not hide: in the Cornell box most light samples taken from the floor are blocked
// by the lamp itself, and a few that miss the lamp's edge by the side panels above it).  p is a point of a surface with a non-delta
// material (4571), q a point of a light or (carrier query: the ray then ends exactly there) of a surface.
//
//  wall[X]      X is a planar rectangle, every surface / area light's shape / point light lies in ONE closed half-space of its plane,
//               and every non-delta surface that comes within |w| of the plane is planar and perpendicular or parallel to X (the shift
//               then keeps ray points on the scene's side, or moves them where the ray only leaves).  Such a ray has no point in X
//               when it ends on a scene point -> X is not in `occ`.
//  light_ok[l]  shadow rays towards samples of light l may use `occ` too: the shape (position) of l stays further than |w| from every
//               wall's plane, so the far end of such a ray cannot be shifted across one.
// Exact arithmetic on the caller's floats where a decision is an equality (the products of an axis-aligned plane are exact in double; a
// tilted wall whose neighbours' corners were rounded to the other side simply stays an occluder).  Indices are the caller's.
constexpr double K_HOST_RAY_OFFSET = 1e-2;   // offset_ray_origin 614-620
static void shape_extent(const ky_shape& sh, const double* n, double& lo, double& hi) {   // range of n.x over the shape
    auto dotp = [&](const float* p) { return n[0] * p[0] + n[1] * p[1] + n[2] * p[2]; };
    if (sh.kind == KY_SHAPE_SPHERE || sh.kind == KY_SHAPE_DISK) {   // a disk: bounded by its sphere
        const double c = dotp(sh.p[0]), r = (double)sh.radius * std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        lo = std::min(lo, c - r); hi = std::max(hi, c + r);
        return;
    }
    const int np = sh.kind == KY_SHAPE_TRIANGLE ? 3 : 4;
    for (int q = 0; q < np; ++q) { const double c = dotp(sh.p[q]); lo = std::min(lo, c); hi = std::max(hi, c); }
}
// find_boxes: groups of axis-aligned rectangles that are WHOLE faces of one axis-aligned box (exact float equality of every coordinate: the corners a scene
// builder copies from one table, 3299-3314 / 3336-3351).  A ray crosses the boundary of a convex box where it enters and where it leaves, so the nearest hit
// among a box's faces needs no test per face (box_update_nearest, ky_device.hpp).  A box is taken when
//   * at least KY_BOX_MIN_FACES of its six faces are surfaces (fewer: testing them one by one is cheaper), no face twice;
//   * no OTHER axis-aligned rectangle of the scene lies in the plane of one of those faces: a hit at exactly the same distance on two surfaces goes to the
//     earlier one in the reference's list (3177-3180), which the per-rectangle scan reproduces and a box cannot -- so a box never takes part in such a tie
//     (ties with parallelograms, spheres and general shapes have measure zero, as between the traversal's groups anyway).
// Rectangles are two-sided (1289), so faces count whichever way their stored normal points.
static bool aar_extent(const ky_shape& sh, int* axis, float lo[3], float hi[3]) {
    DAar unused;
    const int a = sh.kind == KY_SHAPE_RECTANGLE ? axis_aligned_rectangle(sh, &unused) : -1;
    if (a < 0) return false;
    for (int k = 0; k < 3; ++k) {
        lo[k] = hi[k] = sh.p[0][k];
        for (int q = 1; q < 4; ++q) { lo[k] = std::min(lo[k], sh.p[q][k]); hi[k] = std::max(hi[k], sh.p[q][k]); }
    }
    *axis = a;
    return true;
}
void find_boxes(const ky_scene* in, Boxes& B) {
    const int ns = in->surface_count;
    B.box.clear();
    B.box_of.assign(ns, -1);
    struct R { int axis; float lo[3], hi[3]; };
    std::vector<R> r(ns);
    std::vector<char> is_aar(ns, 0);
    for (int i = 0; i < ns; ++i) {
        const ky_surface& sf = in->surfaces[i];
        if (sf.shape < 0 || sf.shape >= in->shape_count) return;   // (pack_scene reports it)
        is_aar[i] = aar_extent(in->shapes[sf.shape], &r[i].axis, r[i].lo, r[i].hi) ? 1 : 0;
    }
    // a box's faces must have sorted surface indices below 15 (DBox), and the axis rectangles come first in the sorted order: a scene with many of them has at
    // most a few faces to offer -- and the search below is cubic in their number
    if (std::count(is_aar.begin(), is_aar.end(), (char)1) > 64) return;
    auto face_of = [&](const Boxes::Box& b, int i) {   // which face of b surface i is exactly, -1: none
        const R& q = r[i];
        const int a = q.axis;
        for (int k = 0; k < 3; ++k)
            if (k != a && !(q.lo[k] == b.lo[k] && q.hi[k] == b.hi[k])) return -1;
        if (q.lo[a] == b.lo[a]) return 2 * a;
        if (q.lo[a] == b.hi[a]) return 2 * a + 1;
        return -1;
    };
    for (int i = 0; i < ns && (int)B.box.size() < KY_MAX_BOXES; ++i) {
        if (!is_aar[i] || B.box_of[i] >= 0) continue;
        const int a = r[i].axis;
        bool taken = false;
        for (int j = 0; j < ns && !taken; ++j) {
            if (j == i || !is_aar[j] || B.box_of[j] >= 0 || r[j].axis == a) continue;
            const int b = r[j].axis, e = 3 - a - b;
            // rectangle j closes an edge of rectangle i: same extent along the third axis, each one's plane an end of the other's extent
            if (!(r[j].lo[e] == r[i].lo[e] && r[j].hi[e] == r[i].hi[e])) continue;
            if (!(r[j].lo[b] == r[i].lo[b] || r[j].lo[b] == r[i].hi[b])) continue;
            if (!(r[i].lo[a] == r[j].lo[a] || r[i].lo[a] == r[j].hi[a])) continue;
            Boxes::Box box{};
            box.lo[a] = r[j].lo[a]; box.hi[a] = r[j].hi[a];
            box.lo[b] = r[i].lo[b]; box.hi[b] = r[i].hi[b];
            box.lo[e] = r[i].lo[e]; box.hi[e] = r[i].hi[e];
            if (!(box.lo[0] < box.hi[0] && box.lo[1] < box.hi[1] && box.lo[2] < box.hi[2])) continue;
            for (int f = 0; f < 6; ++f) box.face[f] = -1;
            int n_faces = 0;
            bool ok = true;
            for (int k = 0; k < ns; ++k) {
                if (!is_aar[k] || B.box_of[k] >= 0) continue;
                const int f = face_of(box, k);
                if (f >= 0 && box.face[f] < 0) { box.face[f] = k; ++n_faces; }
            }
            for (int k = 0; k < ns && ok; ++k) {
                // any other rectangle in the plane of one of the box's FACES (a second copy of a face, a face of another box, a free rectangle): no box here
                // (the plane of an open side is nobody's: a lamp housing may reach the ceiling)
                if (!is_aar[k]) continue;
                const int ka = r[k].axis;
                for (int side = 0; side < 2; ++side) {
                    const int f = 2 * ka + side;
                    if (box.face[f] >= 0 && box.face[f] != k && r[k].lo[ka] == (side ? box.hi[ka] : box.lo[ka])) ok = false;
                }
            }
            //  the slab test multiplies coordinate differences by reciprocal directions clamped to +-1e30: beyond 3.4e8 the product overflows to inf,
            // whose face tag makes a signalling-NaN pattern.  A scene that large (camera included: ray origins) keeps its rectangles.
            for (int k3 = 0; k3 < 3; ++k3)
                if (!(std::fabs(box.lo[k3]) < 1e7f && std::fabs(box.hi[k3]) < 1e7f && std::fabs(in->camera.position[k3]) < 1e7f)) ok = false;
            if (!ok || n_faces < KY_BOX_MIN_FACES) continue;
            for (int f = 0; f < 6; ++f) if (box.face[f] >= 0) B.box_of[box.face[f]] = (int)B.box.size();
            B.box.push_back(box);
            taken = true;
        }
    }
}

void find_non_occluders(const ky_scene* in, NonOccluders& R) {
    const int ns = in->surface_count, nl = in->light_count;
    R.wall.assign(ns, 0);
    R.light_ok.assign(nl, 1);
    for (int l = 0; l < nl; ++l)
        if (in->lights[l].kind == KY_LIGHT_DIRECTION || in->lights[l].kind == KY_LIGHT_ENVIRONMENT) R.light_ok[l] = 0;   // their rays leave the scene
    const double inf = std::numeric_limits<double>::infinity();
    auto shape_of = [&](int i) -> const ky_shape& { return in->shapes[in->surfaces[i].shape]; };
    auto is_delta = [&](int i) { const int k = in->materials[in->surfaces[i].material].kind; return k == KY_MATERIAL_MIRROR || k == KY_MATERIAL_GLASS; };
    // Rays that start on surface y within the origin offset of the plane (unit normal n, offset k; `side` +1: the scene side is n.x >= k)
    // keep their origin on the scene side or leave moving away: y is planar and perpendicular or parallel to the plane.
    auto offset_safe = [&](int y, const double* n, double k, int side, double lo_y, double hi_y) {
        if (is_delta(y)) return true;                                   // no shadow ray starts on a delta surface (4571)
        const double nearest = side > 0 ? lo_y - k : k - hi_y;          // distance of y's nearest point from the plane
        if (nearest > 1.01 * K_HOST_RAY_OFFSET) return true;
        const ky_shape& sh = shape_of(y);
        if (sh.kind == KY_SHAPE_SPHERE) return false;
        if (sh.kind == KY_SHAPE_RECTANGLE) {   // its stored normal must be the normal of all four corners' plane
            double lo = inf, hi = -inf;
            const double m[3] = {sh.normal[0], sh.normal[1], sh.normal[2]};
            shape_extent(sh, m, lo, hi);
            if (hi - lo > 1e-6) return false;
        }
        const double c = std::fabs(n[0] * sh.normal[0] + n[1] * sh.normal[1] + n[2] * sh.normal[2]);
        return c <= 1e-7 || c >= 1.0 - 1e-12;
    };
    for (int i = 0; i < ns; ++i) {
        const ky_shape& sh = shape_of(i);
        if (sh.kind != KY_SHAPE_RECTANGLE) continue;
        // the plane through p1 spanned by the two edges (a quad that is not planar never gets a planar traversal record: pack_shape)
        double a[3], b[3], n[3];
        for (int j = 0; j < 3; ++j) { a[j] = (double)sh.p[0][j] - sh.p[1][j]; b[j] = (double)sh.p[2][j] - sh.p[1][j]; }
        n[0] = a[1] * b[2] - a[2] * b[1]; n[1] = a[2] * b[0] - a[0] * b[2]; n[2] = a[0] * b[1] - a[1] * b[0];
        const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(len > 0)) continue;
        // an axis-aligned plane keeps its exact form (n = +-e_axis after the division when the other two components are exact zeros)
        for (int j = 0; j < 3; ++j) n[j] /= len;
        const double k = n[0] * sh.p[1][0] + n[1] * sh.p[1][1] + n[2] * sh.p[1][2];
        double lo = inf, hi = -inf;
        for (int j = 0; j < ns; ++j) shape_extent(shape_of(j), n, lo, hi);
        for (int l = 0; l < nl; ++l) {
            const ky_light& L = in->lights[l];
            if (L.kind == KY_LIGHT_AREA) shape_extent(in->shapes[L.shape], n, lo, hi);
            if (L.kind == KY_LIGHT_POINT) { const double c = n[0] * L.position[0] + n[1] * L.position[1] + n[2] * L.position[2]; lo = std::min(lo, c); hi = std::max(hi, c); }
        }
        const int side = lo >= k ? 1 : (hi <= k ? -1 : 0);   // nothing strictly on the other side
        if (side == 0) continue;
        bool ok = true;
        for (int y = 0; y < ns && ok; ++y) {
            if (y == i) continue;
            double lo_y = inf, hi_y = -inf;
            shape_extent(shape_of(y), n, lo_y, hi_y);
            ok = offset_safe(y, n, k, side, lo_y, hi_y);
        }
        if (!ok) continue;
        R.wall[i] = 1;
        for (int l = 0; l < nl; ++l) {   // can the far end of a ray towards a sample of light l be shifted across this wall's plane?
            const ky_light& L = in->lights[l];
            if (L.kind == KY_LIGHT_DIRECTION || L.kind == KY_LIGHT_ENVIRONMENT) continue;
            double lo_l = inf, hi_l = -inf;
            if (L.kind == KY_LIGHT_AREA) shape_extent(in->shapes[L.shape], n, lo_l, hi_l);
            else lo_l = hi_l = n[0] * L.position[0] + n[1] * L.position[1] + n[2] * L.position[2];
            const double nearest = side > 0 ? lo_l - k : k - hi_l;
            if (!(nearest > 1.01 * K_HOST_RAY_OFFSET)) R.light_ok[l] = 0;
        }
    }
    R.deferred_ok = true;
    for (int l = 0; l < nl; ++l) R.deferred_ok = R.deferred_ok && R.light_ok[l];
    // Two-stage scan: the first planar area light that may use `occ` and has rectangles mounted behind its plane.  The sampled shape's
    // stored normal is the emitting side (2957-2960); k = the smallest n.q over the sampled points q.  "Entirely in n.x <= k" is decided
    // here in exact arithmetic; the device, which evaluates n.x for a ray's ends in fp32, gets k raised by a margin (pack_scene).
    R.ts_behind.assign(ns, 0);
    for (int l = 0; l < nl && R.ts_light < 0; ++l) {
        const ky_light& L = in->lights[l];
        if (L.kind != KY_LIGHT_AREA || !R.light_ok[l]) continue;
        const ky_shape& ls = in->shapes[L.shape];
        if (ls.kind == KY_SHAPE_SPHERE) continue;
        const double n[3] = {ls.normal[0], ls.normal[1], ls.normal[2]};
        double k_min = inf, k_max = -inf;
        if (ls.kind == KY_SHAPE_DISK) {
            k_min = k_max = n[0] * ls.p[0][0] + n[1] * ls.p[0][1] + n[2] * ls.p[0][2];
        } else if (ls.kind == KY_SHAPE_TRIANGLE) {
            shape_extent(ls, n, k_min, k_max);
        } else {   // rectangle_t samples p1 + (p0 - p1) u + (p2 - p1) v (1310): the parallelogram's fourth corner is p0 + p2 - p1
            const double c0 = n[0] * ls.p[0][0] + n[1] * ls.p[0][1] + n[2] * ls.p[0][2], c1 = n[0] * ls.p[1][0] + n[1] * ls.p[1][1] + n[2] * ls.p[1][2],
                         c2 = n[0] * ls.p[2][0] + n[1] * ls.p[2][1] + n[2] * ls.p[2][2];
            k_min = std::min(std::min(c0, c1), std::min(c2, c0 + c2 - c1));
        }
        int count = 0;
        std::vector<char> behind(ns, 0);
        for (int y = 0; y < ns; ++y) {
            const ky_shape& sh = shape_of(y);
            if (sh.kind != KY_SHAPE_RECTANGLE || R.wall[y] || in->surfaces[y].shape == L.shape) continue;
            double lo_y = inf, hi_y = -inf;
            shape_extent(sh, n, lo_y, hi_y);
            if (hi_y <= k_min) { behind[y] = 1; ++count; }
        }
        if (count == 0) continue;
        R.ts_light = l;
        R.ts_behind = behind;
        R.ts_plane[0] = n[0]; R.ts_plane[1] = n[1]; R.ts_plane[2] = n[2]; R.ts_plane[3] = k_min;
    }
}

// Specialised instantiations (scene facts KY_FEAT_*, and one kernel per direct-lighting strategy other than both_mis) can be switched
// off: KYHIP_SPECIALISE=0 or kyhip_set_specialisation(0).  The image
// does not depend on it (tests/test_configs_gpu.py); the switch exists for that test and for A/B measurements.
static int g_specialise = -1;
bool specialisation_enabled() {
    if (g_specialise < 0) {
        const char* e = std::getenv("KYHIP_SPECIALISE");
        g_specialise = (e && std::atoi(e) == 0) ? 0 : 1;
    }
    return g_specialise != 0;
}

// The boxes (find_boxes, KY_FEAT_BOXES) can be switched off on their own: KYHIP_BOXES=0 or kyhip_set_boxes(0).  A box's slab test computes a hit distance that
// differs from the per-rectangle test's by up to 15 units in the last place (box_update_nearest), so this switch -- unlike the one above -- moves an image beyond
// its last bit (tests/test_boxes.py measures by how much); it exists for that test, for the tests that compare two kernels bit for bit, and for A/B measurements.
static int g_boxes = -1;
bool boxes_enabled() {
    if (g_boxes < 0) {
        const char* e = std::getenv("KYHIP_BOXES");
        g_boxes = (e && std::atoi(e) == 0) ? 0 : 1;
    }
    return g_boxes != 0;
}

int pack_scene(const ky_scene* in, DScene* out) {
    if (!in) return fail(KY_ERR_INVALID_VALUE, "scene is NULL");
    if (in->surface_count < 0 || in->shape_count < 0 || in->material_count < 0 || in->light_count < 0)
        return fail(KY_ERR_INVALID_VALUE, "negative count in scene");
    if (in->surface_count > KYHIP_MAX_SURFACES || in->shape_count > KYHIP_MAX_SHAPES || in->material_count > KYHIP_MAX_MATERIALS ||
        in->light_count > KYHIP_MAX_LIGHTS)
        return fail(KY_ERR_LIMIT, "scene exceeds device limits (%d surfaces, %d shapes, %d materials, %d lights)", in->surface_count,
                    in->shape_count, in->material_count, in->light_count);
    if (in->environment_light < -1 || in->environment_light >= in->light_count) return fail(KY_ERR_INVALID_VALUE, "environment_light out of range");
    std::memset(out, 0, sizeof *out);
    out->n_surfaces = in->surface_count; out->n_lights = in->light_count; out->n_materials = in->material_count;
    out->env_light = in->environment_light;
    cp3(out->cam_position, in->camera.position); cp3(out->cam_front, in->camera.front); cp3(out->cam_right, in->camera.right);
    cp3(out->cam_up, in->camera.up);
    out->cam_inv_w = 1.f / in->camera.resolution[0]; out->cam_inv_h = 1.f / in->camera.resolution[1];
    // validate, build the traversal record of every surface, then lay the surfaces out sorted by traversal kind
    std::vector<DSurf> recs(in->surface_count);
    std::vector<DShapeFull> fulls(in->surface_count);
    for (int i = 0; i < in->surface_count; ++i) {
        const ky_surface& sf = in->surfaces[i];
        if (sf.shape < 0 || sf.shape >= in->shape_count || sf.material < 0 || sf.material >= in->material_count || sf.area_light < -1 ||
            sf.area_light >= in->light_count)
            return fail(KY_ERR_INVALID_VALUE, "surface %d has an index out of range", i);
        const ky_shape& sh = in->shapes[sf.shape];
        if (sh.kind < KY_SHAPE_DISK || sh.kind > KY_SHAPE_SPHERE) return fail(KY_ERR_INVALID_VALUE, "shape %d has an unknown kind", sf.shape);
        if (!shape_normal_ok(sh)) return fail(KY_ERR_INVALID_VALUE, "shape %d: the stored normal must be unit length", sf.shape);
        if (sf.area_light >= 0 && in->lights[sf.area_light].kind != KY_LIGHT_AREA)
            return fail(KY_ERR_INVALID_VALUE, "surface %d: area_light must refer to an area light", i);
        pack_shape(sh, 0, &recs[i], &fulls[i]);
    }
    for (int i = 0; i < in->light_count; ++i)   // checked again, with messages, where the lights are packed
        if (in->lights[i].kind == KY_LIGHT_AREA && (in->lights[i].shape < 0 || in->lights[i].shape >= in->shape_count))
            return fail(KY_ERR_INVALID_VALUE, "area light %d: shape out of range", i);
    for (int i = 0; i < in->surface_count; ++i)   // (checked again, with the other surface fields, below)
        if (in->surfaces[i].shape < 0 || in->surfaces[i].shape >= in->shape_count || in->surfaces[i].material < 0 || in->surfaces[i].material >= in->material_count)
            return fail(KY_ERR_INVALID_VALUE, "surface %d has an index out of range", i);
    NonOccluders non;
    find_non_occluders(in, non);
    int j = 0;
    struct PlanarEntry { int surface, axis; DAar aar; DPar par; };   // axis -1: a parallelogram record
    std::vector<PlanarEntry> planar;   // in traversal order: x, y, z planes, then the other parallelograms
    for (int pass = -3; pass < 3; ++pass) {   // -3, -2, -1: axis-aligned rectangles in the x, y, z planes
        for (int i = 0; i < in->surface_count; ++i) {
            const ky_surface& sf = in->surfaces[i];
            const ky_shape& sh = in->shapes[sf.shape];
            DAar aar{};
            const int axis = recs[i].kind == TK_PARALLELOGRAM ? axis_aligned_rectangle(sh, &aar) : -1;
            const int group = axis >= 0 ? axis - 3 : (recs[i].kind == TK_PARALLELOGRAM ? 0 : (recs[i].kind == TK_SPHERE ? 1 : 2));
            if (group != pass) continue;
            if (pass < 0) {
                const int32_t sorted_index = j;
                std::memcpy(&aar.q1.y, &sorted_index, 4);   // DAar: the scans note the surface from the record
                planar.push_back(PlanarEntry{i, axis, aar, DPar{}});
            } else if (pass == 0) {
                PlanarEntry e{i, -1, DAar{}, DPar{}};
                std::memcpy(&e.par.q0, &recs[i].f[0], 16); std::memcpy(&e.par.q1, &recs[i].f[4], 16); std::memcpy(&e.par.q2, &recs[i].f[8], 16);
                planar.push_back(e);
            } else if (pass == 1) {
                std::memcpy(&out->sph[out->n_sph++].c, &recs[i].f[0], 16);
                out->sph[out->n_sph] = out->sph[out->n_sph - 1];
            } else {
                DSurf& d = out->gen[out->n_gen++];
                d = recs[i];
                d.full = j;
            }
            out->full[j] = fulls[i];
            out->all[j] = recs[i];
            out->all[j].full = j;
            DHit& h = out->hit[j];
            cp3(h.n, sh.kind == KY_SHAPE_SPHERE ? sh.p[0] : sh.normal);
            h.kind = sh.kind; h.material = sf.material; h.area_light = sf.area_light;
            if (sh.kind != KY_SHAPE_SPHERE) host_frame(sh.normal, h.fs, h.ft);   // frame_t(normal), read by every vertex on this surface (ky_device.hpp, surface_frame)
            out->orig[j] = i;
            ++j;
        }
    }
    // the planar tables: every surface (trav: its order is the sorted surface order), and the occluder tables (DScene::occ, occ_front, occ_behind)
    auto build_trav = [&](DTrav& T, auto&& skip) {
        std::memset(&T, 0, sizeof T);
        for (const PlanarEntry& e : planar) {
            if (skip(e.surface)) continue;
            if (e.axis >= 0) { T.n_aar_axis[e.axis]++; T.aar[T.n_aar++] = e.aar; }
            else T.par[T.n_par++] = e.par;
        }
        if (T.n_aar > 0) T.aar[T.n_aar] = T.aar[T.n_aar - 1];   // one readable record past the end for the prefetch of i + 1
        if (T.n_par > 0) T.par[T.n_par] = T.par[T.n_par - 1];
    };
    build_trav(out->trav, [](int) { return false; });
    build_trav(out->occ, [&](int i) { return non.wall[i] != 0; });
    // KY_FEAT_BOXES: the boxes and the rectangles that are no box's face (DScene::boxtrav)
    std::memset(&out->boxtrav, 0, sizeof out->boxtrav);
    if (specialisation_enabled() && boxes_enabled()) {
        Boxes bx;
        find_boxes(in, bx);
        std::vector<int> sorted_of(in->surface_count, -1);
        for (int k = 0; k < j; ++k) sorted_of[out->orig[k]] = k;
        std::vector<char> boxed(in->surface_count, 0);
        DBoxTrav& BT = out->boxtrav;
        for (const Boxes::Box& b : bx.box) {
            uint32_t tag[6];
            bool fits = true;
            for (int f = 0; f < 6; ++f) {
                const int sidx = b.face[f] >= 0 ? sorted_of[b.face[f]] : KY_BOX_NO_FACE;
                fits = fits && sidx >= 0 && (b.face[f] < 0 || sidx < KY_BOX_NO_FACE);   // four bits per face
                tag[f] = (uint32_t)sidx;
            }
            if (!fits) continue;
            DBox& d = BT.box[BT.n_box++];
            d.q0 = make_float4(b.lo[0], b.lo[1], b.lo[2], 0.f);
            d.q1 = make_float4(b.hi[0], b.hi[1], b.hi[2], 0.f);
            std::memcpy(&d.q0.w, &tag[0], 4); std::memcpy(&d.q1.w, &tag[1], 4);
            std::memcpy(&d.q2.x, &tag[2], 4); std::memcpy(&d.q2.y, &tag[3], 4); std::memcpy(&d.q2.z, &tag[4], 4); std::memcpy(&d.q2.w, &tag[5], 4);
            for (int f = 0; f < 6; ++f) if (b.face[f] >= 0) boxed[b.face[f]] = 1;
        }
        for (const PlanarEntry& e : planar)
            if (e.axis >= 0 && !boxed[e.surface]) { BT.n_aar_axis[e.axis]++; BT.aar[BT.n_aar++] = e.aar; }
        if (BT.n_aar > 0) BT.aar[BT.n_aar] = BT.aar[BT.n_aar - 1];
    }
    out->occ_deferred_ok = non.deferred_ok ? 1 : 0;
    out->ts_light = non.ts_light;
    out->feat = 0;
    if (specialisation_enabled()) {   // the KY_FEAT_* facts of this scene
        if (in->light_count == 1 && in->lights[0].kind == KY_LIGHT_AREA && in->environment_light < 0) out->feat |= KY_FEAT_SINGLE_AREA;
        if (in->light_count == 1 && (in->lights[0].kind == KY_LIGHT_POINT || in->lights[0].kind == KY_LIGHT_DIRECTION) && in->environment_light < 0)
            out->feat |= KY_FEAT_SINGLE_DELTA;
        if (in->light_count == 1 && in->lights[0].kind == KY_LIGHT_ENVIRONMENT && in->environment_light == 0) out->feat |= KY_FEAT_SINGLE_ENV;
        bool rect = true;
        for (int i = 0; i < in->light_count; ++i)
            if (in->lights[i].kind == KY_LIGHT_AREA) rect = rect && in->shapes[in->lights[i].shape].kind == KY_SHAPE_RECTANGLE;
        if (rect) out->feat |= KY_FEAT_RECT_LIGHTS;
        bool spheres = in->light_count > 0 && in->environment_light < 0;   // KY_FEAT_SPHERE_LIGHTS; the carriers are checked below
        for (int i = 0; i < in->light_count; ++i)
            spheres = spheres && in->lights[i].kind == KY_LIGHT_AREA && in->shapes[in->lights[i].shape].kind == KY_SHAPE_SPHERE;
        for (int i = 0; i < in->surface_count; ++i)
            if (in->surfaces[i].area_light >= 0) spheres = spheres && in->shapes[in->surfaces[i].shape].kind == KY_SHAPE_SPHERE;
        if (spheres) out->feat |= KY_FEAT_SPHERE_LIGHTS;
        bool no_delta = true;
        for (int i = 0; i < in->material_count; ++i) no_delta = no_delta && in->materials[i].kind != KY_MATERIAL_MIRROR && in->materials[i].kind != KY_MATERIAL_GLASS;
        if (no_delta) out->feat |= KY_FEAT_NO_DELTA;
        if (in->surface_count <= KY_LDS_SURFACES_SMALL && in->material_count <= KY_LDS_MATERIALS_SMALL) out->feat |= KY_FEAT_SMALL_TABLES;
        if (out->boxtrav.n_box > 0) out->feat |= KY_FEAT_BOXES;
        bool flat_phong = true;
        for (int i = 0; i < in->surface_count; ++i)
            if (in->materials[in->surfaces[i].material].kind == KY_MATERIAL_PLASTIC && in->shapes[in->surfaces[i].shape].kind != KY_SHAPE_RECTANGLE) flat_phong = false;
        if (flat_phong) out->feat |= KY_FEAT_FLAT_PHONG;
        if (out->trav.n_aar > 0 && out->trav.n_par == 0 && out->n_gen == 0) out->feat |= KY_FEAT_AXIS_ALIGNED;   // (n_gen: the surfaces are laid out above)
        bool x_planks = out->trav.n_par > 0;   // KY_FEAT_X_PLANKS: the four zeros of every parallelogram record, exactly (a NaN is not a zero)
        for (int i = 0; i < out->trav.n_par; ++i) {
            const DPar& r = out->trav.par[i];
            if (!(r.q0.x == 0.f && r.q1.x == 0.f && r.q2.y == 0.f && r.q2.z == 0.f)) x_planks = false;
        }
        if (x_planks) out->feat |= KY_FEAT_X_PLANKS;
    }
    if (non.ts_light >= 0) {
        build_trav(out->occ_front, [&](int i) { return non.wall[i] != 0 || non.ts_behind[i] != 0; });
        build_trav(out->occ_behind, [&](int i) { return non.ts_behind[i] == 0; });
        // a point x of a surface behind the plane has n.x <= k in exact arithmetic; the device evaluates n.x for a ray's ends in fp32:
        // raise k by what that can be off (1e-5 of the scene's size is 100 ulp), so that a borderline end counts as "behind"
        double size = 0;
        for (int i = 0; i < in->surface_count; ++i)
            for (int q = 0; q < 4; ++q)
                for (int c = 0; c < 3; ++c) size = std::max(size, std::fabs((double)in->shapes[in->surfaces[i].shape].p[q][c]));
        for (int c = 0; c < 3; ++c) out->ts_plane[c] = (float)non.ts_plane[c];
        out->ts_plane[3] = (float)(non.ts_plane[3] + 1e-5 * (1.0 + size));
    }
    for (int i = 0; i < in->material_count; ++i) {
        const ky_material& m = in->materials[i];
        if (m.kind < KY_MATERIAL_MATTE || m.kind > KY_MATERIAL_PLASTIC) return fail(KY_ERR_INVALID_VALUE, "material %d has an unknown kind", i);
        pack_material(m, &out->mat[i]);
    }
    for (int i = 0; i < in->light_count; ++i) {
        const ky_light& l = in->lights[i];
        if (l.kind < KY_LIGHT_POINT || l.kind > KY_LIGHT_ENVIRONMENT) return fail(KY_ERR_INVALID_VALUE, "light %d has an unknown kind", i);
        DLight& d = out->light[i];
        cp3(d.color, l.color); cp3(d.position, l.position); cp3(d.direction, l.direction);
        //  the device decides "this light's colour is black" on the scalar unit from the floats' bits as signed integers (is_black_bits): a NaN with
        // the sign bit set would read as "<= 0", where color_t::is_black (258) says "not black" for every NaN.  A NaN channel is stored as the positive quiet NaN.
        for (int j = 0; j < 3; ++j)
            if (d.color[j] != d.color[j]) d.color[j] = std::numeric_limits<float>::quiet_NaN();
        d.kind = l.kind; d.world_radius = l.world_radius; d.shape_kind = -1;
        d.occ_ok = non.light_ok[i];
        d.shadow_table = (i == non.ts_light) ? ((int32_t)__builtin_offsetof(DScene, occ_front) | 1)
                                             : (int32_t)(non.light_ok[i] ? __builtin_offsetof(DScene, occ) : __builtin_offsetof(DScene, trav));
        if (l.kind == KY_LIGHT_AREA) {
            if (l.shape < 0 || l.shape >= in->shape_count) return fail(KY_ERR_INVALID_VALUE, "area light %d: shape out of range", i);
            const ky_shape& sh = in->shapes[l.shape];
            if (sh.kind < KY_SHAPE_DISK || sh.kind > KY_SHAPE_SPHERE) return fail(KY_ERR_INVALID_VALUE, "shape %d has an unknown kind", l.shape);
            if (!shape_normal_ok(sh)) return fail(KY_ERR_INVALID_VALUE, "shape %d: the stored normal must be unit length", l.shape);
            d.shape_kind = sh.kind; d.radius = sh.radius; d.area = host_shape_area(sh); d.inv_area = 1 / d.area;  // area_pdf = 1 / area(), 1313
            cp3(d.n, sh.normal);
            d.aar_axis = -1;
            if (sh.kind == KY_SHAPE_RECTANGLE) {
                DAar a{};
                d.aar_axis = axis_aligned_rectangle(sh, &a);
                if (d.aar_axis >= 0) { d.aar[0] = a.q0.x; d.aar[1] = a.q0.y; d.aar[2] = a.q0.z; d.aar[3] = a.q0.w; d.aar[4] = a.q1.x; }
            }
            if (sh.kind == KY_SHAPE_RECTANGLE) {  // p1 + (p0 - p1) u0 + (p2 - p1) u1, 1310
                cp3(d.p1, sh.p[1]);
                for (int j = 0; j < 3; ++j) { d.e0[j] = sh.p[0][j] - sh.p[1][j]; d.e1[j] = sh.p[2][j] - sh.p[1][j]; }
            } else if (sh.kind == KY_SHAPE_TRIANGLE) {
                cp3(d.p1, sh.p[0]); cp3(d.e0, sh.p[1]); cp3(d.e1, sh.p[2]);
            } else {
                cp3(d.p1, sh.p[0]);
                if (sh.kind == KY_SHAPE_SPHERE) d.e0[0] = 1.f / sh.radius;   // sphere lights: 1 / radius (the cone sampler's 1 / sin(theta_max) = distance / radius)
            }
            pack_shape(sh, KYHIP_MAX_SURFACES + i, &d.isect, &out->full[KYHIP_MAX_SURFACES + i]);
            if (d.isect.kind != TK_PARALLELOGRAM && d.isect.kind != TK_SPHERE) out->general = 1;   // a quad / triangle / disk light
            d.sampled_is_surface = 0;
            for (int j2 = 0; j2 < in->surface_count; ++j2) d.sampled_is_surface |= in->surfaces[j2].shape == l.shape;
            // the surfaces that carry this light (surface_t::area_light == &light, 3994), in sorted order
            d.n_carriers = 0;
            for (int j2 = 0; j2 < out->n_surfaces; ++j2) {
                if (out->hit[j2].area_light != i) continue;
                if (d.n_carriers >= 0 && d.n_carriers < KY_MAX_CARRIERS) d.carrier[d.n_carriers++] = j2;
                else d.n_carriers = -1;
            }
            d.pdf_from_carrier = (d.n_carriers == 1 && d.isect.kind == TK_PARALLELOGRAM && in->surfaces[out->orig[d.carrier[0]]].shape == l.shape) ? 1 : 0;
            // a sphere lamp's first carrier, if it is a sphere: its record (centre, radius^2) beside the light's own fields -- the BSDF-sampling estimators of the sphere-light
            // kernels test it from here (DLight::aar is free: the sampled shape is no rectangle) instead of through the carrier's index and the surface table, two dependent loads
            if (d.aar_axis < 0 && d.n_carriers >= 1 && out->all[d.carrier[0]].kind == TK_SPHERE) std::memcpy(d.aar, out->all[d.carrier[0]].f, 16);
        }
    }
    if (out->n_gen > 0) out->general = 1;
    if (specialisation_enabled() && !out->general) {   // KY_FEAT_CARRIERS needs the packed lights: carrier lists are built above
        bool carriers = true;
        for (int i = 0; i < in->light_count; ++i)
            if (in->lights[i].kind == KY_LIGHT_AREA && out->light[i].n_carriers < 0) carriers = false;
        if (carriers) out->feat |= KY_FEAT_CARRIERS;
        bool own = true, any_area = false;   // KY_FEAT_OWN_CARRIER: there are area lights, and each is its own (planar) carrier
        for (int i = 0; i < in->light_count; ++i)
            if (in->lights[i].kind == KY_LIGHT_AREA) { any_area = true; own = own && out->light[i].pdf_from_carrier != 0; }
        if (carriers && any_area && own) out->feat |= KY_FEAT_OWN_CARRIER;
    }
    return KY_OK;
}

// Deferred shadow rays (render_kernel<.., QUEUE>): which scenes get them.  They pay where most light samples die BEFORE the occlusion traversal and the few
// survivors of several lights fill one wavefront -- sphere lamps, whose samples hit the sampled sphere itself two times in three (quirk 1) -- and they cost
// where every sample needs its traversal anyway (rectangle lamps: no gain up to eight of them) or where a light has no BSDF-sampling half to share the
// vertex with (point / directional lights: 0.35 ms per light and 39 M samples slower than the inline shadow ray).  Measured on the round-4 kernels
// (tools/queue_policy.py, profiles/r04_h_queue_policy.txt): rooms with N sphere lamps cross over at N = 5; every point light moves the crossing by one; the
// shipped Veach scene (five sphere lamps) is 19 % faster deferred, the Cornell box with lamp and point light 22 % faster inline.  Rounds 2-3 deferred from two
// lights on -- right for the kernels of their time, wrong since the inline estimators accumulate in place.
// kyhip_set_shadow_queue(0 / 1) or the environment variable KYHIP_SHADOW_QUEUE = 0 / 1 switches them off / on for every scene (A/B measurements, tests).
#ifndef KY_SQ_MIN_SPHERE_MARGIN
#define KY_SQ_MIN_SPHERE_MARGIN 5   // sphere area lights minus delta lights
#endif
static int g_shadow_queue = -2;   // -1 by the scene, 0 never, 1 always
int shadow_queue_mode() {
    if (g_shadow_queue == -2) {
        const char* e = std::getenv("KYHIP_SHADOW_QUEUE");
        g_shadow_queue = e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
    }
    return g_shadow_queue;
}
bool shadow_queue_wanted(const ky_scene* scene) {
    const int mode = shadow_queue_mode();
    if (mode >= 0) return mode == 1 && scene->light_count > 0;
    int spheres = 0, deltas = 0;
    for (int i = 0; i < scene->light_count; ++i) {
        const ky_light& l = scene->lights[i];
        if (l.kind == KY_LIGHT_AREA && l.shape >= 0 && l.shape < scene->shape_count && scene->shapes[l.shape].kind == KY_SHAPE_SPHERE) ++spheres;
        if (l.kind == KY_LIGHT_POINT || l.kind == KY_LIGHT_DIRECTION) ++deltas;
    }
    return spheres - deltas >= KY_SQ_MIN_SPHERE_MARGIN;
}

// KYHIP_BLOCKS_PER_CU=n caps the resident workgroups per CU of the render kernels (shard-drain measurements, tools/shard_scan.py); read once
int blocks_per_cu_cap() {
    static const int cap = [] { const char* e = std::getenv("KYHIP_BLOCKS_PER_CU"); return e ? std::atoi(e) : 0; }();
    return cap;
}

// which render kernel runs path_tracing_iteration_t: the lane engine (render_kernel, default) or the queue engine (render_kernel_q)
static int g_engine = -1;
int current_engine() {
    if (g_engine < 0) {
        const char* e = std::getenv("KYHIP_ENGINE");
        g_engine = (e && (!std::strcmp(e, "queue") || !std::strcmp(e, "1"))) ? KY_ENGINE_QUEUE : KY_ENGINE_LANE;
    }
    return g_engine;
}

// 64 bits over the packed scene's words: the cache compares whole scenes only when these agree
uint64_t scene_hash(const DScene& s) {
    static_assert(sizeof(DScene) % 8 == 0, "hashed in 64-bit words");
    const uint64_t* w = reinterpret_cast<const uint64_t*>(&s);
    uint64_t h0 = 0x9E3779B97F4A7C15ull, h1 = 0xC2B2AE3D27D4EB4Full;
    for (size_t i = 0; i + 1 < sizeof(DScene) / 8; i += 2) {   // two independent multiply chains
        h0 = (h0 ^ w[i]) * 0xff51afd7ed558ccdull; h0 ^= h0 >> 29;
        h1 = (h1 ^ w[i + 1]) * 0xc4ceb9fe1a85ec53ull; h1 ^= h1 >> 31;
    }
    if ((sizeof(DScene) / 8) & 1) h0 = (h0 ^ w[sizeof(DScene) / 8 - 1]) * 0xff51afd7ed558ccdull;
    return h0 ^ (h1 * 0x9E3779B97F4A7C15ull);
}

// Everything pack_scene reads of the caller's scene, as one byte string (a few KB for ky's scenes), and 64 bits over it
bool scene_input(const ky_scene* in, std::vector<unsigned char>& out, uint64_t& hash) {
    out.clear();
    if (!in || in->surface_count < 0 || in->shape_count < 0 || in->material_count < 0 || in->light_count < 0 || in->surface_count > KYHIP_MAX_SURFACES ||
        in->shape_count > KYHIP_MAX_SHAPES || in->material_count > KYHIP_MAX_MATERIALS || in->light_count > KYHIP_MAX_LIGHTS)
        return false;   // pack_scene reports what is wrong
    auto put = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; out.insert(out.end(), b, b + n); };
    const int32_t head[6] = {in->shape_count, in->material_count, in->light_count, in->surface_count, in->environment_light, (specialisation_enabled() ? 1 : 0) | (boxes_enabled() ? 2 : 0)};
    put(head, sizeof head);
    put(&in->camera, sizeof in->camera);
    if (in->shape_count) put(in->shapes, sizeof(ky_shape) * (size_t)in->shape_count);
    if (in->material_count) put(in->materials, sizeof(ky_material) * (size_t)in->material_count);
    if (in->light_count) put(in->lights, sizeof(ky_light) * (size_t)in->light_count);
    if (in->surface_count) put(in->surfaces, sizeof(ky_surface) * (size_t)in->surface_count);
    out.resize((out.size() + 7) & ~(size_t)7, 0);
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < out.size(); i += 8) {
        uint64_t w;
        std::memcpy(&w, &out[i], 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 29;
    }
    hash = h;
    return true;
}

// CPUs this process may really use: the affinity mask, cut down to the cgroup's quota where there is one (a container granted 2 CPUs of a 256-thread host
// reports 256 from std::thread::hardware_concurrency(); four adding threads on two CPUs were slower than two)
int cpus_granted() {
    static const int n = [] {
        int cpus = (int)std::thread::hardware_concurrency();
        cpu_set_t mask;   // (hardware_concurrency() reports the machine's threads whatever the mask says: measured under taskset, round 5)
        if (sched_getaffinity(0, sizeof mask, &mask) == 0 && CPU_COUNT(&mask) > 0) cpus = CPU_COUNT(&mask);
        if (cpus < 1) cpus = 1;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char quota[32] = {0};
            long period = 0;
            if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
                const long q = (std::atol(quota) + period / 2) / period;
                if (q >= 1 && q < cpus) cpus = (int)q;
            }
            std::fclose(f);
        }
        return cpus;
    }();
    return n;
}

// Host threads of the seam's banded add (pageable films): the granted CPUs, at most four (tools/seam_trace.py: more bands / threads than that cost more in events
// and wake-ups than they add).  Measured under grants of 2 / 4 / 16 CPUs (tools/seam_cpu_scan.sh, profiles/r05_seam_cpu_scan.txt: configs[1] at 64 spp): with two
// CPUs two threads reach 0.90 of the device-resident rate, one 0.86, four 0.85 -- leaving a core to the runtime (VERDICT round 4's suggestion) is slower.
// KYHIP_SEAM_THREADS=n overrides (measurements).  A PINNED film needs none of this: the GPU adds to it in place (ky_seam.cpp).
int seam_threads() {
    static const int n = [] {
        if (const char* e = std::getenv("KYHIP_SEAM_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 16) return v; }
        const int cpus = cpus_granted();
        return std::max(1, std::min(cpus, 4));
    }();
    return n;
}

void host_add_rows(float* __restrict__ film, size_t stride_px, const float* __restrict__ src, int width, int y0, int y1) {
    const size_t n = (size_t)width * 3;
    for (int y = y0; y < y1; ++y) {
        float* __restrict__ dst = film + (size_t)y * stride_px * 3;
        const float* __restrict__ row = src + (size_t)y * n;
        for (size_t i = 0; i < n; ++i) dst[i] += row[i];   // vectorised by the host compiler
    }
}
void HostPool::run(int n, const std::function<void(int)>& fn) {
    std::lock_guard<std::mutex> one(job_m_);
    if (n <= 1) { if (n == 1) fn(0); return; }
    {
        std::lock_guard<std::mutex> lock(m_);
        if (pid_ != getpid()) { pid_ = getpid(); n_threads_ = 0; }   // a forked child inherits the object, not the (detached) threads
        while (n_threads_ < n - 1) { const int id = ++n_threads_; std::thread([this, id] { loop(id); }).detach(); }
        fn_ = &fn; n_ = n; pending_ = n - 1; ++generation_;
    }
    cv_.notify_all();
    fn(0);
    std::unique_lock<std::mutex> lock(m_);
    done_.wait(lock, [this] { return pending_ == 0; });
    fn_ = nullptr;
}
void HostPool::loop(int id) {
    unsigned long long seen = 0;
    for (;;) {
        const std::function<void(int)>* fn = nullptr;
        {
            std::unique_lock<std::mutex> lock(m_);
            cv_.wait(lock, [&] { return generation_ != seen && id < n_; });
            seen = generation_;
            fn = fn_;
        }
        (*fn)(id);
        std::lock_guard<std::mutex> lock(m_);
        if (--pending_ == 0) done_.notify_one();
    }
}
std::vector<std::unique_lock<std::mutex>> lock_seams(std::vector<std::pair<int, std::mutex*>> by_device) {
    std::sort(by_device.begin(), by_device.end(), [](const std::pair<int, std::mutex*>& a, const std::pair<int, std::mutex*>& b) { return a.first < b.first; });
    by_device.erase(std::unique(by_device.begin(), by_device.end(), [](const std::pair<int, std::mutex*>& a, const std::pair<int, std::mutex*>& b) { return a.first == b.first; }), by_device.end());
    std::vector<std::unique_lock<std::mutex>> locks;
    locks.reserve(by_device.size());
    for (auto& d : by_device) locks.emplace_back(*d.second);
    return locks;
}
HostPool& host_pool() { static HostPool* pool = new HostPool; return *pool; }   // never destroyed: no thread joins at process exit

// smallpt's argument check (kyhip_smallpt_render / kyhip_smallpt_kat_radiance)
int smallpt_check(const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p) {
    if (!spheres || !p) return fail(KY_ERR_INVALID_VALUE, "null argument");
    if (p->variant != KY_SP_VARIANT_SMALLPT && p->variant != KY_SP_VARIANT_REWRITE) return fail(KY_ERR_INVALID_VALUE, "unknown smallpt variant %d", p->variant);
    if (n <= 0 || n > KY_SP_MAX_SPHERES) return fail(KY_ERR_INVALID_VALUE, "1..%d spheres", KY_SP_MAX_SPHERES);
    if (p->width <= 0 || p->height <= 0 || p->width > 16384 || p->height > 16384 || p->samps <= 0 || p->max_depth < 0)
        return fail(KY_ERR_INVALID_VALUE, "invalid smallpt params");
    for (int i = 0; i < n; ++i)
        if (spheres[i].refl < KY_SP_DIFF || spheres[i].refl > KY_SP_REFR || !(spheres[i].rad > 0)) return fail(KY_ERR_INVALID_VALUE, "sphere %d is invalid", i);
    return KY_OK;
}

void set_engine_raw(int v) { g_engine = v; }
void set_specialise_raw(int v) { g_specialise = v; }
void set_boxes_raw(int v) { g_boxes = v; }
void set_shadow_queue_raw(int v) { g_shadow_queue = v; }
}  // namespace kyh

using namespace kyh;

extern "C" {

const char* kyhip_last_error(void) { return kyh::last_error().c_str(); }
int kyhip_seam_threads(void) { return kyh::seam_threads(); }
int kyhip_abi_version(void) { return KYHIP_ABI_VERSION; }


int kyhip_set_engine(int engine) {
    const int prev = current_engine();
    if (engine == KY_ENGINE_LANE || engine == KY_ENGINE_QUEUE) kyh::set_engine_raw(engine);
    return prev;
}
int kyhip_set_specialisation(int on) {
    const int prev = specialisation_enabled() ? 1 : 0;
    if (on == 0 || on == 1) kyh::set_specialise_raw(on);
    return prev;
}
int kyhip_set_boxes(int on) {
    const int prev = boxes_enabled() ? 1 : 0;
    if (on == 0 || on == 1) kyh::set_boxes_raw(on);
    return prev;
}
int kyhip_set_shadow_queue(int mode) {
    const int prev = shadow_queue_mode();
    if (mode >= -1 && mode <= 1) kyh::set_shadow_queue_raw(mode);
    return prev;
}

int64_t kyhip_shard_tile_count(const ky_render_params* p) {
    if (!valid_params(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params");
    return make_shard(p).n_tiles;
}
int64_t kyhip_shard_float_count(const ky_render_params* p) {
    if (!valid_params(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params");
    return (int64_t)make_shard(p).n_pix * 3;
}

size_t kyhip_workspace_bytes(const ky_render_params* p) {
    if (!valid_params(p)) return 0;
    return workspace_bytes_for(make_shard(p));
}


// host only: which surfaces the occluder tables leave out (find_non_occluders)
int kyhip_scene_non_occluders(const ky_scene* scene, int light, int* left_out, int n) {
    if (!scene || !left_out || n < 0) return fail(KY_ERR_INVALID_VALUE, "bad arguments");
    std::vector<DScene> packed(1);   // pack_scene validates the scene
    const int rc = pack_scene(scene, &packed[0]);
    if (rc != KY_OK) return rc;
    if (n < scene->surface_count) return fail(KY_ERR_INVALID_VALUE, "left_out holds %d entries, the scene has %d surfaces", n, scene->surface_count);
    if (light < -1 || light >= scene->light_count) return fail(KY_ERR_INVALID_VALUE, "light %d out of range", light);
    NonOccluders non;
    find_non_occluders(scene, non);
    const DScene& P = packed[0];
    int count = 0;
    for (int j = 0; j < P.n_surfaces; ++j) {
        const int i = P.orig[j];
        const bool planar = P.all[j].kind == TK_PARALLELOGRAM;   // only these have records in the planar tables
        left_out[i] = (planar && non.wall[i] && (light < 0 || non.light_ok[light])) ? 1 : 0;
        if (planar && light >= 0 && light == non.ts_light && non.ts_behind[i]) left_out[i] = 2;
        count += left_out[i] == 1;
    }
    const DTrav& T = (light < 0 || non.light_ok[light]) ? P.occ : P.trav;
    if (count != P.trav.n_aar + P.trav.n_par - T.n_aar - T.n_par) return fail(KY_ERR_DEVICE, "internal: occluder table and classification disagree");
    return count;
}

// host only: the KY_FEAT_* facts pack_scene finds for the scene (what a render-kernel instantiation may assume: ky_scene.hpp)
int kyhip_scene_facts(const ky_scene* scene) {
    if (!scene) return fail(KY_ERR_INVALID_VALUE, "bad arguments");
    std::vector<DScene> packed(1);
    const int rc = pack_scene(scene, &packed[0]);
    return rc != KY_OK ? rc : packed[0].feat;
}

// host only: the boxes the nearest-hit traversal tests whole (find_boxes): box_face[i] = 8 box + face (face = 2 axis + side) for a surface that is a box's face, -1 otherwise
int kyhip_scene_boxes(const ky_scene* scene, int* box_face, int n) {
    if (!scene || !box_face || n < 0) return fail(KY_ERR_INVALID_VALUE, "bad arguments");
    std::vector<DScene> packed(1);   // pack_scene validates the scene
    const int rc = pack_scene(scene, &packed[0]);
    if (rc != KY_OK) return rc;
    if (n < scene->surface_count) return fail(KY_ERR_INVALID_VALUE, "box_face holds %d entries, the scene has %d surfaces", n, scene->surface_count);
    for (int i = 0; i < scene->surface_count; ++i) box_face[i] = -1;
    const DScene& P = packed[0];
    for (int b = 0; b < P.boxtrav.n_box; ++b) {
        const DBox& d = P.boxtrav.box[b];
        const float* src[6] = {&d.q0.w, &d.q1.w, &d.q2.x, &d.q2.y, &d.q2.z, &d.q2.w};
        for (int f = 0; f < 6; ++f) {
            uint32_t sidx;
            std::memcpy(&sidx, src[f], 4);
            if (sidx != (uint32_t)KY_BOX_NO_FACE) box_face[P.orig[sidx]] = 8 * b + f;
        }
    }
    return P.boxtrav.n_box;
}

// ---- smallpt's scene in double precision (ky_smallpt.hpp) ----
int kyhip_smallpt_scene(ky_smallpt_sphere* out) {
    if (!out) return fail(KY_ERR_INVALID_VALUE, "null output");
    struct Row { double rad, p[3], e[3], c[3]; int refl; };
    static const Row rows[9] = {   // smallpt.cpp:42-52
        {1e5, {1e5 + 1, 40.8, 81.6}, {0, 0, 0}, {.75, .25, .25}, KY_SP_DIFF},     // Left
        {1e5, {-1e5 + 99, 40.8, 81.6}, {0, 0, 0}, {.25, .25, .75}, KY_SP_DIFF},   // Rght
        {1e5, {50, 40.8, 1e5}, {0, 0, 0}, {.75, .75, .75}, KY_SP_DIFF},           // Back
        {1e5, {50, 40.8, -1e5 + 170}, {0, 0, 0}, {0, 0, 0}, KY_SP_DIFF},          // Frnt
        {1e5, {50, 1e5, 81.6}, {0, 0, 0}, {.75, .75, .75}, KY_SP_DIFF},           // Botm
        {1e5, {50, -1e5 + 81.6, 81.6}, {0, 0, 0}, {.75, .75, .75}, KY_SP_DIFF},   // Top
        {16.5, {27, 16.5, 47}, {0, 0, 0}, {1 * .999, 1 * .999, 1 * .999}, KY_SP_SPEC},   // Mirr
        {16.5, {73, 16.5, 78}, {0, 0, 0}, {1 * .999, 1 * .999, 1 * .999}, KY_SP_REFR},   // Glas
        {600, {50, 681.6 - .27, 81.6}, {12, 12, 12}, {0, 0, 0}, KY_SP_DIFF}};     // Lite
    for (int i = 0; i < 9; ++i) {
        out[i].rad = rows[i].rad;
        for (int j = 0; j < 3; ++j) { out[i].p[j] = rows[i].p[j]; out[i].e[j] = rows[i].e[j]; out[i].c[j] = rows[i].c[j]; }
        out[i].refl = rows[i].refl;
        out[i].pad_ = 0;
    }
    return 9;
}

int kyhip_smallpt_scene_rewrite(ky_smallpt_sphere* out) {   // smallpt_rewrite.cpp:1201-1211, 1225-1242: z -> -z
    const int n = kyhip_smallpt_scene(out);
    if (n < 0) return n;
    static const double z[9] = {-81.6, -81.6, -1e5, 1e5 - 170, -81.6, -81.6, -47, -78, -81.6};
    for (int i = 0; i < n; ++i) out[i].p[2] = z[i];
    return n;
}

}  // extern "C"
