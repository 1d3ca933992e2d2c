/*
 * kyhip.hip -- kernels and C ABI of libkyhip.so (see include/kyhip.h).
 *
 * Kernel structure (DESIGN.md "Kernels"):
 *   render_kernel     persistent workgroups (the lane engine); each wavefront pulls work items (an 8x8 pixel
 *                     block x a chunk of samples) from a device counter; whichever lane is free takes the next
 *                     (item, pixel) pair and runs a flat state machine over path vertices, regenerating a new
 *                     camera sample the moment its path ends, so lanes never wait for each other and no path state
 *                     ever goes to HBM.  What is hot stays in registers (ray, position, normal, throughput, sampler);
 *                     what is merely alive (the lane's pixel chunk, the vertex's shading frame) lives in LDS, which
 *                     is what lets 6 wavefronts per SIMD be resident.  A finished chunk's pixel sum is added to a
 *                     64-bit fixed-point accumulator with integer atomics (order-independent => bit-identical
 *                     images for every tiling / GPU count).  Instantiations: <sampler, strategy fixed at compile time or -1,
 *                     QUEUE (deferred shadow rays on a per-wave stack, for multi-light scenes), GENERAL (scenes that hold
 *                     quads that are not parallelograms, triangles or disks)>; the host picks one per launch.
 *   render_kernel_q   (ky_queue.hpp) the wavefront formulation with the path pool and per-state queues in LDS;
 *                     experimental, off by default.
 *   smallpt_kernel    (ky_smallpt.hpp) smallpt's own scene and radiance() in double precision.
 *   resolve_kernel    fixed-point accumulator -> clamp01 -> fp32 tile buffer.
 *   film_add_kernel   film_t::add_color (ky.cpp:1586) for a shard's compact tile buffer;
 *   film_add_gathered_kernel  the same for all shards of a frame at once (after the multi-GPU gather).
 *   kat_*             function-level known-answer-test kernels.
 * gfx950 only; no CPU fallback anywhere in this file.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <map>
#include <sys/stat.h>
#include <unistd.h>
#include <cctype>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ky_render.hpp"
#include "ky_smallpt.hpp"

// ------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(KY_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// shard geometry (host + device)
// ------------------------------------------------------------------------------------------------
static bool valid_params(const ky_render_params* p) {
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
static bool shard_in_range(const ky_render_params* p) {
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

static ShardConst make_shard(const ky_render_params* p) {
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

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
#include "ky_queue.hpp"   // the queue engine: render_kernel_q

// fixed-point accumulator -> clamp01(L) (3726) -> fp32 tile buffer
__global__ void resolve_kernel(const unsigned long long* __restrict__ accum, const unsigned* __restrict__ flags, float* __restrict__ tiles, int n_floats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats) return;
    const int ch = i % 3;
    const unsigned fl = flags[i / 3];
    float v = (float)((double)(long long)accum[i] * (1.0 / KY_FIX_SCALE));
    const bool nan = (fl >> ch) & 1u, pinf = (fl >> (3 + ch)) & 1u, ninf = (fl >> (6 + ch)) & 1u;
    if (pinf) v = 1.f;
    if (ninf) v = 0.f;
    if (nan || (pinf && ninf)) v = 0.f;  // a NaN pixel: clamp01 keeps NaN in the reference and its 8-bit image shows 0
    tiles[i] = fminf(fmaxf(v, 0.f), 1.f);
}

__global__ void film_add_kernel(const float* __restrict__ tiles, float* __restrict__ film, size_t stride_px, ShardConst sh, int width, int height) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sh.n_pix) return;
    const int per_tile = sh.tile_w * sh.tile_h;
    const int k = i / per_tile, r = i % per_tile;
    const int tile = sh.tile_first + k * sh.tile_step;
    const int trow = tile / sh.tiles_x, tcol = (tile % sh.tiles_x + trow) % sh.tiles_x;   // rotated rows (kyhip.h)
    const int x = tcol * sh.tile_w + r % sh.tile_w;
    const int y = trow * sh.tile_h + r / sh.tile_w;
    if (x >= width || y >= height) return;
    float* px = film + ((size_t)y * stride_px + x) * 3;
    px[0] += tiles[3 * (size_t)i]; px[1] += tiles[3 * (size_t)i + 1]; px[2] += tiles[3 * (size_t)i + 2];
}

// film_t::add_color for ALL shards of a frame at once: `gathered` holds the compact tile buffers of the `world` shards
// (tile_first + r * tile_step, tile_step * world), r = 0 .. world - 1, shard r at gathered + r * rank_stride floats.
__global__ void film_add_gathered_kernel(const float* __restrict__ gathered, size_t rank_stride, int world, float* __restrict__ film, size_t stride_px,
                                         int tile_w, int tile_h, int tile_first, int tile_step, int tiles_x, int width, int height) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const int x = i % width, y = i / width;
    const int tcol = x / tile_w, trow = y / tile_h;
    const int tile = trow * tiles_x + ((tcol - trow % tiles_x) + tiles_x) % tiles_x;   // inverse of the row rotation (kyhip.h)
    const int rel = tile - tile_first;
    if (rel < 0 || rel % tile_step != 0) return;          // the tile does not belong to this frame's shard set
    const int j = rel / tile_step, r = j % world, k = j / world;
    const float* src = gathered + (size_t)r * rank_stride + (((size_t)k * tile_h + (y % tile_h)) * tile_w + (x % tile_w)) * 3;
    float* px = film + ((size_t)y * stride_px + x) * 3;
    px[0] += src[0]; px[1] += src[1]; px[2] += src[2];
}

// ---- KAT kernels ----
struct KatShape { DSurf surf; DShapeFull full; DHit hit; };

__global__ void kat_intersect_kernel(KatShape sh, const float* __restrict__ rays7, int n, float* __restrict__ out8) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays7 + 7 * (size_t)i;
    const f3 o = ld3(r), d = ld3(r + 3);
    float t;
    const bool hit = surf_hit(sh.surf, &sh.full, o, d, r[6], t);
    float* o8 = out8 + 8 * (size_t)i;
    f3 p = mk3(0, 0, 0), nn = mk3(0, 0, 0);
    if (hit) { p = o + t * d; nn = hit_normal(sh.hit, p, d); }
    o8[0] = hit ? 1.f : 0.f; o8[1] = hit ? t : 0.f;
    o8[2] = p.x; o8[3] = p.y; o8[4] = p.z; o8[5] = nn.x; o8[6] = nn.y; o8[7] = nn.z;
}

__global__ void kat_camera_kernel(const DScene* __restrict__ S, const float* __restrict__ pf, int n, float* __restrict__ out6) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 o, d;
    generate_ray(S, pf[2 * (size_t)i], pf[2 * (size_t)i + 1], o, d);
    float* q = out6 + 6 * (size_t)i;
    q[0] = o.x; q[1] = o.y; q[2] = o.z; q[3] = d.x; q[4] = d.y; q[5] = d.z;
}

__global__ void kat_bsdf_kernel(DMat M, const float* __restrict__ in12, int n, float* __restrict__ out13) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = in12 + 12 * (size_t)i;
    const f3 normal = ld3(r), wo = ld3(r + 3), wi_eval = ld3(r + 8);
    Vertex v;
    v.normal = normal;
    v.bsdf = make_bsdf(M, r[11]);
    const Bsdf& B = v.bsdf;
    vertex_prepare(v, wo);
    const BsdfSample bs = bsdf_sample(v, wo, r[6], r[7]);
    f3 ev; float pd, abs_cos_i;
    bsdf_eval_pdf(v, wo, wi_eval, ev, pd, abs_cos_i);
    float* q = out13 + 13 * (size_t)i;
    q[0] = bs.f.x; q[1] = bs.f.y; q[2] = bs.f.z; q[3] = bs.wi.x; q[4] = bs.wi.y; q[5] = bs.wi.z; q[6] = bs.pdf;
    q[7] = (float)bs.flags; q[8] = ev.x; q[9] = ev.y; q[10] = ev.z; q[11] = pd; q[12] = bsdf_is_delta(B) ? 1.f : 0.f;
}

__global__ void kat_light_kernel(const DScene* __restrict__ S, int li, const float* __restrict__ in11, int n, float* __restrict__ out11) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = in11 + 11 * (size_t)i;
    const f3 p = ld3(r), pn = ld3(r + 3), wi = ld3(r + 8);
    const LightSample ls = light_sample_Li(S->light[li], p, pn, r[6], r[7]);
    const float pdf = light_pdf_Li(S->light[li], S->full, p, pn, wi);
    float* q = out11 + 11 * (size_t)i;
    q[0] = ls.position.x; q[1] = ls.position.y; q[2] = ls.position.z; q[3] = ls.wi.x; q[4] = ls.wi.y; q[5] = ls.wi.z;
    q[6] = ls.pdf; q[7] = ls.Li.x; q[8] = ls.Li.y; q[9] = ls.Li.z; q[10] = pdf;
}

__global__ void kat_scene_intersect_kernel(const DScene* __restrict__ S, const float* __restrict__ rays7, int n, float* __restrict__ out9) {
    const LdsScene Lds = stage_scene<true>(S);   // KAT kernels: always the scene-sized dynamic block
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays7 + 7 * (size_t)i;
    const f3 o = ld3(r), d = ld3(r + 3);
    float t = r[6];
    const int hs = trace_nearest(S, o, d, t);
    f3 p = mk3(0, 0, 0), nn = mk3(0, 0, 0);
    if (hs >= 0) { p = o + t * d; nn = hit_normal(Lds.hit[hs], p, d); }
    float* q = out9 + 9 * (size_t)i;
    q[0] = hs >= 0 ? 1.f : 0.f; q[1] = hs >= 0 ? t : 0.f;
    q[2] = p.x; q[3] = p.y; q[4] = p.z; q[5] = nn.x; q[6] = nn.y; q[7] = nn.z; q[8] = hs >= 0 ? (float)S->orig[hs] : -1.f;
}

// table: -2 every surface, -1 DScene::occ, l >= 0 what by_emitter uses for light l
__global__ void kat_occluded_kernel(const DScene* __restrict__ S, const float* __restrict__ in9, int n, float* __restrict__ out1, int table) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = in9 + 9 * (size_t)i;
    const f3 p = ld3(r), pn = ld3(r + 3), target = ld3(r + 6);
    const f3 dir = normalize(target - p);
    const float dist = sqrtf(length_sq(p - target));
    const f3 o = offset_ray_origin(p, pn, dir);
    bool occ;
    if (table >= 0) occ = light_sample_occluded(S, table, o, dir, dist - 2e-3f);
    else occ = trace_any(S, table == -1 ? S->occ : S->trav, o, dir, dist - 2e-3f);
    out1[i] = occ ? 1.f : 0.f;
}

template <bool DEBUG_SAMPLER>
__global__ void kat_li_kernel(const DScene* __restrict__ S, RenderConst rc, int x, int y, int s0, int n, float* __restrict__ out3) {
    const LdsScene Lds = stage_scene<true>(S);   // KAT kernels: always the scene-sized dynamic block
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    PathState ps;
    bool alive = i < n;
    if (alive) path_begin<DEBUG_SAMPLER>(ps, S, sampler_pixel_key(rc.seed, (uint32_t)(y * rc.width + x)), x, y, s0 + i);
    while (__any(alive)) {  // path_shade is a wave-uniform call
        Vertex v;
        bool have_vertex = false;
        if (alive) have_vertex = path_intersect<DEBUG_SAMPLER>(ps, v, S, Lds, rc);
        const bool cont = path_shade<DEBUG_SAMPLER>(ps, v, S, Lds, rc, have_vertex);
        alive = have_vertex && cont;
    }
    if (i < n) { out3[3 * (size_t)i] = ps.Lo.x; out3[3 * (size_t)i + 1] = ps.Lo.y; out3[3 * (size_t)i + 2] = ps.Lo.z; }
}

// one light's direct-lighting estimate at given vertices with given random numbers (estimate_direct_lighting_*, 3889-4088)
__global__ void kat_nee_kernel(const DScene* __restrict__ S, int strategy, int li, const float* __restrict__ in15, int n, float* __restrict__ out6) {
    const LdsScene Lds = stage_scene<true>(S);   // KAT kernels: always the scene-sized dynamic block
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    const float* r = in15 + 15 * (size_t)(active ? i : 0);
    Vertex v;
    v.position = ld3(r); v.normal = ld3(r + 3);
    v.t = 0.f;
    v.surface = 0;
    for (int j = 0; j < S->n_surfaces; ++j)
        if (S->orig[j] == (int)r[9]) v.surface = j;          // the caller's surface index -> the device's sorted index
    v.bsdf = make_bsdf(Lds.mat[Lds.hit[v.surface].material], r[10]);
    const f3 wo = ld3(r + 6);
    vertex_prepare(v, wo);
    f3 Lb = mk3(0, 0, 0), Ll = mk3(0, 0, 0);
    const bool nee = active && !bsdf_is_delta(v.bsdf);        // sample_all_light runs for non-delta vertices only (4571)
    // (wave-uniform calls: every lane makes them, `nee` says whether it takes part)
    const f3 one = mk3(1, 1, 1);   // the estimators ADD w x their estimate to an accumulator
    if (strategy == KY_DIRECT_BSDF) estimate_by_bsdf<false>(S, Lds, v, wo, li, r[11], r[12], nee, Lb, one);
    else if (strategy == KY_DIRECT_BSDF_MIS || strategy == KY_DIRECT_BOTH_MIS) estimate_by_bsdf<true>(S, Lds, v, wo, li, r[11], r[12], nee, Lb, one);
    if (nee) {
        if (strategy == KY_DIRECT_LIGHT) estimate_by_emitter<false>(S, Lds, v, wo, li, r[13], r[14], Ll, one);
        else if (strategy == KY_DIRECT_LIGHT_MIS || strategy == KY_DIRECT_BOTH_MIS) estimate_by_emitter<true>(S, Lds, v, wo, li, r[13], r[14], Ll, one);
    }
    if (active) {
        float* o = out6 + 6 * (size_t)i;
        o[0] = Lb.x; o[1] = Lb.y; o[2] = Lb.z; o[3] = Ll.x; o[4] = Ll.y; o[5] = Ll.z;
    }
}

// one camera sample, traced vertex by vertex (lane 0 walks the path; the other lanes only keep the wave-uniform calls company)
template <bool DEBUG_SAMPLER>
__global__ void kat_li_trace_kernel(const DScene* __restrict__ S, RenderConst rc, int x, int y, int s, int max_rows, float* __restrict__ out) {
    const LdsScene Lds = stage_scene<true>(S);   // KAT kernels: always the scene-sized dynamic block
    PathState ps;
    bool alive = threadIdx.x == 0;
    VertexTrace tr{out + 4, max_rows, 0};
    if (alive) path_begin<DEBUG_SAMPLER>(ps, S, sampler_pixel_key(rc.seed, (uint32_t)(y * rc.width + x)), x, y, s);
    while (__any(alive)) {
        Vertex v;
        bool have_vertex = false;
        if (alive) have_vertex = path_intersect<DEBUG_SAMPLER>(ps, v, S, Lds, rc);
        const bool cont = path_shade<DEBUG_SAMPLER>(ps, v, S, Lds, rc, have_vertex, -1, &tr);
        alive = have_vertex && cont;
    }
    if (threadIdx.x == 0) { out[0] = (float)tr.n; out[1] = ps.Lo.x; out[2] = ps.Lo.y; out[3] = ps.Lo.z; }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static void cp3(float* d, const float* s) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
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
static void pack_shape(const ky_shape& sh, int full_index, DSurf* surf, DShapeFull* full) {
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

static void pack_material(const ky_material& m, DMat* d) {
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
}

// the stored normal of a disk / triangle / rectangle must be unit length (the reference's constructors normalise it:
// 1105, 1174, 1256); the device code relies on it
static bool shape_normal_ok(const ky_shape& sh) {
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
struct NonOccluders {
    std::vector<char> wall;                  // [surface]
    std::vector<char> light_ok;              // [light]
    bool deferred_ok = false;                // all lights ok (the deferred shadow rays share one stack)
    int ts_light = -1;                       // two-stage scan (DScene::occ_front / occ_behind): the light, its plane n.x = k, and
    double ts_plane[4] = {0, 0, 0, 0};       // the surfaces that lie entirely in n.x <= k (not the light's own)
    std::vector<char> ts_behind;             // [surface]
};
static void find_non_occluders(const ky_scene* in, NonOccluders& R) {
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
static bool specialisation_enabled() {
    if (g_specialise < 0) {
        const char* e = std::getenv("KYHIP_SPECIALISE");
        g_specialise = (e && std::atoi(e) == 0) ? 0 : 1;
    }
    return g_specialise != 0;
}

static int pack_scene(const ky_scene* in, DScene* out) {
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
            DAar aar;
            const int axis = recs[i].kind == TK_PARALLELOGRAM ? axis_aligned_rectangle(sh, &aar) : -1;
            const int group = axis >= 0 ? axis - 3 : (recs[i].kind == TK_PARALLELOGRAM ? 0 : (recs[i].kind == TK_SPHERE ? 1 : 2));
            if (group != pass) continue;
            if (pass < 0) {
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
        d.kind = l.kind; d.world_radius = l.world_radius; d.shape_kind = -1;
        d.occ_ok = non.light_ok[i];
        if (l.kind == KY_LIGHT_AREA) {
            if (l.shape < 0 || l.shape >= in->shape_count) return fail(KY_ERR_INVALID_VALUE, "area light %d: shape out of range", i);
            const ky_shape& sh = in->shapes[l.shape];
            if (sh.kind < KY_SHAPE_DISK || sh.kind > KY_SHAPE_SPHERE) return fail(KY_ERR_INVALID_VALUE, "shape %d has an unknown kind", l.shape);
            if (!shape_normal_ok(sh)) return fail(KY_ERR_INVALID_VALUE, "shape %d: the stored normal must be unit length", l.shape);
            d.shape_kind = sh.kind; d.radius = sh.radius; d.area = host_shape_area(sh); d.inv_area = 1 / d.area;  // area_pdf = 1 / area(), 1313
            cp3(d.n, sh.normal);
            if (sh.kind == KY_SHAPE_RECTANGLE) {  // p1 + (p0 - p1) u0 + (p2 - p1) u1, 1310
                cp3(d.p1, sh.p[1]);
                for (int j = 0; j < 3; ++j) { d.e0[j] = sh.p[0][j] - sh.p[1][j]; d.e1[j] = sh.p[2][j] - sh.p[1][j]; }
            } else if (sh.kind == KY_SHAPE_TRIANGLE) {
                cp3(d.p1, sh.p[0]); cp3(d.e0, sh.p[1]); cp3(d.e1, sh.p[2]);
            } else {
                cp3(d.p1, sh.p[0]);
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
        }
    }
    if (out->n_gen > 0) out->general = 1;
    if (specialisation_enabled() && !out->general) {   // KY_FEAT_CARRIERS needs the packed lights: carrier lists are built above
        bool carriers = true;
        for (int i = 0; i < in->light_count; ++i)
            if (in->lights[i].kind == KY_LIGHT_AREA && out->light[i].n_carriers < 0) carriers = false;
        if (carriers) out->feat |= KY_FEAT_CARRIERS;
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
static int shadow_queue_mode() {
    if (g_shadow_queue == -2) {
        const char* e = std::getenv("KYHIP_SHADOW_QUEUE");
        g_shadow_queue = e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
    }
    return g_shadow_queue;
}
static bool shadow_queue_wanted(const ky_scene* scene) {
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
static int blocks_per_cu_cap() {
    static const int cap = [] { const char* e = std::getenv("KYHIP_BLOCKS_PER_CU"); return e ? std::atoi(e) : 0; }();
    return cap;
}

// which render kernel runs path_tracing_iteration_t: the lane engine (render_kernel, default) or the queue engine (render_kernel_q)
enum { KY_ENGINE_LANE = 0, KY_ENGINE_QUEUE = 1 };
static int g_engine = -1;
static int current_engine() {
    if (g_engine < 0) {
        const char* e = std::getenv("KYHIP_ENGINE");
        g_engine = (e && (!std::strcmp(e, "queue") || !std::strcmp(e, "1"))) ? KY_ENGINE_QUEUE : KY_ENGINE_LANE;
    }
    return g_engine;
}

// device memory that is released on every way out of an entry point
struct DevBuf {
    void* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes < 16 ? 16 : bytes); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// One context per device, created on first use.  Every entry point holds the context's own mutex while it enqueues, so
// calls for different devices never wait for each other.
//
// What a launch writes -- work counter, accumulator workspace, timing events, the wavefronts' shadow-ray stacks -- belongs to the
// STREAM it is enqueued on (StreamState): calls on one stream execute in stream order anyway, and calls on different streams share
// nothing, so a frame's kernel can start on the compute units the previous frame's kernel is draining from (ky_amd/dist.py alternates
// two streams: a persistent kernel pays its start-up and its tail once per launch, and only another launch can fill them).
// What a launch only reads -- the packed scene -- is cached by CONTENT (SceneSlot): a workload that alternates between a few scenes
// (render_multiple_scene, ky.cpp:4819-4876) uploads each once and never synchronises the device again.
struct StreamState {
    bool used = false;
    hipStream_t stream = nullptr;
    unsigned* d_counter = nullptr;
    void* ws = nullptr;
    size_t ws_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing_valid = false;
    float4* d_shadow_queue = nullptr;      // QUEUE instantiations: the wavefronts' stacks, allocated on first use ...
    size_t sq_blocks = 0;                  // ... for this many workgroups; the QUEUE variants differ in occupancy (5 or 6 per CU), so a launch that needs more reallocates
    hipEvent_t done = nullptr;             // behind the last kernel this library enqueued on the stream (what a hand-over of this state or of a scene slot waits for)
    unsigned long long last_use = 0;
};
struct SceneSlot {
    bool valid = false;
    DScene* d = nullptr;                   // device copy
    DScene* h = nullptr;                   // pinned staging copy = what `d` holds (the key of the cache, compared when the hashes agree)
    uint64_t hash = 0;                     // scene_hash(*h)
    std::vector<unsigned char> input;      // the caller's scene this slot was last packed from, flattened (scene_input): a call that passes the same scene
    uint64_t input_hash = 0;               // again skips pack_scene -- 60-80 us of host time on a frame shard that renders in 6 ms
    unsigned readers = 0;                  // bit i: a launch on stream state i has read `d` (its StreamState::done covers that launch)
    hipEvent_t ready = nullptr;            // the upload; launches on other streams than the uploading one wait for it (device side)
    hipStream_t upload_stream = nullptr;
    unsigned long long last_use = 0;
};
constexpr int KY_STREAM_STATES = 8, KY_SCENE_SLOTS = 8;
// what kyhip_render / kyhip_render_multi keep between calls (the host-film seam); `m` serialises such calls per device, it is never
// taken while a context's enqueue mutex is held
// (measured on configs[1]'s 9.4 MB film, tools/seam_trace.py: what a call costs beyond its kernel -- the film's pinned download alone is 0.18 ms -- is
// 0.62 ms with one download and one adding thread; 0.44 with eight bands dealt to four threads; with every thread adding its slice of every band
// 0.39-0.42 (one band), 0.34-0.36 (two), 0.33-0.36 (four), 0.40 (eight): each band is a copy command and an event)
constexpr int KY_SEAM_BANDS = 2, KY_SEAM_THREADS = 4;
struct SeamBuffers {
    std::mutex m;
    void* d_gather = nullptr; size_t gather_bytes = 0;   // root: [n_devices][shard 0's tile buffer]
    void* d_film = nullptr; size_t film_bytes = 0;       // root: the frame, de-interleaved
    float* h_stage = nullptr; size_t stage_bytes = 0;    // root: pinned host copy of d_film
    hipEvent_t band[KY_SEAM_BANDS] = {};                 // root: behind the download of each row band
    std::vector<void*> d_remote;                         // this device as a non-root member of a list: one tile buffer per occurrence
    std::vector<size_t> remote_bytes;
};
struct DeviceCtx {
    std::mutex m;
    SeamBuffers seam;
    int device = 0;
    int cus = 0;
    StreamState ss[KY_STREAM_STATES];
    SceneSlot scenes[KY_SCENE_SLOTS];
    unsigned long long clock = 0;
    StreamState* last_launch = nullptr;    // kyhip_kernel_ms reads its event pair
    hipStream_t stream = nullptr;   // the library's own stream on this device (kyhip_render_multi)
    int variant_blocks[48] = {};           // resident workgroups per CU of g_variants[i] (0: not asked yet) ...
    size_t variant_lds[48] = {};           // ... for a scene block of this many bytes
    int last_variant = -1;
    int q_blocks_per_cu[3] = {0, 0, 0};
    struct JitKernel { hipModule_t module = nullptr; hipFunction_t fn = nullptr; int per_cu = 0; size_t lds = ~(size_t)0; bool failed = false; };
    std::map<std::string, JitKernel> jit;   // run-time instantiations loaded on this device, by template arguments
    std::string last_jit;                   // ... and the one the last launch used (last_variant == -3)
};
static std::mutex g_ctx_mutex;                          // guards g_ctx itself (creation), never held while enqueueing
static std::vector<std::unique_ptr<DeviceCtx>> g_ctx;   // index = HIP device ordinal

static int create_ctx(int device, DeviceCtx& c) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(KY_ERR_NO_DEVICE, "device %d is %s; libkyhip is built for gfx950 only", device, prop.gcnArchName);
    c.device = device;
    c.cus = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c.q_blocks_per_cu[0], render_kernel_q<false, KY_DIRECT_BOTH_MIS>, QE_THREADS, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c.q_blocks_per_cu[1], render_kernel_q<false, -1>, QE_THREADS, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c.q_blocks_per_cu[2], render_kernel_q<true, -1>, QE_THREADS, 0));
    return KY_OK;
}

// Looks the context of `device` up (creating it on first use) and makes the device current for the calling thread.
static int get_ctx(int device, DeviceCtx** out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(KY_ERR_NO_DEVICE, "no HIP device visible (libkyhip has no CPU fallback)");
    if (device < 0 || device >= n) return fail(KY_ERR_INVALID_VALUE, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if ((int)g_ctx.size() < n) g_ctx.resize(n);
    if (!g_ctx[device]) {
        auto c = std::make_unique<DeviceCtx>();
        const int rc = create_ctx(device, *c);
        if (rc != KY_OK) return rc;   // a half-built context is dropped; its few allocations are reclaimed at process exit
        g_ctx[device] = std::move(c);
    }
    *out = g_ctx[device].get();
    return KY_OK;
}
static DeviceCtx* find_ctx(int device) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    return (device >= 0 && device < (int)g_ctx.size()) ? g_ctx[device].get() : nullptr;
}

// The launch state of `stream` on this device (created on first use; with more than KY_STREAM_STATES streams in use the least
// recently used state is handed over, after the device has drained).
static int get_stream_state(DeviceCtx* c, hipStream_t stream, StreamState** out) {
    StreamState* pick = nullptr;
    for (StreamState& st : c->ss)
        if (st.used && st.stream == stream) pick = &st;
    if (!pick) {
        for (StreamState& st : c->ss)
            if (!st.used && !pick) pick = &st;
        if (!pick) {
            pick = &c->ss[0];
            for (StreamState& st : c->ss)
                if (st.last_use < pick->last_use) pick = &st;
            HIP_TRY(hipEventSynchronize(pick->done));   // its buffers may still be in use on the stream that owned them (only that stream is waited for)
            pick->timing_valid = false;
        }
        if (!pick->d_counter) {
            HIP_TRY(hipMalloc(&pick->d_counter, 256));
            HIP_TRY(hipEventCreate(&pick->ev0));
            HIP_TRY(hipEventCreate(&pick->ev1));
            HIP_TRY(hipEventCreateWithFlags(&pick->done, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(pick->done, stream));
        }
        pick->used = true;
        pick->stream = stream;
    }
    pick->last_use = ++c->clock;
    *out = pick;
    return KY_OK;
}

// 64 bits over the packed scene's words: the cache compares whole scenes only when these agree
static uint64_t scene_hash(const DScene& s) {
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
static bool scene_input(const ky_scene* in, std::vector<unsigned char>& out, uint64_t& hash) {
    out.clear();
    if (!in || in->surface_count < 0 || in->shape_count < 0 || in->material_count < 0 || in->light_count < 0 || in->surface_count > KYHIP_MAX_SURFACES ||
        in->shape_count > KYHIP_MAX_SHAPES || in->material_count > KYHIP_MAX_MATERIALS || in->light_count > KYHIP_MAX_LIGHTS)
        return false;   // pack_scene reports what is wrong
    auto put = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; out.insert(out.end(), b, b + n); };
    const int32_t head[6] = {in->shape_count, in->material_count, in->light_count, in->surface_count, in->environment_light, specialisation_enabled() ? 1 : 0};
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

// The device copy of `scene`, from the cache or uploaded on `stream`; launches on `stream` may read it when this returns.
static int upload_scene(DeviceCtx* c, const ky_scene* scene, hipStream_t stream, SceneSlot** out) {
    static thread_local DScene scratch;
    static thread_local std::vector<unsigned char> input;
    uint64_t input_hash = 0;
    const bool keyed = scene_input(scene, input, input_hash);
    SceneSlot* pick = nullptr;
    if (keyed)   // the same scene as a recent call's: its packed form is on the device already
        for (SceneSlot& sl : c->scenes)
            if (sl.valid && sl.input_hash == input_hash && sl.input == input) pick = &sl;
    uint64_t hash = 0;
    if (!pick) {
        const int rc = pack_scene(scene, &scratch);
        if (rc != KY_OK) return rc;
        hash = scene_hash(scratch);
        for (SceneSlot& sl : c->scenes)
            if (sl.valid && sl.hash == hash && std::memcmp(&scratch, sl.h, sizeof(DScene)) == 0) pick = &sl;
        if (pick && keyed) { pick->input = input; pick->input_hash = input_hash; }
    }
    if (pick) {
        if (pick->upload_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, pick->ready, 0));
    } else {
        for (SceneSlot& sl : c->scenes)
            if (!sl.valid && !pick) pick = &sl;
        if (!pick) {   // every slot holds a scene: replace the least recently used one, which launches in flight may still read
            pick = &c->scenes[0];
            for (SceneSlot& sl : c->scenes)
                if (sl.last_use < pick->last_use) pick = &sl;
            // only the streams that have launched on this copy are waited for, not the device (a caller's other streams keep running)
            for (int i = 0; i < KY_STREAM_STATES; ++i)
                if ((pick->readers >> i & 1u) && c->ss[i].done) HIP_TRY(hipEventSynchronize(c->ss[i].done));
            HIP_TRY(hipEventSynchronize(pick->ready));
            pick->valid = false;
        }
        if (!pick->d) {
            HIP_TRY(hipMalloc(&pick->d, sizeof(DScene)));
            HIP_TRY(hipHostMalloc(&pick->h, sizeof(DScene)));
            HIP_TRY(hipEventCreateWithFlags(&pick->ready, hipEventDisableTiming));
        }
        std::memcpy(pick->h, &scratch, sizeof(DScene));
        HIP_TRY(hipMemcpyAsync(pick->d, pick->h, sizeof(DScene), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(pick->ready, stream));
        pick->upload_stream = stream;
        pick->hash = hash;
        if (keyed) { pick->input = input; pick->input_hash = input_hash; } else { pick->input.clear(); pick->input_hash = 0; }
        pick->readers = 0;
        pick->valid = true;
    }
    pick->last_use = ++c->clock;
    *out = pick;
    return KY_OK;
}

static RenderConst make_rc(const ky_render_params* p) {
    RenderConst rc{};
    rc.integrator = p->integrator; rc.max_path_depth = p->max_path_depth; rc.strategy = p->direct_sample; rc.seed = p->seed;
    rc.width = p->width; rc.height = p->height; rc.spp = p->samples_per_pixel;
    rc.inv_spp = (float)(1. / p->samples_per_pixel);  // ky.cpp:3717
    return rc;
}


// shared driver of the KAT entry points
template <typename F>
static int kat_run(int device, const void* in, size_t in_bytes, void* out, size_t out_bytes, F launch) {
    DeviceCtx* c;
    int rcode = get_ctx(device, &c);
    if (rcode != KY_OK) return rcode;
    std::lock_guard<std::mutex> lock(c->m);
    DevBuf d_in, d_out;
    HIP_TRY(d_in.alloc(in_bytes));
    HIP_TRY(d_out.alloc(out_bytes));
    HIP_TRY(hipDeviceSynchronize());   // KAT entries use the default stream and may replace the device's scene copy
    HIP_TRY(hipMemcpy(d_in.p, in, in_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_out.p, 0, out_bytes));
    rcode = launch(c, d_in.as<const float>(), d_out.as<float>());
    if (rcode != KY_OK) return rcode;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_out.p, out_bytes, hipMemcpyDeviceToHost));
    return KY_OK;
}


// ------------------------------------------------------------------------------------------------
// run-time instantiations: a launch's exact render kernel, compiled on first use
//
// g_variants is a fixed table: the both_mis kernel for five combinations of scene facts, one kernel per other strategy, and the run-time-dispatched
// kernel for everything else -- a scene with a triangle in it, a rectangle light next to a point light, light_mis under the debug sampler.  The
// render kernel is a template over exactly those choices (ky_render.hpp), and the library carries its source (ky_rtc_sources.inc: ky_device.hpp,
// ky_render.hpp, include/kyhip.h as text), so with kyhip_set_jit(1) / KYHIP_JIT=1 a launch whose (sampler, strategy, integrator, deferred rays,
// general shapes, ALL of the scene's facts, table size) is not a row of the table gets its own instantiation: the sources are written to the cache
// directory, the ROCm compiler that built the library compiles one extern "C" kernel around render_kernel_body<...> into a gfx950 code object
// (a child process: `hipcc --genco`, 2-3 seconds, blocking the first launch that needs it), and the object is kept in memory and on disk
// ($KYHIP_CACHE_DIR, default ~/.cache/kyhip) and loaded per device with hipModuleLoadData.
// Why a child process and not hiprtc: a process that has PyTorch in it has PyTorch's bundled hiprtc / comgr in it, and the ROCm 7.0 one aborts the
// process on this kernel ("LLVM ERROR: Not supported instr", measured) -- a library cannot pick which comgr its host process has loaded.
// Off by default (the table serves every scene ky ships); if no compiler is found or a compile fails the launch takes the table's kernel and
// kyhip_jit_status() says why.
// ------------------------------------------------------------------------------------------------
#include "ky_rtc_sources.inc"

namespace kyjit {
struct Code {   // one compiled instantiation
    std::vector<char> object;
    bool failed = false;
};
static const char k_entry[] = "ky_jit_kernel";    // the extern "C" name of every run-time instantiation's kernel
static std::mutex g_mutex;                        // compiles are serialised
static std::map<std::string, Code> g_code;        // template arguments -> code object
static std::string g_status = "off";
static int g_mode = -1;                           // 0 off, 1 on

static int mode() {
    if (g_mode < 0) {
        const char* e = std::getenv("KYHIP_JIT");
        g_mode = (e && std::atoi(e) != 0) ? 1 : 0;
    }
    return g_mode;
}

// what the Makefile passes to hipcc for kyhip.hip, as far as device code goes
static const char k_flags[] = "--genco --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fno-hip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function "
                              "-Wno-bitwise-instead-of-logical";

static uint64_t hash_bytes(uint64_t h, const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}
static uint64_t source_hash() {
    static const uint64_t h = [] {
        uint64_t x = 0xcbf29ce484222325ull;
        for (const auto& src : g_rtc_sources) x = hash_bytes(x, src.text, std::strlen(src.text));
        return hash_bytes(x, k_flags, sizeof k_flags);
    }();
    return h;
}
static void mkdir_p(const std::string& dir) {
    for (size_t i = 1; i <= dir.size(); ++i)
        if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0755);
}
static std::string cache_dir() {
    std::string dir;
    if (const char* e = std::getenv("KYHIP_CACHE_DIR")) dir = e;
    else if (const char* home = std::getenv("HOME")) dir = std::string(home) + "/.cache/kyhip";
    else dir = "/tmp/kyhip-cache-" + std::to_string((long)getuid());
    mkdir_p(dir);
    return dir;
}
static bool write_text(const std::string& path, const char* text) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const size_t n = std::strlen(text);
    const bool ok = std::fwrite(text, 1, n, f) == n;
    std::fclose(f);
    return ok;
}
static bool read_file(const std::string& path, std::vector<char>& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    std::fclose(f);
    // a gfx950 code object, bare or as the offload bundle `hipcc --genco` writes (hipModuleLoadData takes both)
    return out.size() > 64 && (std::memcmp(out.data(), "\x7f" "ELF", 4) == 0 || std::memcmp(out.data(), "__CLANG_OFFLOAD_BUNDLE__", 24) == 0);
}
// KYHIP_JIT_FLAGS: more compiler flags for the run-time instantiations (tuning: -DKY_WAVES_PER_EU_QUEUE=5 ...); part of the cache key
static std::string extra_flags() {
    const char* e = std::getenv("KYHIP_JIT_FLAGS");
    if (!e) return "";
    for (const char* c = e; *c; ++c)   // the string goes into a shell command: compiler options only (-DNAME=1 -mllvm ...), no quoting, no metacharacters
        if (!(std::isalnum((unsigned char)*c) || std::strchr("_=-+.,/ ", *c))) return "";
    return e;
}
static std::string compiler() {
    if (const char* e = std::getenv("KYHIP_HIPCC")) return e;
    for (const char* p : {"/opt/rocm/bin/hipcc", "/usr/bin/hipcc"})
        if (access(p, X_OK) == 0) return p;
    return "hipcc";
}

// the code object of render_kernel_body<args> ("false, 48, false, false, 135, 11, false"), from memory, disk or the compiler; nullptr when it
// cannot be had (g_status says why)
static const Code* get_code(const std::string& args) {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_code.find(args);
    if (it != g_code.end()) return it->second.failed ? nullptr : &it->second;
    Code& c = g_code[args];
    const std::string dir = cache_dir();
    char name[64];
    const std::string extra = extra_flags();
    snprintf(name, sizeof name, "%016llx", (unsigned long long)hash_bytes(hash_bytes(source_hash(), args.data(), args.size()), extra.data(), extra.size()));
    const std::string object = dir + "/" + name + ".hsaco";
    if (read_file(object, c.object)) { g_status = "on (code objects from " + dir + ")"; return &c; }
    c.object.clear();
    c.failed = true;
    // the sources, laid out like the repository (ky_device.hpp includes "../../include/kyhip.h"), once per library build
    snprintf(name, sizeof name, "src-%016llx", (unsigned long long)source_hash());
    const std::string root = dir + "/" + name;
    mkdir_p(root + "/ky_amd/csrc");
    mkdir_p(root + "/include");
    bool ok = true;
    for (const auto& src : g_rtc_sources) {
        const std::string n = src.name;
        ok = ok && write_text(n.compare(0, 6, "../../") == 0 ? root + "/" + n.substr(6) : root + "/ky_amd/csrc/" + n, src.text);
    }
    const std::string tag = std::to_string((long)getpid()) + "-" + std::to_string((unsigned long long)hash_bytes(0, args.data(), args.size()));
    const std::string tu = root + "/ky_amd/csrc/jit-" + tag + ".hip", tmp = object + ".tmp" + tag, log = object + ".log";
    const std::string text = "#include \"ky_render.hpp\"\nextern \"C\" __global__ __launch_bounds__(256, (ky_waves_per_eu<" + args + ">())) void " + k_entry +
                             "(const kyd::DScene* __restrict__ S, kyd::RenderConst rc, ShardConst sh, unsigned* __restrict__ counter, unsigned long long* __restrict__ accum, "
                             "unsigned* __restrict__ flags, float4* __restrict__ queue_mem) {\n    render_kernel_body<" + args + ">(S, rc, sh, counter, accum, flags, queue_mem);\n}\n";
    ok = ok && write_text(tu, text.c_str());
    if (!ok) { g_status = "cannot write the sources under " + dir; return nullptr; }
    const std::string cmd = "'" + compiler() + "' " + k_flags + " " + extra + " -o '" + tmp + "' '" + tu + "' > '" + log + "' 2>&1";
    const int rc = std::system(cmd.c_str());
    (void)std::remove(tu.c_str());
    if (rc != 0 || !read_file(tmp, c.object)) {
        std::string tail;
        FILE* f = std::fopen(log.c_str(), "rb");
        if (f) { char buf[700]; const size_t n = std::fread(buf, 1, sizeof buf - 1, f); buf[n] = 0; tail = buf; std::fclose(f); }
        g_status = "compiling render_kernel_body<" + args + "> failed (" + compiler() + ", exit " + std::to_string(rc) + "): " + tail;
        (void)std::remove(tmp.c_str());
        c.object.clear();
        return nullptr;
    }
    (void)std::rename(tmp.c_str(), object.c_str());
    (void)std::remove(log.c_str());
    c.failed = false;
    g_status = "on (" + compiler() + "; code objects cached in " + dir + ")";
    return &c;
}
}  // namespace kyjit

extern "C" {

#ifdef KY_PROFILE_LANES
// debug builds only: reads and clears the lane-utilisation probes (32 x u64)
int kyhip_debug_lane_probe(unsigned long long* out32) {
    unsigned long long zero[32] = {0};
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_lane_probe), sizeof zero) != hipSuccess) return KY_ERR_DEVICE;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lane_probe), zero, sizeof zero) != hipSuccess) return KY_ERR_DEVICE;
    return KY_OK;
}
#endif

const char* kyhip_last_error(void) { return g_error.c_str(); }
#ifdef KY_PROFILE_CLOCKS
int kyhip_debug_clocks(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_clk), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
#ifdef KY_QE_STATS
int kyhip_debug_stats(unsigned long long* out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(kyd::g_qe_stats), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(kyd::g_qe_stats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

int kyhip_set_engine(int engine) {
    const int prev = current_engine();
    if (engine == KY_ENGINE_LANE || engine == KY_ENGINE_QUEUE) g_engine = engine;
    return prev;
}
int kyhip_set_specialisation(int on) {
    const int prev = specialisation_enabled() ? 1 : 0;
    if (on == 0 || on == 1) g_specialise = on;
    return prev;
}
int kyhip_set_shadow_queue(int mode) {
    const int prev = shadow_queue_mode();
    if (mode >= -1 && mode <= 1) g_shadow_queue = mode;
    return prev;
}
int kyhip_set_jit(int mode) {
    const int prev = kyjit::mode();
    if (mode == 0 || mode == 1) {
        kyjit::g_mode = mode;
        std::lock_guard<std::mutex> lock(kyjit::g_mutex);
        if (mode == 0) kyjit::g_status = "off";
        else if (kyjit::g_status == "off") kyjit::g_status = "on (nothing compiled yet)";
    }
    return prev;
}
const char* kyhip_jit_status(void) {
    static thread_local std::string s;
    std::lock_guard<std::mutex> lock(kyjit::g_mutex);
    s = kyjit::g_status;
    return s.c_str();
}
int64_t kyhip_jit_compile(const char* name_expression) {
    const size_t len = name_expression ? std::strlen(name_expression) : 0;
    if (len < 16 || std::strncmp(name_expression, "render_kernel<", 14) != 0 || name_expression[len - 1] != '>') return fail(KY_ERR_INVALID_VALUE, "not a render_kernel instantiation");
    for (size_t i = 14; i + 1 < len; ++i)   // template arguments only: digits, true / false, commas, blanks, a minus sign
        if (!std::strchr("0123456789truefals, -", name_expression[i])) return fail(KY_ERR_INVALID_VALUE, "not a render_kernel instantiation");
    const kyjit::Code* code = kyjit::get_code(std::string(name_expression + 14, len - 15));
    if (!code) return fail(KY_ERR_DEVICE, "%s", kyhip_jit_status());
    return (int64_t)code->object.size();
}
uint64_t kyhip_kernel_source_hash(void) { return kyjit::source_hash(); }
int kyhip_abi_version(void) { return KYHIP_ABI_VERSION; }
int kyhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int64_t kyhip_shard_tile_count(const ky_render_params* p) {
    if (!valid_params(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params");
    return make_shard(p).n_tiles;
}
int64_t kyhip_shard_float_count(const ky_render_params* p) {
    if (!valid_params(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params");
    return (int64_t)make_shard(p).n_pix * 3;
}

// per pixel of the shard: 3 x 64-bit fixed-point sums + one flag word
static size_t workspace_bytes_for(const ShardConst& s) { return (size_t)s.n_pix * (3 * sizeof(unsigned long long) + sizeof(unsigned)); }
size_t kyhip_workspace_bytes(const ky_render_params* p) {
    if (!valid_params(p)) return 0;
    return workspace_bytes_for(make_shard(p));
}

// The render-kernel instantiations of the lane engine, most specific first; a launch takes the first whose assumptions hold.
//   sampler   debug_sampler_t or random_sampler_t
//   strategy  -1: direct_sample_enum_t and integrator are read at run time (11 000 instructions, five waves per SIMD); otherwise both are
//             compile-time constants of the instantiation
//   queue     deferred shadow rays (scenes with two or more lights)
//   general   carries the reference's own formulations for quads that are not parallelograms, triangles and disks
//   feat      the KY_FEAT_* facts of the scene the instantiation assumes
using RenderFn = void (*)(const DScene*, RenderConst, ShardConst, unsigned*, unsigned long long*, unsigned*, float4*);
struct Variant {
    bool dbg;
    int strategy;
    bool queue, general;
    int feat, integrator;
    bool large;
    RenderFn fn;
};
#define KY_VARIANT(D, S, Q, G, F, I) Variant{D, S, Q, G, F, I, false, render_kernel<D, S, Q, G, F, I>}
#define KY_VARIANT_LARGE(D, S, Q, G, F, I) Variant{D, S, Q, G, F, I, true, render_kernel<D, S, Q, G, F, I, true>}
constexpr int IT = KY_INTEGRATOR_PATH_TRACING_ITERATION;
static const Variant g_variants[] = {
    // the iterative integrator, both_mis: by scene facts
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL | KY_FEAT_SMALL_TABLES, IT),   // one rectangle area light, at most 16 surfaces and 8 materials: configs[1], [4]
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL, IT),                  // one rectangle area light
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_DELTA, IT),             // one point / directional light
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV, IT),               // one environment light
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, true, false, KY_FEAT_VEACH, IT),                     // several sphere lights, no mirror / glass: configs[2]
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, true, false, 0, IT),                                 // many sphere lamps (shadow_queue_wanted): deferred shadow rays
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, 0, IT),
    // the iterative integrator, the other five strategies (render_direct_sample_enum 4779, render_mis_scene 4878)
    KY_VARIANT(false, KY_DIRECT_BSDF, false, false, KY_FEAT_VEACH, IT),                        // render_mis_scene's other strategies on its sphere lights
    KY_VARIANT(false, KY_DIRECT_LIGHT, true, false, KY_FEAT_VEACH, IT),
    KY_VARIANT(false, KY_DIRECT_LIGHT, false, false, KY_FEAT_VEACH, IT),
    KY_VARIANT(false, KY_DIRECT_BSDF_MIS, false, false, KY_FEAT_VEACH, IT),
    KY_VARIANT(false, KY_DIRECT_LIGHT_MIS, true, false, KY_FEAT_VEACH, IT),
    KY_VARIANT(false, KY_DIRECT_LIGHT_MIS, false, false, KY_FEAT_VEACH, IT),
    KY_VARIANT(false, KY_DIRECT_BSDF, false, false, KY_FEAT_CORNELL, IT),
    KY_VARIANT(false, KY_DIRECT_BSDF_MIS, false, false, KY_FEAT_CORNELL, IT),
    KY_VARIANT(false, KY_DIRECT_IDLE, false, false, 0, IT),
    KY_VARIANT(false, KY_DIRECT_BSDF, false, false, 0, IT),
    KY_VARIANT(false, KY_DIRECT_LIGHT, true, false, 0, IT),                                    // several lights: deferred shadow rays
    KY_VARIANT(false, KY_DIRECT_LIGHT_MIS, true, false, 0, IT),
    KY_VARIANT(false, KY_DIRECT_LIGHT, false, false, 0, IT),
    KY_VARIANT(false, KY_DIRECT_BSDF_MIS, false, false, 0, IT),
    KY_VARIANT(false, KY_DIRECT_LIGHT_MIS, false, false, 0, IT),
    // direct_lighting_t and the three recursive integrators with both_mis (render_multiple_integrator 4740-4777)
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL, KY_INTEGRATOR_PATH_TRACING_RECURSION),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV, KY_INTEGRATOR_PATH_TRACING_RECURSION),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL, KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV, KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_DELTA, KY_INTEGRATOR_PATH_TRACING_RECURSION),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_DELTA, KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL, KY_INTEGRATOR_DIRECT_LIGHTING),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_DELTA, KY_INTEGRATOR_DIRECT_LIGHTING),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV, KY_INTEGRATOR_DIRECT_LIGHTING),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, 0, KY_INTEGRATOR_DIRECT_LIGHTING),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, 0, KY_INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, 0, KY_INTEGRATOR_PATH_TRACING_RECURSION),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, 0, KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED),
    // scenes with triangles, disks or non-planar quads under the default strategy: the strategy as a compile-time constant is worth 17-23 % over the
    // run-time-dispatched kernel below (round 4: tools/room_rates.py, the random rooms with general shapes)
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, true, 0, IT),
    // everything else: strategy and integrator at run time; the debug sampler; scenes with general shapes
    KY_VARIANT(false, -1, false, false, 0, IT),
    KY_VARIANT(true, -1, false, false, 0, IT),
    KY_VARIANT(false, -1, false, true, 0, IT),
    KY_VARIANT(true, -1, false, true, 0, IT),
    // scenes beyond the static LDS block (more than 64 surfaces or 32 materials): the run-time-dispatched kernels with a scene-sized block
    KY_VARIANT_LARGE(false, -1, false, true, 0, IT),
    KY_VARIANT_LARGE(true, -1, false, true, 0, IT),
};
constexpr int KY_N_VARIANTS = (int)(sizeof g_variants / sizeof g_variants[0]);
static_assert(KY_N_VARIANTS <= 48, "DeviceCtx::variant_blocks");

static const Variant* pick_variant(const ky_render_params* p, const DScene* packed, bool deferred_rays, int n_pix) {
    const bool dbg = p->sampler == KY_SAMPLER_DEBUG;
    const bool general = packed->general != 0;
    const bool large = packed->n_surfaces > KY_LDS_SURFACES || packed->n_materials > KY_LDS_MATERIALS;
    for (const Variant& v : g_variants) {
        if (v.dbg != dbg || v.large != large) continue;
        if (general && !v.general) continue;
        if (v.strategy >= 0) {
            if (!specialisation_enabled() && !(v.strategy == KY_DIRECT_BOTH_MIS && v.feat == 0 && v.integrator == IT)) continue;   // KYHIP_SPECIALISE=0 keeps both_mis (and its queue form)
            if (v.strategy != p->direct_sample || v.integrator != p->integrator) continue;
        }
        if ((v.feat & packed->feat) != v.feat) continue;
        // deferred shadow rays for the scenes shadow_queue_wanted() names.  The ray's destination tag holds the pixel in 26 bits.
        if (v.queue && !(n_pix < (1 << 26) && deferred_rays)) continue;
        return &v;
    }
    return nullptr;   // not reached: the last entries accept everything
}

int kyhip_render_tiles_device(int device, const ky_scene* scene, const ky_render_params* p, float* d_tiles, void* d_workspace,
                              size_t workspace_bytes, void* stream_) {
    if (!valid_params(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params (integrator %d, direct_sample %d)", p ? p->integrator : -1, p ? p->direct_sample : -1);
    if (!shard_in_range(p)) return fail(KY_ERR_LIMIT, "frame too large for the device's 32-bit work-item and pixel indices (%d x %d, %d spp)", p->width, p->height, p->samples_per_pixel);
    if (!d_tiles) return fail(KY_ERR_INVALID_VALUE, "d_tiles is NULL");
    DeviceCtx* c;
    int rcode = get_ctx(device, &c);
    if (rcode != KY_OK) return rcode;
    std::lock_guard<std::mutex> lock(c->m);
    hipStream_t stream = (hipStream_t)stream_;
    SceneSlot* sc;
    rcode = upload_scene(c, scene, stream, &sc);
    if (rcode != KY_OK) return rcode;

    const ShardConst sh = make_shard(p);
    if (sh.n_tiles == 0) return KY_OK;
    const RenderConst rc = make_rc(p);
    StreamState* st;
    rcode = get_stream_state(c, stream, &st);
    if (rcode != KY_OK) return rcode;

    const size_t need = workspace_bytes_for(sh);
    void* ws = d_workspace;
    if (!(d_workspace && workspace_bytes >= need)) {
        if (st->ws_bytes < need) {
            HIP_TRY(hipStreamSynchronize(stream));   // the previous call's kernels on this stream still use the old block
            if (st->ws) HIP_TRY(hipFree(st->ws));
            st->ws = nullptr; st->ws_bytes = 0;
            HIP_TRY(hipMalloc(&st->ws, need));
            st->ws_bytes = need;
        }
        ws = st->ws;
    }
    unsigned long long* accum = (unsigned long long*)ws;
    unsigned* flags = (unsigned*)(accum + (size_t)sh.n_pix * 3);
    const bool large_scene = scene->surface_count > KY_LDS_SURFACES || scene->material_count > KY_LDS_MATERIALS;
    const size_t lds_bytes = large_scene ? (size_t)lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count) : 0;   // LARGE kernels' LdsScene
    HIP_TRY(hipMemsetAsync(ws, 0, need, stream));
    HIP_TRY(hipMemsetAsync(st->d_counter, 0, sizeof(unsigned), stream));

    // the queue engine implements path_tracing_iteration_t; every other integrator runs on the lane engine
    if (current_engine() == KY_ENGINE_QUEUE && p->integrator == KY_INTEGRATOR_PATH_TRACING_ITERATION && !large_scene) {
        const int variant = p->sampler == KY_SAMPLER_DEBUG ? 2 : (p->direct_sample == KY_DIRECT_BOTH_MIS ? 0 : 1);
        const int per_cu = c->q_blocks_per_cu[variant] > 0 ? c->q_blocks_per_cu[variant] : 1;
        unsigned grid = (unsigned)(c->cus * per_cu);
        const unsigned need_blocks = (unsigned)(((unsigned long long)sh.n_items * 64u + QE_SLOTS - 1) / QE_SLOTS);
        if (grid > need_blocks) grid = need_blocks;
        if (grid < 1) grid = 1;
        HIP_TRY(hipEventRecord(st->ev0, stream));
        if (variant == 0) hipLaunchKernelGGL((render_kernel_q<false, KY_DIRECT_BOTH_MIS>), dim3(grid), dim3(QE_THREADS), 0, stream, sc->d, rc, sh, st->d_counter, accum, flags);
        else if (variant == 1) hipLaunchKernelGGL((render_kernel_q<false, -1>), dim3(grid), dim3(QE_THREADS), 0, stream, sc->d, rc, sh, st->d_counter, accum, flags);
        else hipLaunchKernelGGL((render_kernel_q<true, -1>), dim3(grid), dim3(QE_THREADS), 0, stream, sc->d, rc, sh, st->d_counter, accum, flags);
        c->last_variant = -2;
    } else {
        const Variant* v = pick_variant(p, sc->h, shadow_queue_wanted(scene), sh.n_pix);
        if (!v) return fail(KY_ERR_DEVICE, "internal: no render kernel for these parameters");
        const int vi = (int)(v - g_variants);
        // run-time instantiation (kyhip_set_jit(1)): this launch's own kernel -- its sampler, strategy and integrator as compile-time constants and ALL
        // of the scene's facts -- unless the table's pick is exactly that already
        DeviceCtx::JitKernel* jk = nullptr;
        bool queue = v->queue;
        if (kyjit::mode() == 1 && specialisation_enabled() && p->integrator >= KY_INTEGRATOR_DIRECT_LIGHTING) {   // (kyhip_set_specialisation(0) asks for the fact-free kernels: nothing to instantiate)
            const bool dbg = p->sampler == KY_SAMPLER_DEBUG, general = sc->h->general != 0;
            const int feat = (dbg || general) ? 0 : sc->h->feat;
            const bool want_queue = (p->direct_sample == KY_DIRECT_BOTH_MIS || p->direct_sample == KY_DIRECT_LIGHT_MIS || p->direct_sample == KY_DIRECT_LIGHT) &&
                                    p->integrator == KY_INTEGRATOR_PATH_TRACING_ITERATION && sh.n_pix < (1 << 26) && !general && shadow_queue_wanted(scene);
            const bool same = v->dbg == dbg && v->strategy == p->direct_sample && v->queue == want_queue && v->general == general && v->feat == feat &&
                              v->integrator == p->integrator && v->large == large_scene;
            if (!same) {
                char expr[192];
                snprintf(expr, sizeof expr, "%s, %d, %s, %s, %d, %d, %s", dbg ? "true" : "false", p->direct_sample, want_queue ? "true" : "false",
                         general ? "true" : "false", feat, p->integrator, large_scene ? "true" : "false");
                DeviceCtx::JitKernel& k = c->jit[expr];
                if (!k.fn && !k.failed) {
                    const kyjit::Code* code = kyjit::get_code(expr);   // blocks for the compile the first time (a few seconds), then memory / disk
                    if (!(code && hipModuleLoadData(&k.module, code->object.data()) == hipSuccess && hipModuleGetFunction(&k.fn, k.module, kyjit::k_entry) == hipSuccess)) {
                        (void)hipGetLastError();
                        k.fn = nullptr;
                        k.failed = true;   // the table's kernel serves this launch and every later one of its kind
                    }
                }
                if (k.fn) {
                    if (k.lds != lds_bytes) {
                        int per_cu = 0;
                        HIP_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.fn, 256, lds_bytes));
                        k.per_cu = per_cu > 0 ? per_cu : 1;
                        k.lds = lds_bytes;
                    }
                    jk = &k;
                    queue = want_queue;
                    char desc[224];   // the table's way of naming a kernel, then the template arguments it was compiled with
                    snprintf(desc, sizeof desc, "render_kernel<%sstrategy %d%s%s%s, feat %d, integrator %d> = render_kernel<", dbg ? "debug sampler, " : "", p->direct_sample,
                             want_queue ? ", deferred shadow rays" : "", general ? ", general shapes" : "", large_scene ? ", scene-sized LDS block" : "", feat, p->integrator);
                    c->last_jit = std::string(desc) + expr + ">";
                }
            }
        }
        if (!jk && (c->variant_blocks[vi] == 0 || c->variant_lds[vi] != lds_bytes)) {   // resident workgroups per CU: depends on the scene's LDS block
            int per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v->fn, 256, lds_bytes));
            c->variant_blocks[vi] = per_cu > 0 ? per_cu : 1;
            c->variant_lds[vi] = lds_bytes;
        }
        const int per_cu = jk ? jk->per_cu : c->variant_blocks[vi];
        unsigned grid = (unsigned)(c->cus * per_cu);
        const int cap = blocks_per_cu_cap();
        if (cap > 0 && cap < per_cu) grid = (unsigned)(c->cus * cap);
        const unsigned need_blocks = sh.n_items / 4 + 1;
        if (grid > need_blocks) grid = need_blocks;
        if (grid < 1) grid = 1;
        if (queue && st->sq_blocks < (size_t)c->cus * per_cu) {   // the wavefronts' shadow-ray stacks of this stream's launches: one per resident
            // wavefront of the LARGEST grid any QUEUE variant has been launched with on this stream (kernel: queue_mem + (block * 4 + wave) * cap)
            if (st->d_shadow_queue) {
                HIP_TRY(hipStreamSynchronize(stream));   // the previous launches on this stream still push to the old block
                HIP_TRY(hipFree(st->d_shadow_queue));
                st->d_shadow_queue = nullptr; st->sq_blocks = 0;
            }
            const size_t blocks = (size_t)c->cus * per_cu;
            HIP_TRY(hipMalloc(&st->d_shadow_queue, blocks * 4 * KY_SQ_ENTRY * KY_SQ_CAP * sizeof(float4)));
            st->sq_blocks = blocks;
        }
        HIP_TRY(hipEventRecord(st->ev0, stream));
        float4* queue_mem = queue ? st->d_shadow_queue : (float4*)nullptr;
        if (jk) {
            const DScene* a_scene = sc->d;
            RenderConst a_rc = rc;
            ShardConst a_sh = sh;
            unsigned* a_counter = st->d_counter;
            void* args[] = {&a_scene, &a_rc, &a_sh, &a_counter, &accum, &flags, &queue_mem};
            HIP_TRY(hipModuleLaunchKernel(jk->fn, grid, 1, 1, 256, 1, 1, (unsigned)lds_bytes, stream, args, nullptr));
            c->last_variant = -3;
        } else {
            hipLaunchKernelGGL(v->fn, dim3(grid), dim3(256), lds_bytes, stream, (const DScene*)sc->d, rc, sh, st->d_counter, accum, flags, queue_mem);
            c->last_variant = vi;
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(st->ev1, stream));
    st->timing_valid = true;
    c->last_launch = st;
    const int nf = sh.n_pix * 3;
    hipLaunchKernelGGL(resolve_kernel, dim3((nf + 255) / 256), dim3(256), 0, stream, accum, flags, d_tiles, nf);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(st->done, stream));
    sc->readers |= 1u << (unsigned)(st - c->ss);
    return KY_OK;
}

// resolves the event pair of the last launch on `device`; its stream must have been synchronised
float kyhip_kernel_ms(int device) {
    DeviceCtx* c = find_ctx(device);
    if (!c) return -1.f;
    std::lock_guard<std::mutex> lock(c->m);
    if (!c->last_launch || !c->last_launch->timing_valid) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, c->last_launch->ev0, c->last_launch->ev1) != hipSuccess) return -1.f;
    return ms;
}

const char* kyhip_last_kernel(int device) {
    static thread_local std::string name;
    name.clear();
    DeviceCtx* c = find_ctx(device);
    if (!c) return name.c_str();
    std::lock_guard<std::mutex> lock(c->m);
    if (c->last_variant == -2) name = "render_kernel_q (queue engine)";
    else if (c->last_variant == -3) name = c->last_jit + " (run-time instantiation; template arguments: DEBUG_SAMPLER, STRATEGY, QUEUE, GENERAL, FEAT, INTEGRATOR, LARGE)";
    else if (c->last_variant >= 0) {
        const Variant& v = g_variants[c->last_variant];
        char buf[160];
        snprintf(buf, sizeof buf, "render_kernel<%sstrategy %d%s%s%s, feat %d, integrator %d>", v.dbg ? "debug sampler, " : "", v.strategy, v.queue ? ", deferred shadow rays" : "",
                 v.general ? ", general shapes" : "", v.large ? ", scene-sized LDS block" : "", v.feat, v.integrator);
        name = buf;
    }
    return name.c_str();
}

int kyhip_film_add_tiles_device(int device, const ky_render_params* p, const float* d_tiles, float* d_film, size_t stride_px, void* stream_) {
    if (!valid_params(p) || !shard_in_range(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params");
    if (!d_tiles || !d_film || stride_px < (size_t)p->width) return fail(KY_ERR_INVALID_VALUE, "bad film arguments");
    DeviceCtx* c;
    int rcode = get_ctx(device, &c);
    if (rcode != KY_OK) return rcode;
    std::lock_guard<std::mutex> lock(c->m);
    const ShardConst sh = make_shard(p);
    if (sh.n_pix == 0) return KY_OK;
    hipLaunchKernelGGL(film_add_kernel, dim3((sh.n_pix + 255) / 256), dim3(256), 0, (hipStream_t)stream_, d_tiles, d_film, stride_px, sh, p->width, p->height);
    HIP_TRY(hipGetLastError());
    return KY_OK;
}

int kyhip_film_add_gathered_device(int device, const ky_render_params* p, int world, const float* d_gathered, size_t rank_stride_floats,
                                   float* d_film, size_t stride_px, void* stream_) {
    if (!valid_params(p) || !shard_in_range(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params");
    if (world < 1 || (long long)p->tile_step * world > 0x7fffffffLL) return fail(KY_ERR_INVALID_VALUE, "bad shard count %d", world);
    if (!d_gathered || !d_film || stride_px < (size_t)p->width) return fail(KY_ERR_INVALID_VALUE, "bad film arguments");
    ky_render_params q = *p;   // the largest shard is shard 0: its buffer must fit the stride
    q.tile_step = p->tile_step * world;
    if (rank_stride_floats < (size_t)make_shard(&q).n_pix * 3) return fail(KY_ERR_INVALID_VALUE, "rank_stride_floats is smaller than a shard's tile buffer");
    DeviceCtx* c;
    int rcode = get_ctx(device, &c);
    if (rcode != KY_OK) return rcode;
    std::lock_guard<std::mutex> lock(c->m);
    const int n = p->width * p->height, tiles_x = (p->width + p->tile_w - 1) / p->tile_w;
    hipLaunchKernelGGL(film_add_gathered_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, d_gathered, rank_stride_floats, world, d_film, stride_px,
                       p->tile_w, p->tile_h, p->tile_first, p->tile_step, tiles_x, p->width, p->height);
    HIP_TRY(hipGetLastError());
    return KY_OK;
}

// integrator_t::render on a LIST of devices (the reference spreads the pixel loop over all cores inside render(),
// ky.cpp:3696-3699).  Shard i of the frame goes to devices[i] on that device's own stream; the tile buffers are gathered on
// devices[0] (peer copies over xGMI), de-interleaved by one kernel and added into the caller's film.
//
// What the call owns besides the kernels is kept per device and reused (SeamBuffers): the gather block and the device film on the root, a
// PINNED host staging film, the tile buffers of remote shards.  The film comes back in row bands: band b's download is followed by an
// event, and a few host threads add band b into the caller's film (film_t::add_color, 1586-1590) the moment its event has fired, so the
// host's pass over the film overlaps the rest of the download.  (Round 3 allocated and freed two device buffers per call, downloaded into
// pageable memory and added with one scalar loop afterwards: 1-2 ms on a 9.4 MB film, a third of a 64-spp frame.)
// CPUs this process may really use: the affinity mask, cut down to the cgroup's quota where there is one (a container granted 2 CPUs of a 256-thread host
// reports 256 from std::thread::hardware_concurrency(); four adding threads on two CPUs were slower than two)
static int cpus_granted() {
    static const int n = [] {
        int cpus = (int)std::thread::hardware_concurrency();
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

static void host_add_rows(float* __restrict__ film, size_t stride_px, const float* __restrict__ src, int width, int y0, int y1) {
    const size_t n = (size_t)width * 3;
    for (int y = y0; y < y1; ++y) {
        float* __restrict__ dst = film + (size_t)y * stride_px * 3;
        const float* __restrict__ row = src + (size_t)y * n;
        for (size_t i = 0; i < n; ++i) dst[i] += row[i];   // vectorised by the host compiler
    }
}
// A few parked host threads for the banded add (creating and joining threads per call cost 50-100 us of a 4 ms frame).  run(n, fn) calls
// fn(0) ... fn(n - 1), fn(0) on the caller; one job at a time (the callers hold a seam mutex anyway, this one serialises across devices).
class HostPool {
public:
    void run(int n, const std::function<void(int)>& fn) {
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
private:
    void loop(int id) {
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
    std::mutex job_m_, m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, pending_ = 0, n_threads_ = 0;
    pid_t pid_ = 0;
    unsigned long long generation_ = 0;
};
static HostPool& host_pool() { static HostPool* pool = new HostPool; return *pool; }   // never destroyed: no thread joins at process exit

static int seam_reserve(void** p, size_t* have, size_t need, bool pinned) {
    if (*have >= need) return KY_OK;
    if (*p) { HIP_TRY(pinned ? hipHostFree(*p) : hipFree(*p)); *p = nullptr; *have = 0; }
    const size_t bytes = need + need / 4;   // some slack: a caller that alternates frame sizes does not reallocate on every call
    HIP_TRY(pinned ? hipHostMalloc(p, bytes) : hipMalloc(p, bytes < 16 ? 16 : bytes));
    *have = bytes;
    return KY_OK;
}

int kyhip_render_multi(const int* devices, int n_devices, const ky_scene* scene, const ky_render_params* p, float* film_rgb, size_t stride_px) {
    if (!valid_params(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params (integrator %d, direct_sample %d)", p ? p->integrator : -1, p ? p->direct_sample : -1);
    if (!shard_in_range(p)) return fail(KY_ERR_LIMIT, "frame too large for the device's 32-bit work-item and pixel indices (%d x %d, %d spp)", p->width, p->height, p->samples_per_pixel);
    if (!devices || n_devices < 1 || n_devices > 64) return fail(KY_ERR_INVALID_VALUE, "bad device list");
    if ((long long)p->tile_step * n_devices > 0x7fffffffLL) return fail(KY_ERR_INVALID_VALUE, "tile_step x devices overflows");
    if (!film_rgb || stride_px < (size_t)p->width) return fail(KY_ERR_INVALID_VALUE, "bad film arguments");
    const int root = devices[0];
    std::vector<ky_render_params> shard(n_devices, *p);
    std::vector<DeviceCtx*> ctx(n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) {
        shard[i].tile_first = p->tile_first + i * p->tile_step;
        shard[i].tile_step = p->tile_step * n_devices;
        const int rc = get_ctx(devices[i], &ctx[i]);
        if (rc != KY_OK) return rc;
    }
    const size_t rank_stride = (size_t)make_shard(&shard[0]).n_pix * 3;   // shard 0 owns the most tiles
    const size_t film_floats = (size_t)p->width * p->height * 3;

    // The cached buffers of every device of the list belong to this call until it returns: their seam mutexes are taken in ascending
    // device order (two calls with overlapping lists cannot deadlock), never while a context's enqueue mutex is held.
    std::vector<int> order(devices, devices + n_devices);
    std::sort(order.begin(), order.end());
    order.erase(std::unique(order.begin(), order.end()), order.end());
    std::vector<std::unique_lock<std::mutex>> seam_locks;
    for (int d : order) seam_locks.emplace_back(find_ctx(d)->seam.m);

    // buffers: one gather block and the film on the root, the pinned staging film; a tile buffer per remote shard on its device
    HIP_TRY(hipSetDevice(root));
    SeamBuffers& sb = ctx[0]->seam;
    int rcode = seam_reserve(&sb.d_gather, &sb.gather_bytes, rank_stride * n_devices * sizeof(float), false);
    if (rcode == KY_OK) rcode = seam_reserve(&sb.d_film, &sb.film_bytes, film_floats * sizeof(float), false);
    if (rcode == KY_OK) rcode = seam_reserve((void**)&sb.h_stage, &sb.stage_bytes, film_floats * sizeof(float), true);
    if (rcode != KY_OK) return rcode;
    for (hipEvent_t& e : sb.band)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    float* const d_gather = (float*)sb.d_gather;
    float* const d_film = (float*)sb.d_film;
    hipStream_t root_stream = ctx[0]->stream;
    HIP_TRY(hipMemsetAsync(d_film, 0, film_floats * sizeof(float), root_stream));
    std::vector<float*> remote(n_devices, nullptr);
    std::vector<int> remote_slot(n_devices, 0);   // a device listed k times needs k tile buffers
    std::vector<hipEvent_t> done(n_devices, nullptr);
    struct EventGuard {
        std::vector<hipEvent_t>& ev;
        ~EventGuard() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    } event_guard{done};

    // 1. every shard is enqueued before anything is waited for: the devices render concurrently.  From here on a failure must not
    // return before the streams are drained (step 3): shards already launched write into buffers this function uses.
#define HIP_CHECK_BREAK(expr)                                                                                         \
    {                                                                                                               \
        const hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) { rcode = fail(KY_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); break; } \
    }
    for (int i = 0; i < n_devices && rcode == KY_OK; ++i) {
        float* dst = d_gather + rank_stride * i;
        if (devices[i] != root) {
            HIP_CHECK_BREAK(hipSetDevice(devices[i]));
            SeamBuffers& rb = ctx[i]->seam;
            int slot = 0;
            for (int j = 0; j < i; ++j) slot += devices[j] == devices[i];
            if ((int)rb.d_remote.size() <= slot) { rb.d_remote.resize(slot + 1, nullptr); rb.remote_bytes.resize(slot + 1, 0); }
            rcode = seam_reserve(&rb.d_remote[slot], &rb.remote_bytes[slot], rank_stride * sizeof(float), false);
            if (rcode != KY_OK) break;
            remote[i] = dst = (float*)rb.d_remote[slot];
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, root, devices[i]) == hipSuccess && can) {   // direct xGMI copies; staged otherwise
                HIP_CHECK_BREAK(hipSetDevice(root));
                (void)hipDeviceEnablePeerAccess(devices[i], 0);
                (void)hipGetLastError();   // "already enabled" is not an error here
            }
        }
        rcode = kyhip_render_tiles_device(devices[i], scene, &shard[i], dst, nullptr, 0, ctx[i]->stream);
        if (rcode != KY_OK) break;
        if (devices[i] != root) {
            HIP_CHECK_BREAK(hipSetDevice(devices[i]));
            HIP_CHECK_BREAK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
            HIP_CHECK_BREAK(hipEventRecord(done[i], ctx[i]->stream));
        }
    }
    // 2. the gather: one peer copy per remote shard, ordered behind that shard's kernels; then one add into the device film
    for (int once = 0; once < 1 && rcode == KY_OK; ++once) {
        HIP_CHECK_BREAK(hipSetDevice(root));
        bool failed = false;
        for (int i = 0; i < n_devices && !failed; ++i) {
            if (devices[i] == root) continue;
            hipError_t e = hipStreamWaitEvent(root_stream, done[i], 0);
            const size_t bytes = (size_t)make_shard(&shard[i]).n_pix * 3 * sizeof(float);
            if (e == hipSuccess && bytes) e = hipMemcpyPeerAsync(d_gather + rank_stride * i, root, remote[i], devices[i], bytes, root_stream);
            if (e != hipSuccess) { rcode = fail(KY_ERR_DEVICE, "gathering shard %d failed: %s", i, hipGetErrorString(e)); failed = true; }
        }
        if (failed) break;
        rcode = kyhip_film_add_gathered_device(root, p, n_devices, d_gather, rank_stride, d_film, (size_t)p->width, root_stream);
    }
    // 3. the film comes home in row bands, each followed by an event
    const size_t row_bytes = (size_t)p->width * 3 * sizeof(float);
    int n_bands = (int)std::min<size_t>(KY_SEAM_BANDS, std::max<size_t>(1, film_floats * sizeof(float) / (512u << 10)));   // bands of at least 512 KB
    n_bands = std::min(n_bands, p->height);
    auto band_row = [&](int b) { return (int)((long long)p->height * b / n_bands); };
    int bands_enqueued = 0;
    for (int b = 0; b < n_bands && rcode == KY_OK; ++b) {
        const int y0 = band_row(b), y1 = band_row(b + 1);
        HIP_CHECK_BREAK(hipMemcpyAsync(sb.h_stage + (size_t)y0 * p->width * 3, d_film + (size_t)y0 * p->width * 3, (size_t)(y1 - y0) * row_bytes, hipMemcpyDeviceToHost, root_stream));
        HIP_CHECK_BREAK(hipEventRecord(sb.band[b], root_stream));
        bands_enqueued = b + 1;
    }
#undef HIP_CHECK_BREAK
    // 4. host threads add the bands as they arrive: film_t::add_color, 1586-1590.  Every thread works on its slice of the rows of EVERY band (a band is
    // waited for once, by the thread that gets to it first under the band's flag), so the threads are all busy from the first band on.
    hipError_t sync_err = hipSuccess;
    if (rcode == KY_OK && bands_enqueued == n_bands) {
        const int n_threads = std::max(1, std::min({p->height, cpus_granted(), (int)KY_SEAM_THREADS}));
        std::vector<hipError_t> errs(n_threads, hipSuccess);
        auto work = [&](int t) {
            for (int b = 0; b < n_bands; ++b) {
                const hipError_t e = hipEventSynchronize(sb.band[b]);   // (returns at once for a band that has arrived)
                if (e != hipSuccess) { errs[t] = e; return; }
                const int y0 = band_row(b), y1 = band_row(b + 1);
                const int r0 = y0 + (int)((long long)(y1 - y0) * t / n_threads), r1 = y0 + (int)((long long)(y1 - y0) * (t + 1) / n_threads);
                host_add_rows(film_rgb, stride_px, sb.h_stage, p->width, r0, r1);
            }
        };
        host_pool().run(n_threads, work);
        for (hipError_t e : errs) if (e != hipSuccess) sync_err = e;
    }
    // 5. every stream that may still use a buffer of this call is drained before the call returns (also on errors)
    for (int i = 0; i < n_devices; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess) continue;
        const hipError_t e = hipStreamSynchronize(ctx[i]->stream);
        if (e != hipSuccess) sync_err = e;
    }
    (void)hipSetDevice(root);
    if (rcode != KY_OK) return rcode;
    if (sync_err != hipSuccess) return fail(KY_ERR_DEVICE, "render failed: %s (the caller's film may hold a part of the frame)", hipGetErrorString(sync_err));
    return KY_OK;
}

int kyhip_render(int device, const ky_scene* scene, const ky_render_params* p, float* film_rgb, size_t stride_px) {
    return kyhip_render_multi(&device, 1, scene, p, film_rgb, stride_px);
}

// ---- KAT entry points ----
int kyhip_kat_intersect(int device, const ky_shape* shape, const float* rays7, int n, float* out8) {
    if (!shape || !rays7 || !out8 || n <= 0) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    if (shape->kind < KY_SHAPE_DISK || shape->kind > KY_SHAPE_SPHERE) return fail(KY_ERR_INVALID_VALUE, "unknown shape kind");
    if (!shape_normal_ok(*shape)) return fail(KY_ERR_INVALID_VALUE, "the stored normal must be unit length");
    KatShape ks{};
    pack_shape(*shape, 0, &ks.surf, &ks.full);
    cp3(ks.hit.n, shape->kind == KY_SHAPE_SPHERE ? shape->p[0] : shape->normal);
    ks.hit.kind = shape->kind;
    return kat_run(device, rays7, (size_t)n * 7 * 4, out8, (size_t)n * 8 * 4, [&](DeviceCtx*, const float* d_in, float* d_out) {
        hipLaunchKernelGGL(kat_intersect_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, ks, d_in, n, d_out);
        return (int)KY_OK;
    });
}

int kyhip_kat_camera(int device, const ky_camera* camera, const float* p_film2, int n, float* out6) {
    if (!camera || !p_film2 || !out6 || n <= 0) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    ky_scene sc{};
    sc.environment_light = -1;
    sc.camera = *camera;
    if (!(camera->resolution[0] > 0) || !(camera->resolution[1] > 0)) return fail(KY_ERR_INVALID_VALUE, "bad camera resolution");
    return kat_run(device, p_film2, (size_t)n * 2 * 4, out6, (size_t)n * 6 * 4, [&](DeviceCtx* c, const float* d_in, float* d_out) {
        SceneSlot* slot; int r = upload_scene(c, &sc, 0, &slot);
        if (r != KY_OK) return r;
        hipLaunchKernelGGL(kat_camera_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, (const DScene*)slot->d, d_in, n, d_out);
        return (int)KY_OK;
    });
}

int kyhip_kat_bsdf(int device, const ky_material* m, const float* in12, int n, float* out13) {
    if (!m || !in12 || !out13 || n <= 0) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    if (m->kind < KY_MATERIAL_MATTE || m->kind > KY_MATERIAL_PLASTIC) return fail(KY_ERR_INVALID_VALUE, "unknown material kind");
    DMat d{};
    pack_material(*m, &d);
    return kat_run(device, in12, (size_t)n * 12 * 4, out13, (size_t)n * 13 * 4, [&](DeviceCtx*, const float* d_in, float* d_out) {
        hipLaunchKernelGGL(kat_bsdf_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d, d_in, n, d_out);
        return (int)KY_OK;
    });
}

int kyhip_kat_light(int device, const ky_scene* scene, int light, const float* in11, int n, float* out11) {
    if (!scene || !in11 || !out11 || n <= 0 || light < 0 || light >= scene->light_count) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    return kat_run(device, in11, (size_t)n * 11 * 4, out11, (size_t)n * 11 * 4, [&](DeviceCtx* c, const float* d_in, float* d_out) {
        SceneSlot* sc; int r = upload_scene(c, scene, 0, &sc);
        if (r != KY_OK) return r;
        hipLaunchKernelGGL(kat_light_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, (const DScene*)sc->d, light, d_in, n, d_out);
        return (int)KY_OK;
    });
}

int kyhip_kat_scene_intersect(int device, const ky_scene* scene, const float* rays7, int n, float* out9) {
    if (!scene || !rays7 || !out9 || n <= 0) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    return kat_run(device, rays7, (size_t)n * 7 * 4, out9, (size_t)n * 9 * 4, [&](DeviceCtx* c, const float* d_in, float* d_out) {
        SceneSlot* sc; int r = upload_scene(c, scene, 0, &sc);
        if (r != KY_OK) return r;
        hipLaunchKernelGGL(kat_scene_intersect_kernel, dim3((n + 255) / 256), dim3(256), lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count), 0, (const DScene*)sc->d, d_in, n, d_out);
        return (int)KY_OK;
    });
}

static int kat_occluded_impl(int device, const ky_scene* scene, const float* in9, int n, float* out1, int table) {
    if (!scene || !in9 || !out1 || n <= 0) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    if (table >= scene->light_count) return fail(KY_ERR_INVALID_VALUE, "light %d out of range", table);
    return kat_run(device, in9, (size_t)n * 9 * 4, out1, (size_t)n * 4, [&](DeviceCtx* c, const float* d_in, float* d_out) {
        SceneSlot* sc; int r = upload_scene(c, scene, 0, &sc);
        if (r != KY_OK) return r;
        hipLaunchKernelGGL(kat_occluded_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, (const DScene*)sc->d, d_in, n, d_out, table);
        return (int)KY_OK;
    });
}
int kyhip_kat_occluded(int device, const ky_scene* scene, const float* in9, int n, float* out1) { return kat_occluded_impl(device, scene, in9, n, out1, -2); }
int kyhip_kat_occluded_between(int device, const ky_scene* scene, int light, const float* in9, int n, float* out1) {
    if (light < -1) return fail(KY_ERR_INVALID_VALUE, "light %d out of range", light);
    return kat_occluded_impl(device, scene, in9, n, out1, light);
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

int kyhip_kat_li(int device, const ky_scene* scene, const ky_render_params* p, int x, int y, int s0, int n, float* out3) {
    if (!valid_params(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params");
    if (!scene || !out3 || n <= 0 || x < 0 || y < 0 || x >= p->width || y >= p->height) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    const RenderConst rc = make_rc(p);
    const bool dbg = p->sampler == KY_SAMPLER_DEBUG;
    float dummy = 0.f;
    return kat_run(device, &dummy, 4, out3, (size_t)n * 3 * 4, [&](DeviceCtx* c, const float*, float* d_out) {
        SceneSlot* sc; int r = upload_scene(c, scene, 0, &sc);
        if (r != KY_OK) return r;
        if (dbg) hipLaunchKernelGGL(kat_li_kernel<true>, dim3((n + 255) / 256), dim3(256), lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count), 0, (const DScene*)sc->d, rc, x, y, s0, n, d_out);
        else hipLaunchKernelGGL(kat_li_kernel<false>, dim3((n + 255) / 256), dim3(256), lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count), 0, (const DScene*)sc->d, rc, x, y, s0, n, d_out);
        return (int)KY_OK;
    });
}

int kyhip_kat_nee(int device, const ky_scene* scene, int direct_sample, int light, const float* in15, int n, float* out6) {
    if (!scene || !in15 || !out6 || n <= 0 || light < 0 || light >= scene->light_count) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    if (direct_sample != KY_DIRECT_BSDF && direct_sample != KY_DIRECT_LIGHT && direct_sample != KY_DIRECT_BSDF_MIS && direct_sample != KY_DIRECT_LIGHT_MIS &&
        direct_sample != KY_DIRECT_BOTH_MIS)
        return fail(KY_ERR_INVALID_VALUE, "direct_sample %d has no estimator to test", direct_sample);
    for (int i = 0; i < n; ++i) {
        const float sf = in15[15 * (size_t)i + 9];
        if (!(sf >= 0 && sf < scene->surface_count)) return fail(KY_ERR_INVALID_VALUE, "row %d: surface out of range", i);
    }
    return kat_run(device, in15, (size_t)n * 15 * 4, out6, (size_t)n * 6 * 4, [&](DeviceCtx* c, const float* d_in, float* d_out) {
        SceneSlot* sc; int r = upload_scene(c, scene, 0, &sc);
        if (r != KY_OK) return r;
        hipLaunchKernelGGL(kat_nee_kernel, dim3((n + 255) / 256), dim3(256), lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count), 0, (const DScene*)sc->d, direct_sample, light, d_in, n, d_out);
        return (int)KY_OK;
    });
}

int kyhip_kat_li_trace(int device, const ky_scene* scene, const ky_render_params* p, int x, int y, int s, float* rows26, int max_rows, float* li3) {
    if (!valid_params(p)) return fail(KY_ERR_INVALID_VALUE, "invalid render params");
    if (p->integrator != KY_INTEGRATOR_PATH_TRACING_ITERATION) return fail(KY_ERR_INVALID_VALUE, "the vertex trace follows path_tracing_iteration_t");
    if (!scene || !rows26 || max_rows <= 0 || max_rows > 4096 || s < 0 || x < 0 || y < 0 || x >= p->width || y >= p->height) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    const RenderConst rc = make_rc(p);
    const bool dbg = p->sampler == KY_SAMPLER_DEBUG;
    std::vector<float> host((size_t)4 + (size_t)max_rows * 26, 0.f);
    float dummy = 0.f;
    const int rcode = kat_run(device, &dummy, 4, host.data(), host.size() * 4, [&](DeviceCtx* c, const float*, float* d_out) {
        SceneSlot* sc; int r = upload_scene(c, scene, 0, &sc);
        if (r != KY_OK) return r;
        if (dbg) hipLaunchKernelGGL(kat_li_trace_kernel<true>, dim3(1), dim3(64), lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count), 0, (const DScene*)sc->d, rc, x, y, s, max_rows, d_out);
        else hipLaunchKernelGGL(kat_li_trace_kernel<false>, dim3(1), dim3(64), lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count), 0, (const DScene*)sc->d, rc, x, y, s, max_rows, d_out);
        return (int)KY_OK;
    });
    if (rcode != KY_OK) return rcode;
    const int n = (int)host[0];
    std::memcpy(rows26, host.data() + 4, (size_t)n * 26 * sizeof(float));
    if (li3) { li3[0] = host[1]; li3[1] = host[2]; li3[2] = host[3]; }
    return n;
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

static int smallpt_check(const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p) {
    if (!spheres || !p) return fail(KY_ERR_INVALID_VALUE, "null argument");
    if (p->variant != KY_SP_VARIANT_SMALLPT && p->variant != KY_SP_VARIANT_REWRITE) return fail(KY_ERR_INVALID_VALUE, "unknown smallpt variant %d", p->variant);
    if (n <= 0 || n > kysp::SP_MAX_SPHERES) return fail(KY_ERR_INVALID_VALUE, "1..%d spheres", kysp::SP_MAX_SPHERES);
    if (p->width <= 0 || p->height <= 0 || p->width > 16384 || p->height > 16384 || p->samps <= 0 || p->max_depth < 0)
        return fail(KY_ERR_INVALID_VALUE, "invalid smallpt params");
    for (int i = 0; i < n; ++i)
        if (spheres[i].refl < KY_SP_DIFF || spheres[i].refl > KY_SP_REFR || !(spheres[i].rad > 0)) return fail(KY_ERR_INVALID_VALUE, "sphere %d is invalid", i);
    return KY_OK;
}

int kyhip_smallpt_render(int device, const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p, double* image_rgb) {
    int rcode = smallpt_check(spheres, n, p);
    if (rcode != KY_OK) return rcode;
    if (!image_rgb) return fail(KY_ERR_INVALID_VALUE, "null image");
    DeviceCtx* c = nullptr;
    rcode = get_ctx(device, &c);
    if (rcode != KY_OK) return rcode;
    std::lock_guard<std::mutex> lock(c->m);
    kysp::SpSphere packed[kysp::SP_MAX_SPHERES];
    kysp::sp_pack(spheres, n, packed);
    kysp::SpConst k;
    kysp::sp_make_const(p, n, k);
    const size_t n_px = (size_t)p->width * p->height;
    DevBuf d_sph, d_sub, d_img;
    HIP_TRY(d_sph.alloc(sizeof(packed)));
    HIP_TRY(d_sub.alloc(n_px * 12 * sizeof(double)));
    HIP_TRY(d_img.alloc(n_px * 3 * sizeof(double)));
    HIP_TRY(hipMemcpy(d_sph.p, packed, sizeof(packed), hipMemcpyHostToDevice));
    const int blocks = ((p->width + 7) / 8) * ((p->height + 7) / 8);
    StreamState* st;
    rcode = get_stream_state(c, 0, &st);
    if (rcode != KY_OK) return rcode;
    HIP_TRY(hipEventRecord(st->ev0, 0));
    hipLaunchKernelGGL(kysp::smallpt_kernel, dim3(blocks), dim3(256), 0, 0, d_sph.as<kysp::SpSphere>(), k, d_sub.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(st->ev1, 0));
    st->timing_valid = true;
    c->last_launch = st;
    hipLaunchKernelGGL(kysp::smallpt_resolve_kernel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, 0, d_sub.as<double>(), d_img.as<double>(), p->width, p->height, p->variant);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(image_rgb, d_img.p, n_px * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return KY_OK;
}

int kyhip_smallpt_kat_radiance(int device, const ky_smallpt_sphere* spheres, int n_spheres, const ky_smallpt_params* p,
                               int x, int y, int sx, int sy, int s0, int n, double* out3) {
    int rcode = smallpt_check(spheres, n_spheres, p);
    if (rcode != KY_OK) return rcode;
    if (!out3 || n <= 0 || s0 < 0 || x < 0 || y < 0 || x >= p->width || y >= p->height || (sx | sy) < 0 || sx > 1 || sy > 1)
        return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    if (p->variant == KY_SP_VARIANT_REWRITE && (sx | sy) != 0) return fail(KY_ERR_INVALID_VALUE, "variant 1 has no subpixels: sx = sy = 0");
    DeviceCtx* c = nullptr;
    rcode = get_ctx(device, &c);
    if (rcode != KY_OK) return rcode;
    std::lock_guard<std::mutex> lock(c->m);
    kysp::SpSphere packed[kysp::SP_MAX_SPHERES];
    kysp::sp_pack(spheres, n_spheres, packed);
    kysp::SpConst k;
    kysp::sp_make_const(p, n_spheres, k);
    DevBuf d_sph, d_out;
    HIP_TRY(d_sph.alloc(sizeof(packed)));
    HIP_TRY(d_out.alloc((size_t)n * 3 * sizeof(double)));
    HIP_TRY(hipMemcpy(d_sph.p, packed, sizeof(packed), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kysp::smallpt_kat_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_sph.as<kysp::SpSphere>(), k, x, y, sx, sy, s0, n, d_out.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out3, d_out.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return KY_OK;
}

}  // extern "C"
