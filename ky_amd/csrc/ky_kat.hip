/*
 * ky_kat.hip -- function-level known-answer-test kernels and their C-ABI entry points (include/kyhip.h, "KAT entry points"): each runs ONE function
 * of the path -- a shape test, the camera, a BSDF lobe, a light, the scene traversal, an occlusion query, one estimator, one camera sample's
 * radiance with or without a vertex trace -- on caller-supplied inputs, for tests/test_parity_gpu.py to compare with the oracle.  Test surface
 * of the library, not part of a render; the kernels are the product's device functions (ky_device.hpp) called directly.
 */
#include <cstring>
#include <vector>

#include "ky_ctx.hpp"
#include "ky_render.hpp"

using namespace kyh;

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

template <bool BOXES>   // the scene has boxes (KY_FEAT_BOXES): the traversal the box kernels run
__global__ void kat_scene_intersect_kernel(const DScene* __restrict__ S_, const float* __restrict__ rays7, int n, float* __restrict__ out9) {
    const SceneRef S{S_, true, BOXES ? KY_FEAT_BOXES : 0, true};
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

// the environment estimate's pair scan (trace_any_pair): ray A unbounded, ray B up to tmax
template <bool BOXES>
__global__ void kat_any_pair_kernel(const DScene* __restrict__ S_, const float* __restrict__ in13, int n, float* __restrict__ out2) {
    const SceneRef S{S_, true, BOXES ? KY_FEAT_BOXES : 0, true};
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = in13 + 13 * (size_t)i;
    const AnyRay A{ld3(r), ld3(r + 3), K_INF}, B{ld3(r + 6), ld3(r + 9), r[12]};
    bool occ_a, occ_b;
    trace_any_pair(S, A, B, occ_a, occ_b);
    out2[2 * (size_t)i] = occ_a ? 1.f : 0.f;
    out2[2 * (size_t)i + 1] = occ_b ? 1.f : 0.f;
}

// BOXES: the launch these samples replay runs a kernel with the box traversal (render_uses_boxes): the replay takes the same one, so that "per sample what render() did" holds
// to the last decision (a ray along a box's edge may take the other face in the other traversal)
// single_env (a run-time flag: these kernels are not timed): ... whose both_mis estimate is estimate_env_both (KY_FEAT_SINGLE_ENV): the replay takes that too
template <bool DEBUG_SAMPLER, bool BOXES>
__global__ void kat_li_kernel(const DScene* __restrict__ S_, RenderConst rc, int x, int y, int s0, int n, float* __restrict__ out3, int single_env) {
    const SceneRef S{S_, true, (BOXES ? KY_FEAT_BOXES : 0) | (single_env ? KY_FEAT_SINGLE_ENV : 0), true};
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
template <bool DEBUG_SAMPLER, bool BOXES>
__global__ void kat_li_trace_kernel(const DScene* __restrict__ S_, RenderConst rc, int x, int y, int s, int max_rows, float* __restrict__ out, int single_env) {
    const SceneRef S{S_, true, (BOXES ? KY_FEAT_BOXES : 0) | (single_env ? KY_FEAT_SINGLE_ENV : 0), true};
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

extern "C" {

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
        if (sc->h->feat & KY_FEAT_BOXES)
            hipLaunchKernelGGL(kat_scene_intersect_kernel<true>, dim3((n + 255) / 256), dim3(256), lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count), 0, (const DScene*)sc->d, d_in, n, d_out);
        else
            hipLaunchKernelGGL(kat_scene_intersect_kernel<false>, dim3((n + 255) / 256), dim3(256), lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count), 0, (const DScene*)sc->d, d_in, n, d_out);
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
int kyhip_kat_any_pair(int device, const ky_scene* scene, const float* in13, int n, float* out2) {
    if (!scene || !in13 || !out2 || n <= 0) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    return kat_run(device, in13, (size_t)n * 13 * 4, out2, (size_t)n * 2 * 4, [&](DeviceCtx* c, const float* d_in, float* d_out) {
        SceneSlot* sc; int r = upload_scene(c, scene, 0, &sc);
        if (r != KY_OK) return r;
        if (sc->h->feat & KY_FEAT_BOXES) hipLaunchKernelGGL(kat_any_pair_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, 0, (const DScene*)sc->d, d_in, n, d_out);
        else hipLaunchKernelGGL(kat_any_pair_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, 0, (const DScene*)sc->d, d_in, n, d_out);
        return (int)KY_OK;
    });
}
int kyhip_kat_occluded_between(int device, const ky_scene* scene, int light, const float* in9, int n, float* out1) {
    if (light < -1) return fail(KY_ERR_INVALID_VALUE, "light %d out of range", light);
    return kat_occluded_impl(device, scene, in9, n, out1, light);
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
        const size_t lds = lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count);
        const dim3 grid((n + 255) / 256), block(256);
        const int rf = render_replay_feat(scene, p, sc->h);
        const bool boxes = (rf & KY_FEAT_BOXES) != 0;
        const int env = (rf & KY_FEAT_SINGLE_ENV) ? 1 : 0;
        if (dbg && boxes) hipLaunchKernelGGL((kat_li_kernel<true, true>), grid, block, lds, 0, (const DScene*)sc->d, rc, x, y, s0, n, d_out, env);
        else if (dbg) hipLaunchKernelGGL((kat_li_kernel<true, false>), grid, block, lds, 0, (const DScene*)sc->d, rc, x, y, s0, n, d_out, env);
        else if (boxes) hipLaunchKernelGGL((kat_li_kernel<false, true>), grid, block, lds, 0, (const DScene*)sc->d, rc, x, y, s0, n, d_out, env);
        else hipLaunchKernelGGL((kat_li_kernel<false, false>), grid, block, lds, 0, (const DScene*)sc->d, rc, x, y, s0, n, d_out, env);
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
        const size_t lds = lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count);
        const int rf = render_replay_feat(scene, p, sc->h);
        const bool boxes = (rf & KY_FEAT_BOXES) != 0;
        const int env = (rf & KY_FEAT_SINGLE_ENV) ? 1 : 0;
        if (dbg && boxes) hipLaunchKernelGGL((kat_li_trace_kernel<true, true>), dim3(1), dim3(64), lds, 0, (const DScene*)sc->d, rc, x, y, s, max_rows, d_out, env);
        else if (dbg) hipLaunchKernelGGL((kat_li_trace_kernel<true, false>), dim3(1), dim3(64), lds, 0, (const DScene*)sc->d, rc, x, y, s, max_rows, d_out, env);
        else if (boxes) hipLaunchKernelGGL((kat_li_trace_kernel<false, true>), dim3(1), dim3(64), lds, 0, (const DScene*)sc->d, rc, x, y, s, max_rows, d_out, env);
        else hipLaunchKernelGGL((kat_li_trace_kernel<false, false>), dim3(1), dim3(64), lds, 0, (const DScene*)sc->d, rc, x, y, s, max_rows, d_out, env);
        return (int)KY_OK;
    });
    if (rcode != KY_OK) return rcode;
    const int n = (int)host[0];
    std::memcpy(rows26, host.data() + 4, (size_t)n * 26 * sizeof(float));
    if (li3) { li3[0] = host[1]; li3[1] = host[2]; li3[2] = host[3]; }
    return n;
}

}  // extern "C"
