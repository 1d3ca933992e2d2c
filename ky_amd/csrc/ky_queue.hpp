/*
 * ky_queue.hpp -- the "queue" engine for path_tracing_iteration_t::Li (ky.cpp:4529-4617).
 *
 * The lane engine (render_kernel: ky_render.hpp, launched by ky_launch.hip) walks one path per lane; lanes whose path has ended, whose vertex is a
 * delta lobe or whose light sample is dead idle through the phases they do not need (measured lane occupancy 0.60).
 * Here a workgroup keeps the state of QE_SLOTS paths in LDS (SoA) and every path is a small state machine:
 *
 *      REGEN  -> TRACE                       new camera sample (3712-3715) for a free slot
 *      TRACE  -> NEE | CONT | REGEN          scene->intersect (4542), emission (4548-4559), termination (4563), lobe pick (2663)
 *      NEE    -> SHADOW | NEE | CONT         one light of sample_all_light (3834-3872): the BSDF-sampling estimator completely,
 *                                            the light-sampling estimator up to its shadow ray
 *      SHADOW -> NEE | CONT                  scene->occluded (3187-3206) for the pending light sample
 *      CONT   -> TRACE | REGEN               BSDF sample, beta, roulette (4586-4612)
 *
 * One ring queue of slot numbers per state lives next to the path pool.  A WAVEFRONT repeatedly takes the fullest queue,
 * pops up to 64 slots from it (one CAS on the queue head), runs that state's code for them with all lanes busy, and pushes
 * every slot into the queue of its next state (ballot + popcount prefix, one LDS atomic per wave and destination).  There
 * is no workgroup barrier in the loop: waves run different states at the same time and never wait for each other.
 *
 * A path's arithmetic and its random numbers are those of the lane engine, executed strictly in sequence by whichever
 * lanes pick the slot up, and finished samples are added to the film accumulator in fixed point (integer atomics), so
 * the image does not depend on scheduling, tiling or GPU count.  It differs from the lane engine's image only by float
 * summation order (per-sample instead of per-chunk accumulation): tests/test_parity_gpu.py::test_engines_agree.
 */
#pragma once
#include "ky_device.hpp"

namespace kyd {

constexpr int QE_THREADS = 512;   // 8 waves; two workgroups per CU
#ifndef KY_QE_SLOTS
#define KY_QE_SLOTS 640
#endif
constexpr int QE_SLOTS = KY_QE_SLOTS;
constexpr int QE_RING = 1024;     // queue capacity, a power of two > QE_SLOTS (a slot is in at most one queue)
constexpr int QE_ITEMS = 16;      // fetched work items a workgroup remembers: (64 + QE_PREFETCH) / 64 + 1 = 10 can be ahead of the cursor, the rest is margin
constexpr unsigned QE_PREFETCH = 512;   // camera samples fetched ahead of need
#ifndef KY_QE_POLLS
#define KY_QE_POLLS 12
#endif
enum : int { QS_REGEN = 0, QS_TRACE, QS_NEE, QS_SHADOW, QS_CONT, QS_COUNT };
constexpr unsigned short QE_EMPTY = 0xffffu;
constexpr int QE_SPIN_CAP = 1 << 22;

struct QeItem {
    int x0, y0, pix0, s_begin;
    unsigned n_units, u0;   // the item covers camera samples ("units") [u0, u0 + n_units) of the workgroup's sequence
};

struct QeLds {
    // path state: TRACE reads (o, d); after it o holds the vertex position
    float ox[QE_SLOTS], oy[QE_SLOTS], oz[QE_SLOTS], dx[QE_SLOTS], dy[QE_SLOTS], dz[QE_SLOTS];
    float br[QE_SLOTS], bg[QE_SLOTS], bb[QE_SLOTS], lr[QE_SLOTS], lg[QE_SLOTS], lb[QE_SLOTS];
    uint32_t rs[QE_SLOTS], ri[QE_SLOTS];
    uint32_t info[QE_SLOTS];   // bits 0-1 lobe | bit 2 prev_specular | bits 8-15 bounces | bits 16-23 surface | bits 24-31 next light
    int pix[QE_SLOTS];
    // the pending light sample: shadow ray direction, tmax, and what it adds to Lo when unoccluded
    float wx[QE_SLOTS], wy[QE_SLOTS], wz[QE_SLOTS], wt[QE_SLOTS], cr[QE_SLOTS], cg[QE_SLOTS], cb[QE_SLOTS];
    unsigned short ring[QS_COUNT][QE_RING];
    unsigned head[QS_COUNT], tail[QS_COUNT];
    int busy;                  // waves that hold a batch
    // camera-sample pool
    QeItem items[QE_ITEMS];
    unsigned n_items, units_fetched, unit_cursor;
    int exhausted, fetch_lock;
};

#ifdef KY_QE_STATS
__device__ unsigned long long g_qe_stats[32];   // [q] batches, [8+q] lanes, [16+q] clocks in the state's code, [24] clocks in acquire, [25] clocks in push
#define QE_STAT(i, v) do { } while (0)
#define QE_CLOCK() __builtin_amdgcn_s_memtime()
#else
#define QE_STAT(i, v) do { } while (0)
#define QE_CLOCK() 0ull
#endif

KY_DEV unsigned lds_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
KY_DEV int lds_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// L = L + Li * (1. / spp) (3717-3721) for one finished sample, straight into the pixel's fixed-point accumulator
KY_DEV void film_add_sample(unsigned long long* __restrict__ accum, unsigned* __restrict__ flags, int pix, f3 L) {
    const float v[3] = {L.x, L.y, L.z};
    unsigned fl = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float a = v[ch];
        if (a != a) fl |= 1u << ch;               // NaN
        else if (a > 2.0e9f) fl |= 8u << ch;      // +inf (or beyond the accumulator's range)
        else if (a < -2.0e9f) fl |= 64u << ch;    // -inf
        else if (a != 0.f) atomicAdd(&accum[(size_t)pix * 3 + ch], (unsigned long long)to_fixed32(a));
    }
    if (fl) atomicOr(&flags[pix], fl);
}

// Take a batch: returns the state to run (wave-uniform; -1 = nothing left, leave) and this lane's slot (-1 = no slot).
// Lane q looks at queue q; the choice is made on the scalar unit.
KY_DEV int qe_acquire(QeLds& W, int lane, int& slot) {
    int stage = -1;
    unsigned base = 0, n = 0;
    int polls = 0;
    for (int spins = 0; spins < QE_SPIN_CAP; ++spins) {   // the cap only guards against a scheduling bug hanging the GPU
        const int busy = __builtin_amdgcn_readfirstlane(lds_load(&W.busy));
        unsigned h = 0, a = 0;
        if (lane < QS_COUNT) {
            h = lds_load(&W.head[lane]);
            a = lds_load(&W.tail[lane]) - h;
        }
        int best = 0;
        unsigned best_n = 0;
#pragma unroll
        for (int q = 0; q < QS_COUNT; ++q) {   // the fullest queue; later states win ties
            const unsigned aq = (unsigned)__builtin_amdgcn_readlane((int)a, q);
            if (aq >= best_n) { best = q; best_n = aq; }
        }
        if (best_n == 0) {
            if (busy == 0) break;             // nothing queued and nobody who could queue something
            QE_STAT(17, 1);
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        if (best_n < 64u && busy > 0 && polls < KY_QE_POLLS) {   // a fuller batch is probably on its way
            ++polls;
            QE_STAT(16, 1);
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        const unsigned take = best_n < 64u ? best_n : 64u;
        const unsigned best_h = (unsigned)__builtin_amdgcn_readlane((int)h, best);
        unsigned old = ~best_h;
        if (lane == 0) {
            atomicAdd(&W.busy, 1);
            old = atomicCAS(&W.head[best], best_h, best_h + take);
            if (old != best_h) atomicSub(&W.busy, 1);
        }
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)old) == best_h) {
            stage = best; base = best_h; n = take;
            QE_STAT(best, 1); QE_STAT(8 + best, take);
            break;
        }
        QE_STAT(18, 1);
    }
    slot = -1;
    if (stage >= 0 && (unsigned)lane < n) {
        unsigned short* e = &W.ring[stage][(base + (unsigned)lane) & (QE_RING - 1)];
        unsigned v = QE_EMPTY;
        // the producer has reserved this entry but may not have written it yet
        for (int spins = 0; spins < QE_SPIN_CAP && v == QE_EMPTY; ++spins) v = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(e, QE_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        slot = v == QE_EMPTY ? -1 : (int)v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return stage;
}

KY_DEV void qe_push(QeLds& W, int lane, unsigned long long lanes_below, int q, bool want, int slot) {
    const unsigned long long m = __ballot(want);
    if (!m) return;
    const int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(&W.tail[q], (unsigned)__popcll(m));
    base = __shfl(base, leader);
    if (want) {
        __hip_atomic_store(&W.ring[q][(base + (unsigned)__popcll(m & lanes_below)) & (QE_RING - 1)], (unsigned short)slot, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// Hand out the next `n` camera samples of the workgroup's sequence (wave-uniform); returns the first one.
// Fetches work items from the global queue when the sequence runs short.
KY_DEV unsigned qe_take_units(QeLds& W, int lane, unsigned n, unsigned* __restrict__ counter, const ShardConst& sh, int spp) {
    unsigned ubase = 0;
    if (lane == 0) {
        ubase = atomicAdd(&W.unit_cursor, n);
        const unsigned want = ubase + n + QE_PREFETCH;
        if ((int)(lds_load(&W.units_fetched) - want) < 0 && !lds_load(&W.exhausted)) {
            for (int spins = 0; spins < QE_SPIN_CAP && atomicCAS(&W.fetch_lock, 0, 1) != 0; ++spins) __builtin_amdgcn_s_sleep(2);
            while (!lds_load(&W.exhausted) && (int)(lds_load(&W.units_fetched) - want) < 0) {
                const unsigned id = atomicAdd(counter, 1u);
                if (id >= sh.n_items) { __hip_atomic_store(&W.exhausted, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
                const int c = (int)(id / (unsigned)sh.n_blocks), b = (int)(id % (unsigned)sh.n_blocks);   // chunk-major
                const int k = b / sh.blocks_per_tile, inner = b % sh.blocks_per_tile;
                const int bx = inner % sh.blocks_w, by = inner / sh.blocks_w;
                const int tile = sh.tile_first + k * sh.tile_step;
                const int trow = tile / sh.tiles_x, tcol = (tile % sh.tiles_x + trow) % sh.tiles_x;   // rotated rows
                int s_begin, s_end;
                chunk_range(chunk_plan(spp), c, s_begin, s_end);
                const unsigned u0 = lds_load(&W.units_fetched), ni = lds_load(&W.n_items);
                QeItem* it = &W.items[ni % QE_ITEMS];
                __hip_atomic_store(&it->n_units, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                it->x0 = tcol * sh.tile_w + bx * 8;
                it->y0 = trow * sh.tile_h + by * 8;
                it->pix0 = (k * sh.tile_h + by * 8) * sh.tile_w + bx * 8;
                it->s_begin = s_begin;
                it->u0 = u0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&it->n_units, 64u * (unsigned)(s_end - s_begin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&W.n_items, ni + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&W.units_fetched, u0 + 64u * (unsigned)(s_end - s_begin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            atomicExch(&W.fetch_lock, 0);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return __builtin_amdgcn_readfirstlane(ubase);
}

// The light-sampling estimator (by_emitter 3933-3962 / by_emitter_mis 4035-4074) up to its shadow ray: returns true when
// a shadow ray is pending, with its direction / tmax and the estimator's value if the ray turns out unoccluded.
// The reference tests occlusion first and evaluates the BSDF only for unoccluded samples; evaluating first drops the shadow
// rays of samples whose f*cos is black (their Ld is 0 either way).
template <bool MIS>
KY_DEV bool emitter_sample(SceneRef S, const Vertex& v, f3 wo, int li, float u0, float u1, f3& dir, float& tmax, f3& Ld) {
    const DLight& L = S->light[li];
    const LightSample ls = light_sample_Li(L, v.position, v.normal, u0, u1);
    const bool dead = is_black(ls.Li) || (MIS ? (ls.pdf <= 0) : (ls.pdf == 0));
    if (dead) return false;
    const f3 to = ls.position - v.position;   // scene_t::occluded(isect, ls.position), 3187-3201
    const float d2 = length_sq(to);
    const float inv_d = rsq(d2);
    dir = to * inv_d;
    tmax = d2 * inv_d - 2e-3f;
    f3 f;
    float bsdf_pdf, abs_cos_i;
    bsdf_eval_pdf(v, wo, ls.wi, f, bsdf_pdf, abs_cos_i);
    const f3 f_cos = f * abs_cos_i;
    if (is_black(f_cos)) return false;
    const bool delta_light = L.kind == KY_LIGHT_POINT || L.kind == KY_LIGHT_DIRECTION;
    if (!MIS || delta_light) Ld = (f_cos * ls.Li) * rcp(ls.pdf);          // 3956 / 4057
    else Ld = (f_cos * ls.Li) * (2.f * rcp(ls.pdf + bsdf_pdf));            // 4070
    return true;
}

// One light of sample_all_light (3834-3872).  Wave-uniform call.  L_now: what the light adds right away (the BSDF-sampling
// estimator); pending / dir / tmax / L_pending: the light-sampling estimator's shadow ray and its value.
template <bool DEBUG_SAMPLER>
KY_DEV void nee_one_light(SceneRef S, const LdsScene& Lds, const Vertex& v, f3 wo, Sampler& smp, int strategy, int li, bool active,
                          f3& L_now, bool& pending, f3& dir, float& tmax, f3& L_pending) {
    L_now = mk3(0, 0, 0);
    L_pending = mk3(0, 0, 0);
    dir = mk3(0, 0, 1);
    tmax = 0.f;
    pending = false;
    // the reference's GCC build draws random_bsdf first, then random_light (3866-3868), for every light and strategy
    float ub0 = 0.f, ub1 = 0.f, ul0 = 0.f, ul1 = 0.f;
    if (active) {
        ub0 = sampler_next<DEBUG_SAMPLER>(smp); ub1 = sampler_next<DEBUG_SAMPLER>(smp);
        ul0 = sampler_next<DEBUG_SAMPLER>(smp); ul1 = sampler_next<DEBUG_SAMPLER>(smp);
    }
    if (strategy == KY_DIRECT_BOTH_MIS) {  // 4076-4088
        estimate_by_bsdf<true>(S, Lds, v, wo, li, ub0, ub1, active, L_now, mk3(0.5f, 0.5f, 0.5f));   // adds w x the estimate to L_now (zero above)
        if (active) {
            f3 Ll = mk3(0, 0, 0);
            pending = emitter_sample<true>(S, v, wo, li, ul0, ul1, dir, tmax, Ll);
            L_pending = 0.5f * Ll;
        }
    } else if (strategy == KY_DIRECT_BSDF_MIS) {
        estimate_by_bsdf<true>(S, Lds, v, wo, li, ub0, ub1, active, L_now, mk3(1, 1, 1));
    } else if (strategy == KY_DIRECT_LIGHT_MIS) {
        if (active) pending = emitter_sample<true>(S, v, wo, li, ul0, ul1, dir, tmax, L_pending);
    } else if (strategy == KY_DIRECT_LIGHT) {
        if (active) pending = emitter_sample<false>(S, v, wo, li, ul0, ul1, dir, tmax, L_pending);
    } else if (strategy == KY_DIRECT_BSDF) {
        const int lk = S->light[li].kind;
        if (!(lk == KY_LIGHT_POINT || lk == KY_LIGHT_DIRECTION)) {  // the third float2 is drawn after the delta test (3894-3900)
            float u0 = 0.f, u1 = 0.f;
            if (active) { u0 = sampler_next<DEBUG_SAMPLER>(smp); u1 = sampler_next<DEBUG_SAMPLER>(smp); }
            estimate_by_bsdf<false>(S, Lds, v, wo, li, u0, u1, active, L_now, mk3(1, 1, 1));
        }
    }
    if (!pending) L_pending = mk3(0, 0, 0);
}

// the vertex a slot holds after TRACE, rebuilt from (position, incoming direction, surface, lobe)
KY_DEV void qe_vertex(Vertex& v, const LdsScene& Lds, f3 position, f3 d, int surface, int lobe) {
    v.t = 0.f;
    v.position = position;
    v.surface = surface;
    v.normal = hit_normal(Lds.hit[surface], position, d);
    v.bsdf = make_bsdf_for_lobe(Lds.mat[Lds.hit[surface].material], lobe);
    vertex_prepare(v, -d, &Lds.hit[surface]);   // the lobe's basis in registers (in_lds is false here)
}

#ifndef KY_QE_WAVES
#define KY_QE_WAVES 4
#endif

// STRATEGY >= 0 fixes direct_sample_enum at compile time; -1 reads rc.strategy.  rc.integrator is path_tracing_iteration.
template <bool DEBUG_SAMPLER, int STRATEGY>
__global__ __launch_bounds__(QE_THREADS, KY_QE_WAVES) void render_kernel_q(const DScene* __restrict__ S, RenderConst rc, ShardConst sh,
                                                                            unsigned* __restrict__ counter, unsigned long long* __restrict__ accum,
                                                                            unsigned* __restrict__ flags) {
    __shared__ QeLds W;
    if (STRATEGY >= 0) rc.strategy = STRATEGY;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
    for (int i = tid; i < QS_COUNT * QE_RING; i += QE_THREADS) (&W.ring[0][0])[i] = QE_EMPTY;
    __syncthreads();
    for (int i = tid; i < QE_SLOTS; i += QE_THREADS) W.ring[QS_REGEN][i] = (unsigned short)i;   // every slot starts free
    if (tid < QS_COUNT) { W.head[tid] = 0; W.tail[tid] = tid == QS_REGEN ? QE_SLOTS : 0; }
    if (tid < QE_ITEMS) { W.items[tid].n_units = 0; W.items[tid].u0 = 0; }
    if (tid == 0) { W.busy = 0; W.n_items = 0; W.units_fetched = 0; W.unit_cursor = 0; W.exhausted = 0; W.fetch_lock = 0; }
    const LdsScene Lds = stage_scene<false>(S);   // ends with a barrier (static block: the queue engine takes scenes of up to 64 surfaces)

    const int n_lights = S->n_lights;
    const bool nee_on = n_lights > 0;   // estimate_direct_lighting_idle (3880) still consumes its four numbers per light

#ifdef KY_QE_STATS
    unsigned long long st_cycles[QS_COUNT + 2] = {}, st_batches[QS_COUNT] = {}, st_lanes[QS_COUNT] = {};
#endif
    for (;;) {
        int slot;
        [[maybe_unused]] const unsigned long long clk0 = QE_CLOCK();
        const int stage = qe_acquire(W, lane, slot);   // wave-uniform
        [[maybe_unused]] const unsigned long long clk1 = QE_CLOCK();
        if (stage < 0) break;
        const bool active = slot >= 0;
        const int sl = active ? slot : 0;
        int dest = -1;

        if (stage == QS_REGEN) {
            // ---- next camera sample of the workgroup's sequence, 3712-3715 ----
            const unsigned long long am = __ballot(active);
            const unsigned u = qe_take_units(W, lane, (unsigned)__popcll(am), counter, sh, rc.spp) + (unsigned)__popcll(am & lanes_below);
            if (active) {
                int found = -1;
                unsigned r = 0;
#pragma unroll
                for (int k = 0; k < QE_ITEMS; ++k) {
                    const unsigned rk = u - W.items[k].u0;
                    if (rk < W.items[k].n_units) { found = k; r = rk; }
                }
                if (found >= 0) {
                    const QeItem& it = W.items[found];
                    const int p = (int)(r & 63u), px = p & 7, py = p >> 3;
                    const int x = it.x0 + px, y = it.y0 + py;
                    dest = QS_REGEN;   // a pixel of the 8x8 block outside the film: take another sample
                    if (x < rc.width && y < rc.height) {
                        PathState ps;
                        path_begin<DEBUG_SAMPLER>(ps, S, sampler_pixel_key(rc.seed, (uint32_t)(y * rc.width + x)), x, y, it.s_begin + (int)(r >> 6));
                        W.ox[sl] = ps.o.x; W.oy[sl] = ps.o.y; W.oz[sl] = ps.o.z;
                        W.dx[sl] = ps.d.x; W.dy[sl] = ps.d.y; W.dz[sl] = ps.d.z;
                        W.br[sl] = 1.f; W.bg[sl] = 1.f; W.bb[sl] = 1.f;
                        W.lr[sl] = 0.f; W.lg[sl] = 0.f; W.lb[sl] = 0.f;
                        W.rs[sl] = ps.smp.s0; W.ri[sl] = ps.smp.s1;
                        W.info[sl] = 0;
                        W.pix[sl] = it.pix0 + py * sh.tile_w + px;
                        dest = QS_TRACE;
                    }
                }
                // found < 0: the global queue is exhausted; the slot retires
            }
        } else if (stage == QS_TRACE) {
            // ---- scene->intersect (4542), emission (4548-4559), termination (4563), the material's lobe (3083, 2663) ----
            f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
            if (active) {
                o = mk3(W.ox[sl], W.oy[sl], W.oz[sl]);
                d = mk3(W.dx[sl], W.dy[sl], W.dz[sl]);
            }
            float t = K_INF;
            int hs = -1;
            if (active) hs = trace_nearest(S, o, d, t);
            if (active) {
                const uint32_t inf = W.info[sl];
                const int bounces = (int)((inf >> 8) & 0xffu);
                const bool prev_specular = (inf & 4u) != 0;
                const bool hit = hs >= 0;
                f3 position = o, emission = mk3(0, 0, 0);
                if (hit) {
                    position = o + t * d;
                    emission = surface_emission(Lds, hs, hit_normal(Lds.hit[hs], position, d), -d);
                }
                f3 Lo = mk3(W.lr[sl], W.lg[sl], W.lb[sl]);
                if (bounces == 0 || prev_specular) {
                    const f3 env = S->env_light >= 0 ? ld3(S->light[S->env_light].color) : mk3(0, 0, 0);  // environment_lighting, 3231
                    const f3 beta = mk3(W.br[sl], W.bg[sl], W.bb[sl]);
                    Lo = Lo + beta * (hit ? emission : env);
                }
                if (!hit || bounces >= rc.max_path_depth) {
                    film_add_sample(accum, flags, W.pix[sl], Lo * rc.inv_spp);
                    dest = QS_REGEN;
                } else {
                    const DMat& M = Lds.mat[Lds.hit[hs].material];
                    int lobe;
                    if (M.kind == KY_MATERIAL_PLASTIC) {
                        Sampler smp{W.rs[sl], W.ri[sl]};
                        lobe = pick_lobe(M, sampler_next<DEBUG_SAMPLER>(smp));
                        W.rs[sl] = smp.s0; W.ri[sl] = smp.s1;
                    } else {
                        lobe = pick_lobe(M, 0.f);
                    }
                    W.ox[sl] = position.x; W.oy[sl] = position.y; W.oz[sl] = position.z;
                    W.lr[sl] = Lo.x; W.lg[sl] = Lo.y; W.lb[sl] = Lo.z;
                    W.info[sl] = (uint32_t)lobe | (inf & 0x0000ff04u) | ((uint32_t)hs << 16);   // next light = 0
                    const bool delta = lobe == LOBE_MIRROR || lobe == LOBE_GLASS;
                    dest = (delta || !nee_on) ? QS_CONT : QS_NEE;   // 4571
                }
            }
        } else if (stage == QS_NEE) {
            // ---- one light of sample_all_light (4575) ----
            Vertex v;
            Sampler smp{0u, 1u};
            uint32_t inf = 0;
            f3 d = mk3(0, 0, 1);
            if (active) {
                inf = W.info[sl];
                d = mk3(W.dx[sl], W.dy[sl], W.dz[sl]);
                qe_vertex(v, Lds, mk3(W.ox[sl], W.oy[sl], W.oz[sl]), d, (int)((inf >> 16) & 0xffu), (int)(inf & 3u));
                smp.s0 = W.rs[sl]; smp.s1 = W.ri[sl];
            }
            // lanes of a batch can be at different lights; one wave-uniform pass per light present in the batch
            const int my_light = (int)(inf >> 24);
            f3 L_now = mk3(0, 0, 0), L_pending = mk3(0, 0, 0), sdir = mk3(0, 0, 1);
            float stmax = 0.f;
            bool pending = false;
            for (int li = 0; li < n_lights; ++li) {
                const bool mine = active && my_light == li;
                if (!__any(mine)) continue;
                f3 a, b, c;
                float tm;
                bool pe;
                nee_one_light<DEBUG_SAMPLER>(S, Lds, v, -d, smp, rc.strategy, li, mine, a, pe, c, tm, b);
                if (mine) { L_now = a; L_pending = b; sdir = c; stmax = tm; pending = pe; }
            }
            if (active) {
                const f3 beta = mk3(W.br[sl], W.bg[sl], W.bb[sl]);
                if (L_now.x != 0.f || L_now.y != 0.f || L_now.z != 0.f) {
                    W.lr[sl] += beta.x * L_now.x; W.lg[sl] += beta.y * L_now.y; W.lb[sl] += beta.z * L_now.z;
                }
                W.rs[sl] = smp.s0; W.ri[sl] = smp.s1;
                W.info[sl] = (inf & 0x00ffffffu) | ((uint32_t)(my_light + 1) << 24);
                if (pending) {
                    W.wx[sl] = sdir.x; W.wy[sl] = sdir.y; W.wz[sl] = sdir.z; W.wt[sl] = stmax;
                    W.cr[sl] = beta.x * L_pending.x; W.cg[sl] = beta.y * L_pending.y; W.cb[sl] = beta.z * L_pending.z;
                    dest = QS_SHADOW;
                } else {
                    dest = my_light + 1 < n_lights ? QS_NEE : QS_CONT;
                }
            }
        } else if (stage == QS_SHADOW) {
            // ---- scene->occluded for the pending light sample, 3187-3206 ----
            bool occluded = true;
            uint32_t inf = 0;
            if (active) {
                inf = W.info[sl];
                const f3 position = mk3(W.ox[sl], W.oy[sl], W.oz[sl]);
                const f3 normal = hit_normal(Lds.hit[(inf >> 16) & 0xffu], position, mk3(W.dx[sl], W.dy[sl], W.dz[sl]));
                const f3 dir = mk3(W.wx[sl], W.wy[sl], W.wz[sl]);
                occluded = trace_any(S, S->trav, offset_ray_origin(position, normal, dir), dir, W.wt[sl]);
            }
            if (active) {
                if (!occluded) { W.lr[sl] += W.cr[sl]; W.lg[sl] += W.cg[sl]; W.lb[sl] += W.cb[sl]; }
                dest = (int)(inf >> 24) < n_lights ? QS_NEE : QS_CONT;
            }
        } else {
            // ---- sample the BSDF for the next direction, beta, roulette, 4586-4612 ----
            if (active) {
                const uint32_t inf = W.info[sl];
                Vertex v;
                const f3 d = mk3(W.dx[sl], W.dy[sl], W.dz[sl]);
                qe_vertex(v, Lds, mk3(W.ox[sl], W.oy[sl], W.oz[sl]), d, (int)((inf >> 16) & 0xffu), (int)(inf & 3u));
                Sampler smp{W.rs[sl], W.ri[sl]};
                const float u0 = sampler_next<DEBUG_SAMPLER>(smp), u1 = sampler_next<DEBUG_SAMPLER>(smp);
                const BsdfContinue bs = bsdf_continue(v, -d, u0, u1);   // direction + f |cos| / pdf in one factor (ky_device.hpp)
                bool ended = !bs.ok;   // 4588
                if (!ended) {
                    f3 beta = mk3(W.br[sl], W.bg[sl], W.bb[sl]);
                    beta = beta * bs.weight;  // 4592
                    const bool specular = bs.specular;                                    // 4596
                    const f3 o = offset_ray_origin(v.position, v.normal, bs.wi);         // 4597
                    int bounces = (int)((inf >> 8) & 0xffu);
                    if (bounces > 3) {  // Russian roulette, 4601-4612
                        const float q = fmaxf(0.05f, 1 - max3(beta));
                        const float u = sampler_next<DEBUG_SAMPLER>(smp);
                        if (u < q) ended = true;
                        beta = beta * rcp(1 - q);
                    }
                    bounces += 1;
                    // the vertex at bounces == max_depth can only add emission after a delta bounce (4548, 4563)
                    if (bounces >= rc.max_path_depth && !specular) ended = true;
                    if (!ended) {
                        W.ox[sl] = o.x; W.oy[sl] = o.y; W.oz[sl] = o.z;
                        W.dx[sl] = bs.wi.x; W.dy[sl] = bs.wi.y; W.dz[sl] = bs.wi.z;
                        W.br[sl] = beta.x; W.bg[sl] = beta.y; W.bb[sl] = beta.z;
                        W.rs[sl] = smp.s0; W.ri[sl] = smp.s1;
                        W.info[sl] = (specular ? 4u : 0u) | ((uint32_t)bounces << 8);
                        dest = QS_TRACE;
                    }
                }
                if (ended) {
                    film_add_sample(accum, flags, W.pix[sl], mk3(W.lr[sl], W.lg[sl], W.lb[sl]) * rc.inv_spp);
                    dest = QS_REGEN;
                }
            }
        }

        [[maybe_unused]] const unsigned long long clk2 = QE_CLOCK();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the slot's state before its number
#pragma unroll
        for (int q = 0; q < QS_COUNT; ++q) qe_push(W, lane, lanes_below, q, dest == q, slot);
        if (lane == 0) atomicSub(&W.busy, 1);
#ifdef KY_QE_STATS
        const unsigned long long clk3 = QE_CLOCK();
#pragma unroll
        for (int q = 0; q < QS_COUNT; ++q)
            if (stage == q) { st_cycles[q] += clk2 - clk1; st_batches[q] += 1; st_lanes[q] += __popcll(__ballot(active)); }
        st_cycles[QS_COUNT] += clk1 - clk0;
        st_cycles[QS_COUNT + 1] += clk3 - clk2;
#endif
    }
#ifdef KY_QE_STATS
    if (lane == 0) {
        for (int q = 0; q < QS_COUNT; ++q) {
            atomicAdd(&g_qe_stats[q], st_batches[q]); atomicAdd(&g_qe_stats[8 + q], st_lanes[q]); atomicAdd(&g_qe_stats[16 + q], st_cycles[q]);
        }
        atomicAdd(&g_qe_stats[24], st_cycles[QS_COUNT]); atomicAdd(&g_qe_stats[25], st_cycles[QS_COUNT + 1]);
    }
#endif
}

}  // namespace kyd
