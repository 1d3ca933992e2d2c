/*
 * ky_render.hpp -- the lane engine's render kernel (persistent workgroups, one lane = one path) and the chunk schedule it shares with the host.
 *
 * Its own header since round 4 because it is compiled twice: into libkyhip.so's table of instantiations (ky_launch.hip, g_variants) and, at run time,
 * by hiprtc for the exact instantiation a scene admits (ky_jit.cpp, "run-time instantiations": the library carries this file, ky_device.hpp and
 * include/kyhip.h as text).  Device code and __host__ __device__ arithmetic only; nothing here touches the HIP runtime API.
 */
#pragma once
#include "ky_device.hpp"
#include "ky_shard.hpp"   // the chunk schedule and ShardConst (shared with the host)

using namespace kyd;

#ifndef KY_MAX_RETRACE
#define KY_MAX_RETRACE 1
#endif
#ifndef KY_REFILL_BATCH
#define KY_REFILL_BATCH 6     // lanes that make a work-item refill pass worth running at once ...
#endif
#ifndef KY_REFILL_WAIT
#define KY_REFILL_WAIT 4      // ... or turns the first of them waits at most
#endif
#ifndef KY_RETRACE_THRESHOLD
#define KY_RETRACE_THRESHOLD 80
#endif
#ifndef KY_WAVES_PER_EU
#define KY_WAVES_PER_EU 7           // the hot instantiation <false, both_mis, feat 7>: 72 VGPRs, nothing spilled (six: 80; +3.5 % for the seventh wavefront)
#endif
#ifndef KY_WAVES_PER_EU_QUEUE
#define KY_WAVES_PER_EU_QUEUE 6     // the instantiations with deferred shadow rays and no scene facts: round 4 (LDS block 23.0 KB since the chunk sum left it): 80 VGPRs with 20 spilled
                                    // beat 96 with 6 at five by 3.4 % on the Cornell box with lamp and point light (20.35 against 21.05 ms at 256 spp); seven: 20.43
#endif
#ifndef KY_WAVES_PER_EU_QUEUE_FEAT
#define KY_WAVES_PER_EU_QUEUE_FEAT 7   // ... with scene facts (the sphere-lights kernel): round 4, with its LDS block at 22.9 KB (KY_FEAT_SMALL_TABLES, the pixel key recomputed): 72 VGPRs with 10 spilled beat 80 with 8 at six by 3.4 % (111.7 against 115.5 ms at 1024 spp)
#endif
#ifndef KY_WAVES_PER_EU_HOT
#define KY_WAVES_PER_EU_HOT 8          // the iterative integrator's both_mis with a single-light fact (every Cornell configuration): 64 VGPRs with 0-4 spilled, +1 % over
#endif                                 // seven (+1.0 ... +2.9 % per Cornell light variant); the other strategies lose up to 15 % at eight
#ifndef KY_WAVES_PER_EU_ENV
#define KY_WAVES_PER_EU_ENV 7          // ... with one environment light (round 6): 72 VGPRs and 15 spilled SGPRs against 64 / 31 at eight, +1.8 % (its both_mis estimate scans every surface with two rays at once: trace_any_pair)
#endif
#ifndef KY_WAVES_PER_EU_NO_FACTS
#define KY_WAVES_PER_EU_NO_FACTS 7     // both_mis without scene facts (any lights, inline shadow rays; every integrator).  Round 3: six (19-39 spilled VGPRs at seven); round 4, with
#endif                                 // the path state the loop no longer carries: seven is +1 ... +2.4 % with two or more lights (the Cornell box with lamp and point light 17.41 -> 17.00 ms), -2 % with one
#ifndef KY_WAVES_PER_EU_NO_FACTS_GENERAL
#define KY_WAVES_PER_EU_NO_FACTS_GENERAL 6   // ... with general shapes (triangle / disk tests inline): -4 ... +2 % at seven, left at six; and path_tracing_recursion_t, whose
#endif                                       // recursion keeps a frame per level: 10.8 -> 15.0 ms at seven on the Cornell box with lamp and point light (the other integrators +1.5 ... +3 %)
#ifndef KY_WAVES_PER_EU_GENERIC
#define KY_WAVES_PER_EU_GENERIC 5   // strategy / integrator read at run time: more code alive at once, 96 VGPRs measured best
#endif

constexpr int KY_FEAT_CORNELL = KY_FEAT_SINGLE_AREA | KY_FEAT_RECT_LIGHTS | KY_FEAT_CARRIERS | KY_FEAT_OWN_CARRIER;   // what the Cornell-lamp instantiations assume (263; with the small tables 391)
constexpr int KY_FEAT_VEACH = KY_FEAT_SPHERE_LIGHTS | KY_FEAT_CARRIERS | KY_FEAT_NO_DELTA | KY_FEAT_SMALL_TABLES;   // what the sphere-lights instantiations assume (create_mis_scene)

struct ItemSlot {  // one fetched work item, decoded once (wave-uniform) and read per lane
    int x0, y0, pix0, s_begin, s_end;
};

// STRATEGY >= 0 fixes direct_sample_enum AND the integrator at compile time (prunes the other estimators and integrators); -1 reads both
// from rc.
// QUEUE (with STRATEGY = both_mis): the light-sampling halves' shadow rays are deferred to the wave's stack `queue_mem`
// (ky_device.hpp, "deferred shadow rays") and traced 64 at a time.
// GENERAL: the scene may hold quads that are not parallelograms, triangles or disks (SceneRef::general); no shipped scene does.
// FEAT: KY_FEAT_* facts the instantiation assumes about the scene (SceneRef::feat): one rectangle area light (every Cornell-box
// configuration of BASELINE.json), one point / directional light, one environment light (the other Cornell variants of ky's drivers).
// INTEGRATOR (with STRATEGY >= 0): path_tracing_iteration_t, or direct_lighting_t / one of the three recursive integrators
// (render_multiple_integrator, ky.cpp:4740-4777, runs all five side by side).
// The host's table of instantiations is g_variants below; kyhip_render_tiles_device launches the first one whose assumptions hold.
// LARGE: the scene's per-lane tables live in dynamic shared memory sized by the scene (more than KY_LDS_SURFACES surfaces or
// KY_LDS_MATERIALS materials; ky_device.hpp, LdsScene).
// resident wavefronts per SIMD an instantiation is compiled for (its register budget): measured per family, see the KY_WAVES_PER_EU_* notes above
template <bool DEBUG_SAMPLER, int STRATEGY, bool QUEUE, bool GENERAL, int FEAT, int INTEGRATOR, bool LARGE>
constexpr int ky_waves_per_eu() {
    return STRATEGY >= 0 ? (QUEUE ? (FEAT ? KY_WAVES_PER_EU_QUEUE_FEAT : KY_WAVES_PER_EU_QUEUE)
                                  : ((FEAT == 0 && STRATEGY == KY_DIRECT_BOTH_MIS) ? ((GENERAL || INTEGRATOR == KY_INTEGRATOR_PATH_TRACING_RECURSION) ? KY_WAVES_PER_EU_NO_FACTS_GENERAL : KY_WAVES_PER_EU_NO_FACTS)
                                     : ((FEAT != 0 && STRATEGY == KY_DIRECT_BOTH_MIS && INTEGRATOR == KY_INTEGRATOR_PATH_TRACING_ITERATION) ? ((FEAT & KY_FEAT_SINGLE_ENV) ? KY_WAVES_PER_EU_ENV : KY_WAVES_PER_EU_HOT) : KY_WAVES_PER_EU)))
                         : KY_WAVES_PER_EU_GENERIC;
}

// The kernel's body is a device function so that two kinds of __global__ entry can wrap it: the template render_kernel below (the library's table
// of instantiations) and the extern "C" kernel of a run-time instantiation (ky_jit.cpp).
template <bool DEBUG_SAMPLER, int STRATEGY, bool QUEUE = false, bool GENERAL = false, int FEAT = 0, int INTEGRATOR = KY_INTEGRATOR_PATH_TRACING_ITERATION, bool LARGE = false>
KY_DEV void render_kernel_body(const DScene* __restrict__ S_, RenderConst rc, ShardConst sh, unsigned* __restrict__ counter, unsigned long long* __restrict__ accum,
                               unsigned* __restrict__ flags, float4* __restrict__ queue_mem) {
    static_assert(!QUEUE || ((STRATEGY == KY_DIRECT_BOTH_MIS || STRATEGY == KY_DIRECT_LIGHT_MIS || STRATEGY == KY_DIRECT_LIGHT) && INTEGRATOR == KY_INTEGRATOR_PATH_TRACING_ITERATION),
                  "the deferred shadow rays belong to the iterative integrator's strategies with a light-sampling half");
    static_assert(FEAT == 0 || (STRATEGY >= 0 && !GENERAL && !DEBUG_SAMPLER), "scene facts are instantiated for kernels with a fixed strategy only");
    static_assert(STRATEGY >= 0 || INTEGRATOR == KY_INTEGRATOR_PATH_TRACING_ITERATION, "the run-time-dispatched kernel reads the integrator from rc");
    const SceneRef S{S_, GENERAL, FEAT, LARGE};
    __shared__ ItemSlot ring[4][KY_RING];
    // the lane's pixel chunk (touched when a path starts or ends, not while a vertex is shaded) lives in LDS, not in registers
    __shared__ int c_xy[256], c_pix[256];
    __shared__ unsigned c_se[256];   // next sample << 7 | samples left in the chunk
    __shared__ uint32_t c_key[QUEUE ? 1 : 256];   // the pixel's sampler key; the deferred-rays kernels recompute it per sample instead (1 KB of their LDS block)
    __shared__ unsigned long long c_def[QUEUE ? 3 * 256 : 1];   // QUEUE: fixed-point sums of the lane's resolved shadow rays
    const int tid = threadIdx.x;
    const LdsScene Lds = stage_scene<LARGE, (FEAT & KY_FEAT_SMALL_TABLES) != 0>(S);
    if (STRATEGY >= 0) { rc.strategy = STRATEGY; rc.integrator = INTEGRATOR; }  // compile-time constants from here on
    const int nee_weight = (rc.strategy == KY_DIRECT_IDLE || rc.integrator < KY_INTEGRATOR_DIRECT_LIGHTING ||
                            rc.integrator == KY_INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION) ? 0 : S->n_lights;

    const int lane = threadIdx.x & 63;
    ItemSlot* my_ring = ring[threadIdx.x >> 6];

    // wave-uniform: the wave's pool of work is the sequence of (item, pixel) pairs of the items it has fetched;
    // `cursor` counts the pairs handed out so far (pair n = pixel n % 64 of fetched item n / 64)
    int fetched = 0;
    int cursor = 0;
    bool exhausted = false;   // the global counter ran past n_items
    // per lane: the pixel chunk being worked on
    c_pix[tid] = -1;
    ShadowQueue sq{nullptr, 0, c_pix, c_def, accum, flags};
    if (QUEUE) {
        c_def[tid] = 0; c_def[256 + tid] = 0; c_def[512 + tid] = 0;
        sq.base = queue_mem + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * (KY_SQ_ENTRY * KY_SQ_CAP);
    }
    bool open = false;        // the chunk has samples left to start (s < s_end)
    bool has_item = false, done = false, alive = false;
    int waited = 0;           // turns the longest-waiting lane has waited for a refill (wave-uniform)
    PathState ps;
    ps.Lo = mk3(0, 0, 0);

    KY_CLK(-1);
    for (;;) {
        KY_CLK(9);   // continuation sampling, roulette, loop overhead
        // ---- (1) lanes whose pixel chunk is finished flush it and take the next (item, pixel) pair of the wave's pool.
        // A lane is NOT tied to one pixel position: whichever lane is free takes the next pixel, so lanes never wait
        // for each other and the wave drains within one chunk of the end of the queue.
        const bool need = !alive && !done && !open;
        const unsigned long long need_mask = __ballot(need);
        // The pass below costs every lane of the wavefront ~120 VALU instructions whoever needs it, and with 24-sample chunks some lane needs it in 62 % of the
        // loop's turns.  So it runs when KY_REFILL_BATCH lanes wait for it, or when the first of them has waited KY_REFILL_WAIT turns, or when no lane has
        // anything else to do (round 4: 6 / 4 is worth +1.3 % on configs[1], +0.7 % on configs[2]; 2 / 1 costs 1 %, 16 / 12 1.3 %: profiles/r04_c_ab_loop_state.txt).
        // Which lane renders which pixel chunk changes with it; the image does not (a chunk's sum does not depend on its lane).
        bool refill = need_mask != 0;
        if (refill && __popcll(need_mask) < KY_REFILL_BATCH && waited < KY_REFILL_WAIT && __any(alive || (open && !done))) { refill = false; ++waited; }
        if (refill) {  // wave-uniform branch: every lane runs the bookkeeping below
            waited = 0;
            if (need && has_item) {
                const float v[3] = {ps.Lo.x * rc.inv_spp, ps.Lo.y * rc.inv_spp, ps.Lo.z * rc.inv_spp};   // 3717, once per chunk
                ps.Lo = mk3(0, 0, 0);
                const int pix = c_pix[tid];
                c_pix[tid] = -1;   // rays of this chunk that are still on the stack go to the global accumulator directly
                // Nearly every chunk sum is three finite, non-negative numbers far below the accumulator's range: when that holds for every flushing lane of the wavefront
                // (one min3, two adds, two compares; a NaN fails the sum's test) the conversion needs no classification, no sign and no flag word -- 30 instead of 75
                // VALU instructions for every lane of the wavefront, flushing or not.  Any other wavefront takes the complete form below.
                const bool plain = fminf(fminf(v[0], v[1]), v[2]) >= 0.f && (v[0] + v[1] + v[2]) < 2.0e9f;
                if (__all(plain)) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float hi = __builtin_floorf(v[ch]);
                        const float lo = __builtin_rintf((v[ch] - hi) * 4294967296.0f);   // fixed_from_float for a >= 0
                        unsigned long long fx = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
                        if (QUEUE) { fx += c_def[ch * 256 + tid]; c_def[ch * 256 + tid] = 0; }
                        atomicAdd(&accum[(size_t)pix * 3 + ch], fx);
                    }
                } else
                {
                    unsigned fl = 0;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        unsigned long long fx = film_fixed(v[ch], ch, fl);   // NaN / +-inf: flag bits, nothing added
                        if (QUEUE) { fx += c_def[ch * 256 + tid]; c_def[ch * 256 + tid] = 0; }
                        if (fx != 0) atomicAdd(&accum[(size_t)pix * 3 + ch], fx);
                    }
                    if (fl) atomicOr(&flags[pix], fl);
                }
                has_item = false;
            }
            const int n_need = __popcll(need_mask);
            // fetch until the pool covers every requesting lane (at most two items: n_need <= 64), or the queue is empty.
            // Slot reuse: item fetched - KY_RING was handed out completely long ago (cursor >= (fetched - 2) * 64).
            while (!exhausted && cursor + n_need > fetched * 64) {
                // a wave's first item is its own index: the launch does not begin with every wavefront of the chip queueing at one
                // counter (one word serves ~90 dequeues per microsecond); later items come from the counter, offset by the wave count
                unsigned id = blockIdx.x * 4u + (threadIdx.x >> 6);
                if (fetched > 0 && lane == 0) id = atomicAdd(counter, 1u) + gridDim.x * 4u;
                id = __builtin_amdgcn_readfirstlane(id);
                if (id >= sh.n_items) { exhausted = true; break; }
                const int c = (int)(id / (unsigned)sh.n_blocks), b = (int)(id % (unsigned)sh.n_blocks);   // chunk-major
                const int k = b / sh.blocks_per_tile, inner = b % sh.blocks_per_tile;
                const int bx = inner % sh.blocks_w, by = inner / sh.blocks_w;
                const int tile = sh.tile_first + k * sh.tile_step;
                if (lane == 0) {
                    ItemSlot it;
                    const int trow = tile / sh.tiles_x, tcol = (tile % sh.tiles_x + trow) % sh.tiles_x;   // rotated rows
                    it.x0 = tcol * sh.tile_w + bx * 8;
                    it.y0 = trow * sh.tile_h + by * 8;
                    it.pix0 = (k * sh.tile_h + by * 8) * sh.tile_w + bx * 8;
                    chunk_range(chunk_plan(rc.spp), c, it.s_begin, it.s_end);
#ifdef KY_EXP_TRANSPOSE   // experiment (1024 spp, 64 chunks of 16): item c of a block = its pixel c, the item's 64 units = that pixel's 64 chunks
                    it.x0 += c & 7; it.y0 += c >> 3; it.pix0 += (c >> 3) * sh.tile_w + (c & 7);
#endif
                    my_ring[fetched % KY_RING] = it;
                }
                ++fetched;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // lane 0's slot writes before the other lanes' reads
            if (need) {
                // number of requesting lanes below this one: v_mbcnt, no lane mask kept in registers
                const int mine = cursor + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need_mask, 0u));
                if (mine < fetched * 64) {
                    const ItemSlot it = my_ring[(mine >> 6) % KY_RING];
#ifdef KY_EXP_TRANSPOSE
                    const int px = 0, py = 0;
#else
                    const int px = mine & 7, py = (mine >> 3) & 7;
#endif
                    const int x = it.x0 + px, y = it.y0 + py;
                    const bool in_range = x < rc.width && y < rc.height;
                    c_xy[tid] = x | (y << 16);
                    c_pix[tid] = it.pix0 + py * sh.tile_w + px;
                    if (!QUEUE) c_key[tid] = sampler_pixel_key(rc.seed, (uint32_t)(y * rc.width + x));
#ifdef KY_EXP_TRANSPOSE
                    c_se[tid] = ((unsigned)((mine & 63) * 16) << 7) | 16u;
                    open = in_range;
#else
                    c_se[tid] = ((unsigned)it.s_begin << 7) | (unsigned)(it.s_end - it.s_begin);
                    open = in_range && it.s_begin < it.s_end;
#endif
                    has_item = in_range;
                } else {
                    done = true;  // only reachable once the queue is exhausted
                }
            }
            cursor = min(cursor + n_need, fetched * 64);
        }
        KY_CLK(0);
        // ---- (2) regenerate + trace until enough lanes hold a vertex ----
        // A lane whose path ends at the traversal itself (a miss, the depth cap) would sit out the whole shading phase,
        // which costs 2 traversals per light.  When many lanes are in that state, they regenerate and trace once more
        // before the wave moves on (wave-uniform decision), so the expensive phase runs with fuller lanes.
        Vertex v;
        v.in_lds = true;   // shading frame and local wo in LDS (ky_device.hpp, VertexLds)
        bool have_vertex = false;
        for (int attempt = 0;; ++attempt) {
            if (!alive && !done && open) {  // next camera sample of this lane's pixel, 3712-3715
                const int xy = c_xy[tid];
                const unsigned se = c_se[tid];
                const uint32_t key = QUEUE ? sampler_pixel_key(rc.seed, (uint32_t)((xy >> 16) * rc.width + (xy & 0xffff))) : c_key[tid];
                path_begin<DEBUG_SAMPLER, true>(ps, S, key, xy & 0xffff, xy >> 16, (int)(se >> 7));
                c_se[tid] = se + 127;             // next sample + 1, samples left - 1
                open = (se & 127) > 1;
                alive = true;
            }
            const bool tracing = alive && !have_vertex;
            if (!__any(tracing)) break;
            if (tracing) {
                const bool ended = !path_intersect<DEBUG_SAMPLER>(ps, v, S, Lds, rc);
                if (!ended) have_vertex = true;
                if (ended) {
                    alive = false;
                }
            }
            // (single-light instantiations never retrace: one more traversal against two per vertex never paid there, and the loop
            // around it cost 1.9 % by itself; with Veach's five lights the retrace is worth 16 %)
            if (attempt >= ((FEAT & KY_FEAT_SINGLE_LIGHT) ? 0 : KY_MAX_RETRACE)) break;
            // lanes that could start another path right now; worth one more traversal if they would otherwise idle
            // through (2 traversals x lights + shading) that is worth more than the extra traversal
            const int idle = __popcll(__ballot(!alive && !done && open));
            if (idle * (2 * nee_weight + 1) < KY_RETRACE_THRESHOLD) break;
        }
        KY_CLK(1);
        if (!__any(alive)) {
            if (__all(done)) break;
            path_state_dead(ps);   // no lane holds a path: nothing of the path state is carried into the next turn (but Lo, the chunks' sums)
            continue;  // lanes are between items: (1) serves them on the next turn
        }
        unsigned tag = 0;
        if (QUEUE) tag = ((unsigned)c_pix[tid] << 6) | (unsigned)lane;
        // ---- (3) shade the vertex: direct lighting, continuation ----
        {
            const bool cont = path_shade<DEBUG_SAMPLER>(ps, v, S, Lds, rc, have_vertex, -1, nullptr, QUEUE ? &sq : nullptr, tag,
                                                        STRATEGY >= 0 && INTEGRATOR == KY_INTEGRATOR_PATH_TRACING_RECURSION, true);  // wave-uniform call
            if (have_vertex && !cont) {
                alive = false;
            }
        }
    }
    if (QUEUE) {  // what is left on the stack: the lanes have flushed, so these go to the global accumulators
        sq_drain(S, sq);
    }
    KY_CLK(-2);
}

template <bool DEBUG_SAMPLER, int STRATEGY, bool QUEUE = false, bool GENERAL = false, int FEAT = 0, int INTEGRATOR = KY_INTEGRATOR_PATH_TRACING_ITERATION, bool LARGE = false>
__global__ __launch_bounds__(256, (ky_waves_per_eu<DEBUG_SAMPLER, STRATEGY, QUEUE, GENERAL, FEAT, INTEGRATOR, LARGE>())) void render_kernel(
    const DScene* __restrict__ S_, RenderConst rc, ShardConst sh, unsigned* __restrict__ counter, unsigned long long* __restrict__ accum, unsigned* __restrict__ flags,
    float4* __restrict__ queue_mem) {
    render_kernel_body<DEBUG_SAMPLER, STRATEGY, QUEUE, GENERAL, FEAT, INTEGRATOR, LARGE>(S_, rc, sh, counter, accum, flags, queue_mem);
}
