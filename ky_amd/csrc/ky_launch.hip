/*
 * ky_launch.hip -- the render kernels' table and launch path of libkyhip.so (see include/kyhip.h): kyhip_render_tiles_device and what it needs
 * (device contexts, per-stream launch state, the scene cache), the film kernels, the fp64 smallpt kernels.  The other translation units: ky_pack.cpp
 * (host: scene packing, occluder proof, policies), ky_jit.cpp (run-time instantiations' code cache), ky_seam.cpp (host-film calls), ky_kat.hip (KAT entries).
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
#include <cstdio>
#include <cstring>

#include "ky_ctx.hpp"
#include "ky_render.hpp"
#include "ky_smallpt.hpp"

using namespace kyh;
static_assert(kyh::KY_SP_MAX_SPHERES == kysp::SP_MAX_SPHERES, "smallpt_check (ky_pack.cpp) and the kernels' LDS table");

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
#include "ky_queue.hpp"   // the queue engine: render_kernel_q

// fixed-point accumulator -> clamp01(L) (3726) -> fp32 tile buffer, one thread per pixel.  The kernel leaves the accumulators, the flag words and the work counter
// ZERO behind it: the next frame on this stream starts from them without the two fills (round 6: 2 x ~5 us of a launch's fixed cost, a twentieth of what a 1/8
// shard of configs[1] spends outside its render kernel).
__global__ void resolve_kernel(unsigned long long* __restrict__ accum, unsigned* __restrict__ flags, float* __restrict__ tiles, int n_pix, unsigned* __restrict__ counter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *counter = 0u;
    if (i >= n_pix) return;
    const unsigned fl = flags[i];
    flags[i] = 0u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float v = (float)((double)(long long)accum[3 * (size_t)i + ch] * (1.0 / KY_FIX_SCALE));
        accum[3 * (size_t)i + ch] = 0ull;
        const bool nan = (fl >> ch) & 1u, pinf = (fl >> (3 + ch)) & 1u, ninf = (fl >> (6 + ch)) & 1u;
        if (pinf) v = 1.f;
        if (ninf) v = 0.f;
        if (nan || (pinf && ninf)) v = 0.f;  // a NaN pixel: clamp01 keeps NaN in the reference and its 8-bit image shows 0
        tiles[3 * (size_t)i + ch] = fminf(fmaxf(v, 0.f), 1.f);
    }
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

// ------------------------------------------------------------------------------------------------
// device contexts, launch state per stream, the scene cache (ky_ctx.hpp)
// ------------------------------------------------------------------------------------------------
namespace kyh {
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
int get_ctx(int device, DeviceCtx** out) {
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
DeviceCtx* find_ctx(int device) {
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    return (device >= 0 && device < (int)g_ctx.size()) ? g_ctx[device].get() : nullptr;
}

// The launch state of `stream` on this device (created on first use; with more than KY_STREAM_STATES streams in use the least
// recently used state is handed over, after the device has drained).
int get_stream_state(DeviceCtx* c, hipStream_t stream, StreamState** out) {
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

// The device copy of `scene`, from the cache or uploaded on `stream`; launches on `stream` may read it when this returns.
int upload_scene(DeviceCtx* c, const ky_scene* scene, hipStream_t stream, SceneSlot** out) {
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
}  // namespace kyh

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

int kyhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
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
#ifdef KY_FEW_VARIANTS   // measurement builds (tools/mkvariant.sh -DKY_FEW_VARIANTS): the headline kernels and one catch-all, compiled in a sixth of the time
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL | KY_FEAT_SMALL_TABLES | KY_FEAT_BOXES | KY_FEAT_AXIS_ALIGNED | KY_FEAT_FLAT_PHONG, IT),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL | KY_FEAT_SMALL_TABLES | KY_FEAT_BOXES, IT),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL | KY_FEAT_SMALL_TABLES, IT),
#ifndef KY_NO_XPLANK_ROW   // (measurement builds: the sphere-lights kernel without the planks' fact)
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, true, false, KY_FEAT_VEACH | KY_FEAT_FLAT_PHONG | KY_FEAT_X_PLANKS, IT),
#endif
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, true, false, KY_FEAT_VEACH | KY_FEAT_FLAT_PHONG, IT),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV | KY_FEAT_SMALL_TABLES | KY_FEAT_BOXES | KY_FEAT_AXIS_ALIGNED | KY_FEAT_FLAT_PHONG, IT),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV | KY_FEAT_SMALL_TABLES | KY_FEAT_AXIS_ALIGNED | KY_FEAT_FLAT_PHONG, IT),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV, IT),
    KY_VARIANT(false, -1, false, false, 0, IT),
#else
    // the iterative integrator, both_mis: by scene facts
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL | KY_FEAT_SMALL_TABLES | KY_FEAT_BOXES | KY_FEAT_AXIS_ALIGNED | KY_FEAT_FLAT_PHONG, IT),   // ... and nothing planar but axis rectangles: configs[1], [4]
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL | KY_FEAT_SMALL_TABLES | KY_FEAT_BOXES, IT),   // ... whose walls / lamp housing are faces of boxes
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL | KY_FEAT_SMALL_TABLES | KY_FEAT_AXIS_ALIGNED, IT),   // ... axis rectangles only, no boxes (kyhip_set_boxes(0): the
                                                                                                           // kernel the tests hold against the fact-free one to 2.4e-7)
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL | KY_FEAT_SMALL_TABLES, IT),   // one rectangle area light, at most 16 surfaces and 8 materials
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_CORNELL, IT),                  // one rectangle area light
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_DELTA | KY_FEAT_BOXES | KY_FEAT_AXIS_ALIGNED | KY_FEAT_FLAT_PHONG, IT),   // one point / directional light in a room that is a box (configs[3]'s frames)
    // (round 5 measured KY_FEAT_SINGLE_ENV | KY_FEAT_BOXES slower -- its BSDF-sampled rays were a second inlined nearest-hit traversal; since round 6 both of the
    // estimate's rays are any-hit queries that share one scan, estimate_env_both, and the box rows below pay: ky's default frame 38.3 -> 32.6 ms at 512 spp)
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_DELTA, IT),             // one point / directional light
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV | KY_FEAT_SMALL_TABLES | KY_FEAT_BOXES | KY_FEAT_AXIS_ALIGNED | KY_FEAT_FLAT_PHONG, IT),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV | KY_FEAT_SMALL_TABLES | KY_FEAT_AXIS_ALIGNED | KY_FEAT_FLAT_PHONG, IT),
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, false, false, KY_FEAT_SINGLE_ENV, IT),               // one environment light
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, true, false, KY_FEAT_VEACH | KY_FEAT_FLAT_PHONG | KY_FEAT_X_PLANKS, IT),   // ... whose tilted rectangles are planks about the x axis: configs[2]
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, true, false, KY_FEAT_VEACH | KY_FEAT_FLAT_PHONG, IT),   // several sphere lights, no mirror / glass, plastic on rectangles only: create_mis_scene's materials
    KY_VARIANT(false, KY_DIRECT_BOTH_MIS, true, false, KY_FEAT_VEACH, IT),                     // several sphere lights, no mirror / glass
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
#endif
};
constexpr int KY_N_VARIANTS = (int)(sizeof g_variants / sizeof g_variants[0]);
static_assert(KY_N_VARIANTS <= 64, "DeviceCtx::variant_blocks");

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

}  // extern "C"
int kyh::render_replay_feat(const ky_scene* scene, const ky_render_params* p, const DScene* packed) {   // (ky_ctx.hpp)
    const Variant* v = pick_variant(p, packed, shadow_queue_wanted(scene), p->width * p->height);
    return v != nullptr ? (v->feat & (KY_FEAT_BOXES | KY_FEAT_SINGLE_ENV)) : 0;
}
bool kyh::render_uses_boxes(const ky_scene* scene, const ky_render_params* p, const DScene* packed) { return (render_replay_feat(scene, p, packed) & KY_FEAT_BOXES) != 0; }
extern "C" {

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
            st->ws_clean = false;
        }
        ws = st->ws;
    }
    unsigned long long* accum = (unsigned long long*)ws;
    unsigned* flags = (unsigned*)(accum + (size_t)sh.n_pix * 3);
    const bool large_scene = scene->surface_count > KY_LDS_SURFACES || scene->material_count > KY_LDS_MATERIALS;
    const size_t lds_bytes = large_scene ? (size_t)lds_scene_bytes(scene->surface_count, scene->material_count, scene->light_count) : 0;   // LARGE kernels' LdsScene
    // accumulators, flags and the work counter start at zero: resolve_kernel leaves the library's own block that way (a caller's workspace, a new block and the frame
    // after a failed launch are filled here)
    const bool own_ws = ws == st->ws;
    if (!(own_ws && st->ws_clean)) {
        HIP_TRY(hipMemsetAsync(ws, 0, own_ws ? st->ws_bytes : need, stream));
        HIP_TRY(hipMemsetAsync(st->d_counter, 0, sizeof(unsigned), stream));
    }
    st->ws_clean = false;   // until this frame's resolve_kernel is enqueued

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
        const int jit_mode = kyjit::mode();
        c->last_note.clear();
        if (v->strategy < 0 && jit_mode == 0) c->last_note = " [the run-time-dispatched kernel: no row of the table holds this launch's strategy / integrator / shapes; kyhip_set_jit(1 / 2) instantiates its own]";
        // (a mode nobody chose -- the default of round 6 -- instantiates only where the table's pick knows nothing of the scene: a launch served by a row of facts stays on it,
        // a scene outside the table of facts gets its own kernel; kyhip_set_jit(1 / 2) / KYHIP_JIT instantiate for every launch that is not exactly a row)
        const bool jit_wanted = jit_mode != 0 && (!kyjit::mode_by_default() || v->feat == 0 || v->strategy < 0);
        if (jit_wanted && specialisation_enabled() && p->integrator >= KY_INTEGRATOR_DIRECT_LIGHTING) {   // (kyhip_set_specialisation(0) asks for the fact-free kernels: nothing to instantiate)
            const bool dbg = p->sampler == KY_SAMPLER_DEBUG, general = sc->h->general != 0;
            int feat = (dbg || general) ? 0 : sc->h->feat;
            // the box traversal pays where it was measured to (one lamp, one point / directional light: +3-4 %; one environment light under both_mis, whose estimate's two
            // rays share one any-hit scan: estimate_env_both); in instantiations that inline the nearest-hit traversal more than once (an environment light's BSDF-sampled
            // rays under the other strategies, several lights) it measured 7-10 % SLOWER: those keep the rectangle scan
            const bool env_pair = (feat & KY_FEAT_SINGLE_ENV) && p->direct_sample == KY_DIRECT_BOTH_MIS && p->integrator != KY_INTEGRATOR_PATH_TRACING_RECURSION;
            if (!((feat & (KY_FEAT_SINGLE_AREA | KY_FEAT_SINGLE_DELTA)) || env_pair)) feat &= ~KY_FEAT_BOXES;
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
                    // mode 1: blocks for the compile the first time (a few seconds), then memory / disk.  mode 2: a missing object is compiled by a background
                    // thread; this launch (and every one until the object is there) takes the table's kernel -- `pending` -- and asks again next time.
                    bool pending = false;
                    const kyjit::Code* code = kyjit::get_code(expr, jit_mode == 1, &pending);
                    if (code) {
                        if (!(hipModuleLoadData(&k.module, code->object.data()) == hipSuccess && hipModuleGetFunction(&k.fn, k.module, kyjit::k_entry) == hipSuccess)) {
                            (void)hipGetLastError();
                            k.fn = nullptr;
                            k.failed = true;
                        }
                    } else if (!pending) {
                        k.failed = true;   // the table's kernel serves this launch and every later one of its kind (kyhip_jit_status() says why)
                    }
                    if (pending) c->last_note = " [its own instantiation render_kernel<" + std::string(expr) + "> is being compiled (kyhip_set_jit mode 2): this frame ran on the table's kernel]";
                }
                if (k.failed) c->last_note = " [its own instantiation is unavailable: " + kyjit::status() + "]";
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
    hipLaunchKernelGGL(resolve_kernel, dim3((sh.n_pix + 255) / 256), dim3(256), 0, stream, accum, flags, d_tiles, sh.n_pix, st->d_counter);
    HIP_TRY(hipGetLastError());
    st->ws_clean = own_ws;
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
    name += c->last_note;
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

// ---- smallpt's scene in double precision (ky_smallpt.hpp); the scene tables and the argument check are host code: ky_pack.cpp ----
int kyhip_smallpt_render(int device, const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p, double* image_rgb) {
    int rcode = kyh::smallpt_check(spheres, n, p);
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
    int rcode = kyh::smallpt_check(spheres, n_spheres, p);
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
