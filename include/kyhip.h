/*
 * kyhip.h -- C ABI of the MI355X (gfx950) path-tracing integrator that replaces the body of
 *            ky's `integrator_t::render()` for `path_tracing_iteration_t`.
 *
 * The reference (infancy/ky, one C++ translation unit) has no FFI layer; its seam for this hot
 * path is the C++ call
 *
 *     void integrator_t::render(scene_t* scene, sampler_t* sampler, film_t* film)   ky.cpp:3689
 *     virtual color_t integrator_t::Li(ray_t, scene_t*, sampler_t*)                 ky.cpp:3792
 *     std::unique_ptr<integrator_t> create_integrator(enum, depth, direct_sample)   ky.cpp:4621
 *
 * This header is what a binding for that seam would declare: plain structs, pointers and sizes,
 * no C++ / torch types.  Every entry point cites the reference interface it stands in for.
 * The library (libkyhip.so) owns all device memory it allocates; caller-owned buffers are only
 * read or accumulated into.  No entry point throws; all return 0 on success or a negative
 * ky_status, and kyhip_last_error() returns a thread-local message for the last failure.
 *
 * All arithmetic on the path is fp32 (ky.cpp:171-172).
 */
#ifndef KYHIP_H
#define KYHIP_H

#ifndef __HIPCC_RTC__   /* the library compiles this header at run time too (hiprtc: no system headers, the types are built in) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define KYHIP_ABI_VERSION 4   /* 4 (round 5): + kyhip_jit_failures, kyhip_multi_status, kyhip_seam_threads, kyhip_film_alloc / _free; kyhip_set_jit mode 2 */

/* ------------------------------------------------------------------------------------------
 * Scene description: a flat restatement of what scene_t holds (ky.cpp:3535-3546) and what the
 * path reads from it (SURVEY.md 8(b) "Inputs read by the path").
 * ---------------------------------------------------------------------------------------- */

/* shape_t subclasses, ky.cpp:1100 (disk), 1165 (triangle), 1245 (rectangle), 1326 (sphere) */
typedef enum ky_shape_kind {
    KY_SHAPE_DISK      = 0,
    KY_SHAPE_TRIANGLE  = 1,
    KY_SHAPE_RECTANGLE = 2,
    KY_SHAPE_SPHERE    = 3
} ky_shape_kind;

typedef struct ky_shape {
    int32_t kind;      /* ky_shape_kind */
    float   p[4][3];   /* rectangle: p0..p3 (1318-1321); triangle: p0..p2 (1238-1240);
                          disk: p[0] = position_ (1159); sphere: p[0] = center_ (1516) */
    float   normal[3]; /* stored normal_ after flip_normal (1174-1176, 1256-1258), disk: normalize(normal) (1105); MUST be unit
                          length (the reference's constructors normalise it); KY_ERR_INVALID_VALUE otherwise */
    float   radius;    /* disk radius_ (1161), sphere radius_ (1517) */
} ky_shape;

/* material_t subclasses, ky.cpp:2579 (matte), 2596 (mirror), 2613 (glass), 2639 (plastic) */
typedef enum ky_material_kind {
    KY_MATERIAL_MATTE   = 0,
    KY_MATERIAL_MIRROR  = 1,
    KY_MATERIAL_GLASS   = 2,
    KY_MATERIAL_PLASTIC = 3
} ky_material_kind;

typedef struct ky_material {
    int32_t kind;                 /* ky_material_kind */
    float   color0[3];            /* matte diffuse_color_ / mirror specular_color_ / glass reflection_color_ / plastic diffuse_color_ */
    float   color1[3];            /* glass transmission_color_ / plastic specular_color_ */
    float   eta;                  /* glass eta_ (2634); the outside index is 1 (2630) */
    float   exponent;             /* plastic exponent_ (2677) */
    float   diffuse_probability;  /* plastic diffuse_probility_  = lum(Kd)/(lum(Kd)+lum(Ks))   (2653-2657) */
    float   specular_probability; /* plastic specular_probility_ = lum(Ks)/(lum(Kd)+lum(Ks))   (2658) */
} ky_material;

/* light_t subclasses, ky.cpp:2810 (point), 2868 (direction), 2923 (area), 2999 (environment) */
typedef enum ky_light_kind {
    KY_LIGHT_POINT       = 0,
    KY_LIGHT_DIRECTION   = 1,
    KY_LIGHT_AREA        = 2,
    KY_LIGHT_ENVIRONMENT = 3
} ky_light_kind;

typedef struct ky_light {
    int32_t kind;          /* ky_light_kind */
    int32_t shape;         /* area: index into ky_scene.shapes of the shape it SAMPLES (area_light_t::shape_, 2993).
                              May differ from the shape of the surface that carries the light (Veach lights 1/2, 3498-3499 vs 3525-3526) */
    float   color[3];      /* point intensity_ / direction irradiance_ / area radiance_ / environment radiance_ */
    float   position[3];   /* point: world_position_ (2802) */
    float   direction[3];  /* direction: normalized world_direction_ (2874) */
    float   world_radius;  /* direction / environment: world_radius_ set by preprocess (3555-3574) */
} ky_light;

/* surface_t, ky.cpp:3071-3075 */
typedef struct ky_surface {
    int32_t shape;       /* index into shapes */
    int32_t material;    /* index into materials */
    int32_t area_light;  /* index into lights, or -1 (nullptr) */
} ky_surface;

/* camera_t members after its constructor ran, ky.cpp:1900-1905 */
typedef struct ky_camera {
    float position[3];
    float front[3];       /* normalized */
    float right[3];       /* scaled by tan(fov/2) * aspect (1878) */
    float up[3];          /* scaled by tan(fov/2)          (1879) */
    float resolution[2];
} ky_camera;

typedef struct ky_scene {
    const ky_shape*    shapes;    int32_t shape_count;
    const ky_material* materials; int32_t material_count;
    const ky_light*    lights;    int32_t light_count;     /* light_list_, in order (3541) */
    const ky_surface*  surfaces;  int32_t surface_count;   /* surface_list_, in order: order decides ties (3177) */
    int32_t            environment_light;                  /* index into lights of environment_light_ (3542), or -1 */
    ky_camera          camera;
} ky_scene;

/* Hard limits of the device path (the scene lives in on-chip memory). */
#define KYHIP_MAX_SHAPES    256
#define KYHIP_MAX_SURFACES  256
#define KYHIP_MAX_MATERIALS 64
#define KYHIP_MAX_LIGHTS    16

/* ------------------------------------------------------------------------------------------
 * Render parameters.
 * ---------------------------------------------------------------------------------------- */

/* integrator_enum_t values that have a device path (ky.cpp:3625-3654). */
typedef enum ky_integrator_kind {
    KY_INTEGRATOR_POSITION               = 0,  /* debug_integrator_t, 4112 */
    KY_INTEGRATOR_NORMAL                 = 1,  /* debug_integrator_t, 4114 */
    KY_INTEGRATOR_BASECOLOR              = 2,  /* debug_integrator_t, 4116 */
    KY_INTEGRATOR_DIRECT_LIGHTING        = 6,  /* direct_lighting_t, 4125 */
    KY_INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION   = 8,   /* simple_path_tracing_recursion_t, 4191 (BSDF sampling only) */
    KY_INTEGRATOR_PATH_TRACING_RECURSION          = 9,   /* path_tracing_recursion_t, 4305 */
    KY_INTEGRATOR_PATH_TRACING_RECURSION_DEFERED  = 10,  /* path_tracing_recursion_defered_t, 4409 */
    KY_INTEGRATOR_PATH_TRACING_ITERATION = 11  /* path_tracing_iteration_t, 4523 -- the hot path */
} ky_integrator_kind;

/* direct_sample_enum_t (ky.cpp:3608-3623).  Matched by exact value (3840-3862). */
typedef enum ky_direct_sample {
    KY_DIRECT_IDLE      = 0,
    KY_DIRECT_BSDF      = 4,
    KY_DIRECT_LIGHT     = 8,
    KY_DIRECT_BSDF_MIS  = 16,
    KY_DIRECT_LIGHT_MIS = 32,
    KY_DIRECT_BOTH_MIS  = 48
} ky_direct_sample;

/* sampler_t subclasses that can be instantiated (ky.cpp:922, 949). */
typedef enum ky_sampler_kind {
    KY_SAMPLER_DEBUG  = 0,  /* debug_sampler_t: every number 0.5, camera sample = pixel centre (933-946) */
    KY_SAMPLER_RANDOM = 1   /* random_sampler_t semantics (uniform [0,1) fp32) on a counter-based generator
                               keyed (seed, pixel, sample, dimension); see DESIGN.md "Random numbers" */
} ky_sampler_kind;

typedef struct ky_render_params {
    int32_t  integrator;         /* ky_integrator_kind */
    int32_t  max_path_depth;     /* path_integrator_t::max_path_depth_ (4182) */
    int32_t  direct_sample;      /* ky_direct_sample, path_integrator_t::direct_sample_enum_ (4183) */
    int32_t  samples_per_pixel;  /* sampler_t::samples_per_pixel_ (918) */
    int32_t  sampler;            /* ky_sampler_kind */
    uint32_t seed;               /* rng_t seed, default 1234 (833) */
    int32_t  width, height;      /* film->get_resolution() (3692): the pixel loop bounds */
    /* Image-tile sharding.  The film is cut into tile_w x tile_h tiles; with
       tiles_x = ceil(width / tile_w), tile number t lies in tile row t / tiles_x at tile column
       (t % tiles_x + t / tiles_x) % tiles_x -- row-major with every row rotated by its index, so that
       an interleaved shard is a comb of diagonals even when tiles_x is a multiple of tile_step.
       This call renders tiles tile_first, tile_first+tile_step, ...  A single-GPU render uses
       tile_first = 0, tile_step = 1.  tile_w and tile_h must be multiples of 8. */
    int32_t  tile_w, tile_h;
    int32_t  tile_first, tile_step;
} ky_render_params;

typedef enum ky_status {
    KY_OK                = 0,
    KY_ERR_INVALID_VALUE = -1,  /* bad argument, unknown enum (create_integrator returns nullptr, 4638;
                                   sample_all_light leaves the std::function empty, 3860) */
    KY_ERR_LIMIT         = -2,  /* scene larger than the KYHIP_MAX_* limits, or a frame whose work items / pixels do not fit
                                   the device code's 32-bit indices (e.g. 4096 x 4096 at more than ~1.8e6 spp) */
    KY_ERR_DEVICE        = -3,  /* HIP runtime error (message has the hipError string) */
    KY_ERR_NO_DEVICE     = -4   /* no gfx950 device visible: the product has no CPU fallback */
} ky_status;

const char* kyhip_last_error(void);
/* Selects the render kernel for path_tracing_iteration_t: 0 = lane engine (default: one lane walks one path, state in
   registers), 1 = queue engine (experimental: path state in LDS, one queue per path state, wavefronts take full batches
   from the fullest queue -- ky_amd/csrc/ky_queue.hpp).  Both run the same per-sample arithmetic and random streams; the
   images differ by float summation order only.  The default is read from the environment variable KYHIP_ENGINE
   ("lane" / "queue").  Returns the previous engine.  Not part of the reference's interface: a tuning knob. */
int         kyhip_set_engine(int engine);
/* Specialised kernel instantiations -- a scene lit by exactly one rectangle area light runs a both_mis kernel compiled without the
   other light kinds, the environment term, the lights loop and the other light shapes' sampling; each of the other five
   direct-lighting strategies of the iterative integrator has its own kernel instead of the run-time-dispatched one -- on = 1 (default;
   environment variable KYHIP_SPECIALISE=0 turns them off) / off = 0.  The image does not depend on it (to the last bit of a pixel: the
   compiler contracts a few multiply-adds differently once code around them is gone).  Returns the previous setting.
   A tuning knob, like kyhip_set_engine. */
int         kyhip_set_specialisation(int on);
/* Boxes: axis-aligned rectangles that are whole faces of a common axis-aligned box -- the five walls of a Cornell box, its lamp housing -- are tested by the
   nearest-hit traversal with ONE slab test per box instead of face by face (kernels with the fact KY_FEAT_BOXES; kyhip_scene_boxes says which surfaces).  The slab
   test's hit distance differs from the rectangle test's by up to 15 units in the last place, so unlike the switch above this one moves an image beyond its last bit
   (tests/test_boxes.py: by how much).  on = 1 (default; environment variable KYHIP_BOXES=0 turns it off) / off = 0; needs kyhip_set_specialisation(1).  Returns the
   previous setting.  A tuning knob. */
int         kyhip_set_boxes(int on);
/* Deferred shadow rays (the render kernels' QUEUE instantiations: light samples wait on a per-wavefront stack until 64 of them fill a traversal).
   mode -1 (default): by the scene -- when its sphere area lights outnumber its point / directional lights by five or more (create_mis_scene's five
   lamps; measured crossing, tools/queue_policy.py); 0: never; 1: for every scene with lights.  The environment variable KYHIP_SHADOW_QUEUE = 0 / 1 sets the
   initial mode.  The image does not depend on it beyond float summation order (contributions enter the film in fixed point either way).  Returns the
   previous mode.  A tuning knob, like kyhip_set_engine. */
int         kyhip_set_shadow_queue(int mode);
/* Run-time instantiations.  The render kernel is a template over (sampler, strategy, integrator, deferred shadow rays, general shapes, scene
   facts, table size); the library ships a fixed table of instantiations and picks the nearest one per launch.  mode 1: a launch whose exact
   combination -- with ALL of its scene's facts -- is not in the table gets its own kernel, compiled from the library's embedded source by the
   ROCm compiler (a child process: $KYHIP_HIPCC, default /opt/rocm/bin/hipcc) on first use (a few seconds, blocking that launch; afterwards a
   code object in memory and under $KYHIP_CACHE_DIR, default ~/.cache/kyhip).
   mode 2 (round 5): the same without the wait -- a missing instantiation is compiled by a background thread while the table's kernel renders, and
   launches switch to it once it is there; all shards of one kyhip_render_multi call use one kernel.  WHEN a process switches is a matter of timing,
   and table and own kernel differ in the last bit of a pixel: use mode 1 where frames must be reproducible bit for bit (the ranks of a multi-process
   job: ky_amd/dist.py refuses mode 2 there).
   mode 0: the table only.  The image does not depend on the mode beyond the last bit of a pixel (like kyhip_set_specialisation).
   The mode a process STARTS in (round 6; rounds 4-5: 0): the environment variable KYHIP_JIT = 0 / 1 / 2 if set; otherwise 2 when this process can run instantiations
   without surprising anyone -- a compiler at a known path, no profiler attached, a single-process job (WORLD_SIZE unset or 1) -- and 0 when not; kyhip_jit_status()
   says which and why.  kyhip_last_kernel() names the kernel a launch ran on and adds "[its own instantiation ... is being compiled ...]" / "[... is unavailable: ...]"
   when that was the table's kernel for the time being.  kyhip_set_jit(-1) returns the mode without changing it.  If no compiler is found or a compile fails, the table's kernel runs and kyhip_jit_status() says why
   (kyhip_jit_failures() counts such compiles: a multi-rank job checks it, because a rank that fell back renders its shards on another kernel).
   The compiler is started with posix_spawn (argv array, no shell) and an environment without LD_PRELOAD / profiler variables; in a process that
   is itself being profiled nothing is compiled ("stands down under a profiler").  Processes share the cache directory safely (flock, write-once).
   Returns the previous mode.  A tuning knob, not part of the reference's interface. */
int         kyhip_set_jit(int mode);
const char* kyhip_jit_status(void);
int         kyhip_jit_failures(void);
/* Host only (no GPU needed): compiles -- or fetches from the cache -- the instantiation named by a C++ expression such as
   "render_kernel<false, 48, false, false, 135, 11, false>" and returns the size of its gfx950 code object, or a negative ky_status. */
int64_t     kyhip_jit_compile(const char* name_expression);
/* 64 bits over the text of the device headers the render kernels are compiled from (ky_device.hpp, ky_render.hpp, this file) and their compile flags:
   what identifies the KERNELS of a build (host-side changes do not move it).  profiles/valu.json carries the hash of the build its counters were
   measured on, and bench.py withholds those counters when it runs another. */
uint64_t    kyhip_kernel_source_hash(void);
int         kyhip_abi_version(void);
int         kyhip_device_count(void);

/* Number of tiles this (first, step) shard owns, and the float count of its compact tile buffer
   (tiles * tile_w * tile_h * 3).  Pure host arithmetic. */
int64_t kyhip_shard_tile_count(const ky_render_params* params);
int64_t kyhip_shard_float_count(const ky_render_params* params);

/*
 * kyhip_render -- drop-in for integrator_t::render(scene, sampler, film) (ky.cpp:3689-3729).
 *   film_rgb           caller-owned HOST buffer of film_t::pixels_ layout (AoS RGB fp32, row-major,
 *                      y down, ky.cpp:1574); the library ADDS clamp01(mean radiance) per pixel,
 *                      exactly like film_t::add_color (1586-1590, 3726).
 *   film_row_stride_px row stride in pixels (>= params->width), and film_rgb may point at the
 *                      first pixel of a sub-film, so a film_grid_t cell can be the target (1817-1822).
 * Renders the shard named by params (all tiles when tile_first = 0, tile_step = 1) on `device`.
 * Blocking.
 */
int kyhip_render(int device, const ky_scene* scene, const ky_render_params* params,
                 float* film_rgb, size_t film_row_stride_px);

/*
 * Device-resident variants used by the multi-GPU path and by bench.py (inputs and outputs stay in
 * HBM; nothing crosses PCIe inside the timed region).
 *
 * kyhip_render_tiles_device: renders this shard's tiles into `d_tiles`, a DEVICE buffer of
 *   kyhip_shard_float_count() floats laid out [local_tile][ty][tx][rgb]; every pixel is written
 *   with clamp01(mean radiance) (pixels of edge tiles outside the film are written as 0).
 *   `stream` is a hipStream_t (0 = default stream).  Asynchronous w.r.t. the host.
 *   `d_workspace`/`workspace_bytes`: device scratch of at least kyhip_workspace_bytes(); may be
 *   NULL, in which case the library allocates and caches one per device.
 *
 * kyhip_film_add_tiles_device: the de-interleave step after the gather: adds the compact tiles of
 *   shard (tile_first, tile_step) into a DEVICE film (same layout as kyhip_render's film_rgb).
 *
 * Streams: what a launch writes (work counter, accumulator workspace, timing events, shadow-ray stacks) belongs to the STREAM it is
 *   enqueued on -- the library keeps one such state per (device, stream), up to eight per device -- so calls on one stream execute in
 *   stream order and calls on different streams share nothing and may overlap on the device: a frame's kernel starts on the compute
 *   units the previous frame's persistent kernel is draining from (ky_amd/dist.py alternates two streams).  The packed scene is cached
 *   per device by content (eight slots): alternating between a few scenes uploads each once.  Calls for different devices are
 *   independent (one lock per device).
 *
 * kyhip_film_add_gathered_device: the same de-interleave for ALL shards of a frame in one kernel.  The frame described
 *   by params (tile_first, tile_step) was rendered as `world` shards -- shard r = (tile_first + r * tile_step,
 *   tile_step * world) -- whose compact tile buffers lie at d_gathered + r * rank_stride_floats (the layout a gather
 *   collective leaves on the root); rank_stride_floats >= kyhip_shard_float_count() of shard 0.
 */
size_t kyhip_workspace_bytes(const ky_render_params* params);
int kyhip_render_tiles_device(int device, const ky_scene* scene, const ky_render_params* params,
                              float* d_tiles, void* d_workspace, size_t workspace_bytes, void* stream);
int kyhip_film_add_tiles_device(int device, const ky_render_params* params, const float* d_tiles,
                                float* d_film_rgb, size_t film_row_stride_px, void* stream);
int kyhip_film_add_gathered_device(int device, const ky_render_params* params, int world, const float* d_gathered,
                                   size_t rank_stride_floats, float* d_film_rgb, size_t film_row_stride_px, void* stream);

/*
 * kyhip_render_multi -- integrator_t::render(scene, sampler, film) on a LIST of devices of one node.  The reference
 * spreads render()'s pixel loop over all cores itself (`#pragma omp parallel for schedule(dynamic, 1)`, ky.cpp:3696-3699);
 * here the frame's tiles are interleaved over the listed GPUs: shard i = (tile_first + i * tile_step, tile_step * n_devices)
 * renders on devices[i] on that device's own stream, all shards concurrently and without communication; the tile buffers
 * are then gathered on devices[0] (one peer copy per remote shard over xGMI), de-interleaved by one kernel and ADDED into
 * the caller's host film exactly like kyhip_render.  A device may be listed more than once (its shards then run one after
 * the other).  The image does not depend on the device list: it is bit-identical to kyhip_render's.  Blocking.
 * kyhip_render(device, ...) is kyhip_render_multi(&device, 1, ...).
 * kyhip_multi_status(root): how the last kyhip_render_multi call whose devices[0] was `root` gathered its shards, one clause per shard -- "local" (rendered on
 *   the root), "peer" (hipMemcpyPeerAsync over a peer mapping that hipDeviceEnablePeerAccess set up: a direct xGMI copy; the mapping is made ONCE per
 *   (root, device) pair and remembered) or "staged" (the runtime reports no peer access: the same call, which the runtime then stages through the host) --
 *   and how many host threads added the film.  Valid until the next call on this thread.
 * kyhip_film_alloc(bytes) / kyhip_film_free: film memory the GPU can add to IN PLACE -- pinned, mapped host memory (hipHostMalloc).  kyhip_render /
 *   kyhip_render_multi recognise such a film (or one the caller registered with hipHostRegister) and let the root GPU's add kernel read and write it over
 *   PCIe: no staging copy, no host pass, no dependence on the CPUs the process is granted.  Any other film (malloc, new[], numpy) takes the banded
 *   download + host add.  The images are bit-identical.  NULL without a device: callers fall back to ordinary memory (ky.hpp's film_t does).
 * kyhip_seam_threads(): host threads the banded add of the next kyhip_render / kyhip_render_multi call will use (the caller's plus parked ones): the CPUs
 *   the process is granted (affinity mask and cgroup quota), at most four (environment variable KYHIP_SEAM_THREADS overrides).  A pinned film uses none.
 */
void*       kyhip_film_alloc(size_t bytes);
void        kyhip_film_free(void* film);
const char* kyhip_multi_status(int root_device);
int kyhip_seam_threads(void);
int kyhip_render_multi(const int* devices, int n_devices, const ky_scene* scene, const ky_render_params* params,
                       float* film_rgb, size_t film_row_stride_px);

/*
 * Duration in milliseconds of the integrator kernel (render_kernel) of the most recent
 * kyhip_render* call on `device`, from hipEvents recorded on the launch stream around that one
 * kernel.  The stream must have been synchronised.  Negative if no timing is available.
 */
float kyhip_kernel_ms(int device);
/* Which render-kernel instantiation that call launched, e.g. "render_kernel<strategy 48, feat 7, integrator 11>" (the library holds one
   per direct-lighting strategy, integrator and set of scene facts; ky_launch.hip, g_variants).  The string stays valid until the next call
   of this function from the same thread.  "" if nothing was launched. */
const char* kyhip_last_kernel(int device);

/* ------------------------------------------------------------------------------------------
 * Function-level entry points (known-answer tests).  Each runs the DEVICE implementation of one
 * reference function over n host-side inputs and copies the results back.  Blocking.
 * ---------------------------------------------------------------------------------------- */

/* shape_t::intersect (1111, 1179, 1261, 1336).  rays: n x {o[3], d[3], tmax}.
   out: n x {hit(0/1), t, p[3], n[3]}. */
int kyhip_kat_intersect(int device, const ky_shape* shape, const float* rays7, int n, float* out8);

/* camera_t::generate_ray (1884-1892).  p_film: n x {x, y}.  out: n x {o[3], d[3]}. */
int kyhip_kat_camera(int device, const ky_camera* camera, const float* p_film2, int n, float* out6);

/* surface_t::intersect + material_t::scattering + bsdf_t::{sample, eval, pdf} (3077, 2587-2671, 2162-2179).
   in: n x {normal[3], wo[3], u[2], wi_eval[3], lobe_u}.  out: n x {f[3], wi[3], pdf, flags, eval[3], pdf_eval, is_delta}. */
int kyhip_kat_bsdf(int device, const ky_material* material, const float* in12, int n, float* out13);

/* light_t::sample_Li / pdf_Li (2825, 2891, 2964, 3026).
   in: n x {p[3], normal[3], u[2], wi[3]}.  out: n x {position[3], wi[3], pdf, Li[3], pdf_Li}. */
int kyhip_kat_light(int device, const ky_scene* scene, int light, const float* in11, int n, float* out11);

/* scene_t::intersect (3172) and scene_t::occluded (3187-3201).
   rays7 as above.  out: n x {hit, t, p[3], n[3], surface}.
   occluded: in n x {p[3], normal[3], target[3]}, out n x {0/1}. */
int kyhip_kat_scene_intersect(int device, const ky_scene* scene, const float* rays7, int n, float* out9);
int kyhip_kat_occluded(int device, const ky_scene* scene, const float* in9, int n, float* out1);
/* The any-hit pair scan of the environment light's both_mis estimate (round 6: trace_any_pair, ky_device.hpp): per row two rays through ONE scan of every surface --
   ray A without an end ("does it leave the scene": scene_t::intersect finds nothing, 4000), ray B ending at tmax_b (scene_t::occluded's scan, 3193-3195) -- with the box
   traversal when the scene has boxes.  in: n x {o_a[3], d_a[3], o_b[3], d_b[3], tmax_b}.  out: n x {A meets a surface 0/1, B meets one before tmax_b 0/1}. */
int kyhip_kat_any_pair(int device, const ky_scene* scene, const float* in13, int n, float* out2);
/* The same query against the occluder tables the render kernels use for shadow rays ("occluder tables").
   light = -1: p and target are promised to lie on surfaces, area lights' shapes or point lights of the scene; rectangles that have the
   whole scene in one closed half-space of their plane (the walls of a room) are not tested.
   light >= 0: additionally target is a point of the shape area light `light` samples and p lies in front of it (where light_t::sample_Li
   returns a non-black Li, 2957-2960); for a planar sampled shape the surfaces on or behind its plane are not tested either.
   For segments that keep the promise the answer equals kyhip_kat_occluded's. */
int kyhip_kat_occluded_between(int device, const ky_scene* scene, int light, const float* in9, int n, float* out1);
/* Host only (no GPU needed): left_out[i] = 1 when surface i (the caller's index) is not in the occluder table for `light` (-1: the table
   for rays that end on a scene point), 2 when it is scanned only for rays with an end behind that light's plane (the two-stage scan:
   what is mounted behind a lamp), 0 when it is always tested.  n = entries in left_out, at least scene->surface_count.  Returns the
   number of 1s, or a negative ky_status. */
int kyhip_scene_non_occluders(const ky_scene* scene, int light, int* left_out, int n);
/* Host only (no GPU needed): the axis-aligned boxes whose faces are surfaces of the scene, which a nearest-hit traversal (scene_t::intersect, ky.cpp:3172-3184) tests
   with one slab test each instead of face by face ("boxes"): box_face[i] = 8 * box + 2 * axis + side (side 0: the box's low plane on that axis, 1: its high
   plane) when surface i (the caller's index) is such a face, -1 otherwise.  n = entries in box_face, at least scene->surface_count.  Returns the number of boxes
   (0 with kyhip_set_specialisation(0)), or a negative ky_status. */
int kyhip_scene_boxes(const ky_scene* scene, int* box_face, int n);
/* Host only: the facts the library finds for a scene when it packs it -- a mask of the KY_FEAT_* values of ky_amd/csrc/ky_scene.hpp (1 exactly one area light, 2 every
   area light samples a rectangle, 4 few carrier surfaces per light, 8 exactly one point / directional light, 16 exactly one environment light, 32 sphere lamps only,
   64 no mirror or glass, 128 at most 16 surfaces and 8 materials, 256 every lamp is its own one carrier, 512 boxes, 1024 every planar surface is a rectangle in an axis
   plane, 2048 every plastic surface is a rectangle, 4096 every tilted rectangle is a plank about the x axis) -- which decide the render-kernel instantiation a launch takes (kyhip_last_kernel names it).  0 with kyhip_set_specialisation(0); negative: a ky_status. */
int kyhip_scene_facts(const ky_scene* scene);

/* integrator_t::Li per camera sample (3714-3717): for pixel (x, y) and samples [s0, s0+n) writes the
   unclamped radiance Li (3 floats per sample) -- the quantity `dL` is built from. */
int kyhip_kat_li(int device, const ky_scene* scene, const ky_render_params* params,
                 int x, int y, int s0, int n, float* out3);

/* estimate_direct_lighting_{by_bsdf, by_emitter, by_bsdf_mis, by_emitter_mis} (3889-4074) for ONE light at given vertices with given
   random numbers.  in15: n x {position[3], normal[3], wo[3], surface (index into scene->surfaces), lobe_u (the plastic material's lobe
   number, 2663), random_bsdf[2], random_light[2]}; out6: n x {the BSDF-sampling half [3], the light-sampling half [3]}.
   direct_sample 4 / 8: the plain estimators (by_bsdf uses random_bsdf as the float2 it draws itself, 3900); 16 / 32: the MIS halves;
   48: both MIS halves (not yet weighted by the 0.5 of 4083).  Vertices on delta surfaces return zeros (4571). */
int kyhip_kat_nee(int device, const ky_scene* scene, int direct_sample, int light, const float* in15, int n, float* out6);

/* The same sample of path_tracing_iteration_t vertex by vertex (the reference's LOG_VAST at ky.cpp:4578 prints the same facts):
   one row of 26 floats per vertex that reaches the continuation sample (4586):
   {bounces, surface (caller's index), lobe (0 lambert, 1 mirror, 2 glass, 3 phong), position[3], normal[3], wo[3],
    beta[3] before the bounce, Lo[3] after this vertex's direct lighting, bs.f[3], bs.pdf, |dot(bs.wi, normal)|, bsdf flags,
    bits of the lights whose BSDF-sampling estimate was non-black, bits of the lights whose light-sampling estimate was}.
   Returns the number of rows written (<= max_rows) or a negative ky_status; li3 (optional) receives the sample's radiance. */
int kyhip_kat_li_trace(int device, const ky_scene* scene, const ky_render_params* params, int x, int y, int s,
                       float* rows26, int max_rows, float* li3);

/* ---- the smallpt lineage's own scene, in double precision -----------------------------------------
   smallpt2pbrt/smallpt.cpp is the 99-line path tracer ky grew out of (smallpt_milo.cpp is its 256 x 256 build):
   9 spheres -- the walls are spheres of radius 1e5, which is why ky's fp32 sphere test cannot render this scene
    -- recursive radiance() with Russian roulette after 5 bounces and a two-way split at the glass
   sphere for the first two bounces (smallpt.cpp:56-89), 2 x 2 subpixels with a tent filter, per-subpixel clamp
   (91-118).  Random numbers: smallpt seeds erand48 per image ROW and walks the row sequentially (98); this library
   gives every (pixel, subpixel, sample) its own stream keyed by (seed, pixel, subpixel, sample) and consumes numbers in
   radiance()'s order, reflection subtree before transmission subtree. */
enum ky_smallpt_refl { KY_SP_DIFF = 0, KY_SP_SPEC = 1, KY_SP_REFR = 2 };   /* Refl_t, smallpt.cpp:24 */

typedef struct ky_smallpt_sphere {   /* struct Sphere, smallpt.cpp:26-39 */
    double rad;
    double p[3], e[3], c[3];         /* position, emission, colour */
    int refl;
    int pad_;
} ky_smallpt_sphere;

/* Which program of the smallpt lineage the fp64 path restates.
   KY_SP_VARIANT_SMALLPT  smallpt2pbrt/smallpt.cpp as described above.
   KY_SP_VARIANT_REWRITE  smallpt2pbrt/smallpt_rewrite.cpp ("structured smallpt", the pbrt-style step towards ky.cpp; the one
                          reference program that builds in this image and therefore pins this path, oracle/_ref): the nine
                          spheres mirrored in z (1199-1244), PerspectiveCamera fov 53 (1391), RandomSampler (uniform jitter,
                          `samps` samples per PIXEL, 369-392), RecursionPathIntegrater(max_depth 10) (1335-1372), one clamp per
                          pixel (1316).  Streams are keyed (seed, pixel, sample). */
enum ky_smallpt_variant { KY_SP_VARIANT_SMALLPT = 0, KY_SP_VARIANT_REWRITE = 1 };

typedef struct ky_smallpt_params {
    int width, height;
    int samps;                       /* variant 0: samples per SUBPIXEL (smallpt.cpp:92: argv[1] / 4), spp = 4 * samps;
                                        variant 1: samples per pixel (smallpt_rewrite.cpp:1388: argv[1] / 4) */
    uint32_t seed;
    int max_depth;                   /* `if (depth > 10) return obj.e` (smallpt.cpp:63) / RecursionPathIntegrater(10) (1396): 10 */
    int variant;                     /* ky_smallpt_variant */
} ky_smallpt_params;

/* The scene of smallpt.cpp:42-52; `out` has room for 9 spheres.  Returns 9.  Pure host code. */
int kyhip_smallpt_scene(ky_smallpt_sphere* out);
/* Scene::CreateSmallptScene of smallpt_rewrite.cpp:1199-1244 (the same spheres at -z).  Returns 9.  Pure host code. */
int kyhip_smallpt_scene_rewrite(ky_smallpt_sphere* out);

/* main()'s loop nest (smallpt.cpp:91-118) with the fixed camera of 93-94.  image_rgb: width * height * 3 doubles in
   smallpt's own order, c[(height - y - 1) * width + x], i.e. row 0 is the TOP of the picture; the image is overwritten.
   Variant 1: Integrater::Render (smallpt_rewrite.cpp:1287-1318) into a cleared Film, pixels_[y * width + x] with y = 0 the
   top row (517-520).  For variant 1 the kat entry ignores (sx, sy), which must be 0. */
int kyhip_smallpt_render(int device, const ky_smallpt_sphere* spheres, int n_spheres, const ky_smallpt_params* params,
                         double* image_rgb);

/* radiance(Ray(cam.o + d * 140, d.norm()), 0, Xi) (smallpt.cpp:109) for samples [s0, s0 + n) of subpixel (sx, sy) of
   pixel (x, y): 3 doubles per sample, unclamped. */
int kyhip_smallpt_kat_radiance(int device, const ky_smallpt_sphere* spheres, int n_spheres, const ky_smallpt_params* params,
                               int x, int y, int sx, int sy, int s0, int n, double* out3);

#ifdef __cplusplus
}
#endif
#endif /* KYHIP_H */
