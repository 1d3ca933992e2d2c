/*
 * ky_ctx.hpp -- the HIP side the translation units of libkyhip.so share: one context per device (streams' launch state, the scene cache, the
 * host-film seam's buffers), created on first use.  Implemented in ky_launch.hip.
 */
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ky_host.hpp"

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return kyh::fail(KY_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace kyh {
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
    bool ws_clean = false;                 // `ws` and the work counter hold zeros: the last frame's resolve_kernel put them back (no fills before the next frame's launch)
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
    std::vector<char> peer;                              // root: per device ordinal, 0 not asked yet, 1 peer mapping enabled (direct copies), 2 no peer access (staged by the runtime)
    std::string last_status;                             // root: kyhip_multi_status()
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
    int variant_blocks[64] = {};           // resident workgroups per CU of g_variants[i] (0: not asked yet) ...
    size_t variant_lds[64] = {};           // ... for a scene block of this many bytes
    int last_variant = -1;
    int q_blocks_per_cu[3] = {0, 0, 0};
    struct JitKernel { hipModule_t module = nullptr; hipFunction_t fn = nullptr; int per_cu = 0; size_t lds = ~(size_t)0; bool failed = false; };
    std::map<std::string, JitKernel> jit;   // run-time instantiations loaded on this device, by template arguments
    std::string last_jit;                   // ... and the one the last launch used (last_variant == -3)
    std::string last_note;                  // what kyhip_last_kernel adds about the last launch's own instantiation: being compiled / unavailable / not asked for
};

int get_ctx(int device, DeviceCtx** out);       // looks the context of `device` up (creating it on first use) and makes the device current for the calling thread
DeviceCtx* find_ctx(int device);
int get_stream_state(DeviceCtx* c, hipStream_t stream, StreamState** out);
int upload_scene(DeviceCtx* c, const ky_scene* scene, hipStream_t stream, SceneSlot** out);
// The facts of the table kernel kyhip_render_tiles_device runs these parameters on that decide WHICH tests a ray goes through (KY_FEAT_BOXES: the box traversal;
// KY_FEAT_SINGLE_ENV: the environment estimate's any-hit pair scan): the per-sample replay entries (ky_kat.hip) take the same ones, so that a replayed sample takes
// every decision the rendered one took
int render_replay_feat(const ky_scene* scene, const ky_render_params* p, const kyd::DScene* packed);
bool render_uses_boxes(const ky_scene* scene, const ky_render_params* p, const kyd::DScene* packed);
}  // namespace kyh
