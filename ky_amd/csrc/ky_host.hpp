/*
 * ky_host.hpp -- what the host-side translation units of libkyhip.so share that needs NO HIP runtime: error reporting, parameter checks and shard
 * geometry, scene packing and the occluder classification, the launch policies (specialisation, deferred shadow rays, engine), the run-time
 * instantiations' code cache, and the small thread pool of the host-film seam.  Implemented in ky_pack.cpp and ky_jit.cpp, which are plain C++:
 * `make sanitize` builds them with g++ -fsanitize=address,undefined / thread next to the oracle and runs them through tests/test_sanitize.py.
 * (The HIP side -- device contexts, streams, the scene cache -- is ky_ctx.hpp.)
 */
#pragma once
#include <cstddef>
#include <cstdint>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include <sys/types.h>

#include "ky_shard.hpp"   // ky_scene.hpp (DScene ...), ShardConst, the chunk schedule

namespace kyh {
using namespace kyd;

// ---- errors: kyhip_last_error() returns the calling thread's last message ----
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const std::string& last_error();

// ---- parameters and shard geometry ----
bool valid_params(const ky_render_params* p);
bool shard_in_range(const ky_render_params* p);
ShardConst make_shard(const ky_render_params* p);
RenderConst make_rc(const ky_render_params* p);
inline size_t workspace_bytes_for(const ShardConst& s) { return (size_t)s.n_pix * (3 * sizeof(unsigned long long) + sizeof(unsigned)); }   // per pixel: 3 x 64-bit fixed-point sums + one flag word

// ---- scene packing (ky_pack.cpp) ----
void cp3(float* d, const float* s);
void pack_shape(const ky_shape& sh, int full_index, DSurf* surf, DShapeFull* full);
void pack_material(const ky_material& m, DMat* d);
bool shape_normal_ok(const ky_shape& sh);
struct NonOccluders {
    std::vector<char> wall;                  // [surface]
    std::vector<char> light_ok;              // [light]
    bool deferred_ok = false;                // all lights ok (the deferred shadow rays share one stack)
    int ts_light = -1;                       // two-stage scan (DScene::occ_front / occ_behind): the light, its plane n.x = k, and
    double ts_plane[4] = {0, 0, 0, 0};       // the surfaces that lie entirely in n.x <= k (not the light's own)
    std::vector<char> ts_behind;             // [surface]
};
void find_non_occluders(const ky_scene* in, NonOccluders& R);
// Axis-aligned boxes whose faces are surfaces of the scene (DBox, ky_scene.hpp): which, and which surface is which face.
struct Boxes {
    struct Box { float lo[3], hi[3]; int face[6]; };   // face[2 axis + side] = the caller's surface index, -1: an open side
    std::vector<Box> box;
    std::vector<int> box_of;                            // [surface] -> box, -1: none
};
void find_boxes(const ky_scene* in, Boxes& B);
int pack_scene(const ky_scene* in, DScene* out);
uint64_t scene_hash(const DScene& s);
bool scene_input(const ky_scene* in, std::vector<unsigned char>& out, uint64_t& hash);

// ---- the fp64 smallpt kernels' host side ----
constexpr int KY_SP_MAX_SPHERES = 32;   // == kysp::SP_MAX_SPHERES (ky_launch.hip asserts it)
int smallpt_check(const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p);

// ---- launch policies (each has an environment variable and a kyhip_set_* entry) ----
bool specialisation_enabled();
bool boxes_enabled();
int shadow_queue_mode();
bool shadow_queue_wanted(const ky_scene* scene);
int blocks_per_cu_cap();
enum { KY_ENGINE_LANE = 0, KY_ENGINE_QUEUE = 1 };
int current_engine();

// ---- the host-film seam's CPU side ----
int cpus_granted();
int seam_threads();   // host threads of the banded add (kyhip_seam_threads)
void host_add_rows(float* __restrict__ film, size_t stride_px, const float* __restrict__ src, int width, int y0, int y1);
// A few parked host threads for the banded add (creating and joining threads per call cost 50-100 us of a 4 ms frame).  run(n, fn) calls
// fn(0) ... fn(n - 1), fn(0) on the caller; one job at a time (the callers hold a seam mutex anyway, this one serialises across devices).
class HostPool {
public:
    void run(int n, const std::function<void(int)>& fn);
private:
    void loop(int id);
    std::mutex job_m_, m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0, pending_ = 0, n_threads_ = 0;
    pid_t pid_ = 0;
    unsigned long long generation_ = 0;
};
HostPool& host_pool();
// The seam mutexes of a call's device list, taken in ascending device order whatever order the list names them in (and once per device): two calls with
// overlapping lists cannot deadlock.  Never called while a context's enqueue mutex is held.
std::vector<std::unique_lock<std::mutex>> lock_seams(std::vector<std::pair<int, std::mutex*>> by_device);
}  // namespace kyh

// ---- run-time instantiations: the code cache (ky_jit.cpp) ----
namespace kyjit {
struct Code {   // one compiled instantiation
    std::vector<char> object;
    bool failed = false;
};
extern const char k_entry[];                       // the extern "C" name of every run-time instantiation's kernel
int mode();                                        // 0 off, 1 on (blocking: the first launch of a kind waits for its compile), 2 on, asynchronous
// the code object of render_kernel_body<args> from memory or disk, compiling it when neither has it.  wait = false: a missing object is
// compiled by a background thread and nullptr is returned until it is there (`*pending` says so); nullptr with !*pending: it cannot be had
// (kyhip_jit_status() says why)
const Code* get_code(const std::string& args, bool wait = true, bool* pending = nullptr);
uint64_t source_hash();
int set_mode(int m);                               // -> the previous mode
bool mode_by_default();                            // nobody chose the mode: it is default_mode()'s (the launch code then instantiates for fact-free / run-time-dispatched picks only)
std::string status();
int failures();                                    // compiles that failed so far in this process
// A frame that is rendered by several launches (kyhip_render_multi's shards) must not switch kernels in the middle: between frame_begin() and
// frame_end() the calling thread's non-blocking get_code() treats objects that were finished after frame_begin() as still pending.
void frame_begin();
void frame_end();
}  // namespace kyjit
