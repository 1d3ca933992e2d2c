/*
 * ky_hostcheck.cpp -- entry points of the SANITIZER builds only (`make sanitize`: g++ -fsanitize=address,undefined and -fsanitize=thread over ky_pack.cpp,
 * ky_jit.cpp and this file; never part of libkyhip.so).  They drive the host code that has no C-ABI entry of its own -- scene packing into a heap DScene,
 * the scene cache's keys, the chunk schedule, the banded add, HostPool and the seam's lock order under contention and across a fork -- so that
 * tests/test_sanitize.py can run it under the sanitizers from Python (ctypes) or from the stress binary (tools/sanitize/stress.cpp).
 */
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <sys/wait.h>
#include <unistd.h>

#include "ky_host.hpp"

using namespace kyh;

extern "C" {

// pack_scene + scene_hash + scene_input on the caller's scene; returns pack_scene's status, the facts and both hashes
int kyhostcheck_pack(const ky_scene* scene, int* feat, uint64_t* packed_hash, uint64_t* input_hash, int* n_planar_occ, int* ts_light) {
    std::unique_ptr<DScene> d(new DScene);
    const int rc = pack_scene(scene, d.get());
    if (rc != KY_OK) return rc;
    if (feat) *feat = d->feat;
    if (packed_hash) *packed_hash = scene_hash(*d);
    std::vector<unsigned char> in;
    uint64_t h = 0;
    const bool keyed = scene_input(scene, in, h);
    if (input_hash) *input_hash = keyed ? h : 0;
    if (n_planar_occ) *n_planar_occ = d->occ.n_aar + d->occ.n_par;
    if (ts_light) *ts_light = d->ts_light;
    // every index the device follows must stay inside its table
    for (int i = 0; i < d->n_lights; ++i) {
        const DLight& L = d->light[i];
        for (int k = 0; k < L.n_carriers; ++k)
            if (L.carrier[k] < 0 || L.carrier[k] >= d->n_surfaces) return fail(KY_ERR_DEVICE, "carrier index out of range");
        const unsigned tab = (unsigned)L.shadow_table & ~1u;
        if (tab != __builtin_offsetof(DScene, occ_front) && tab != __builtin_offsetof(DScene, occ) && tab != __builtin_offsetof(DScene, trav)) return fail(KY_ERR_DEVICE, "shadow_table is not a table");
    }
    for (int j = 0; j < d->n_surfaces; ++j)
        if (d->orig[j] < 0 || d->orig[j] >= scene->surface_count || d->hit[j].material < 0 || d->hit[j].material >= d->n_materials) return fail(KY_ERR_DEVICE, "surface table out of range");
    return KY_OK;
}

// the chunk schedule of `spp` samples covers [0, spp) exactly once, in order; returns the chunk count or -1
int kyhostcheck_chunks(int spp) {
    const ChunkPlan p = chunk_plan(spp);
    const int n = chunk_count(p);
    int next = 0;
    for (int c = 0; c < n; ++c) {
        int b, e;
        chunk_range(p, c, b, e);
        if (b != next || e <= b || e - b > KY_CHUNK) return -1;
        next = e;
    }
    return next == spp ? n : -1;
}

// make_shard / shard_in_range / valid_params on the caller's parameters: n_items, or a negative status
long long kyhostcheck_shard(const ky_render_params* p) {
    if (!valid_params(p)) return KY_ERR_INVALID_VALUE;
    if (!shard_in_range(p)) return KY_ERR_LIMIT;
    const ShardConst s = make_shard(p);
    return (long long)s.n_items;
}

// film += src over rows [y0, y1) by n_threads workers of the pool (what the seam's step 4 does); returns 0 when the sum is right
int kyhostcheck_add_rows(int width, int height, int stride_px, int n_threads, int rounds) {
    std::vector<float> film((size_t)height * stride_px * 3, 1.f), src((size_t)height * width * 3);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (float)(i % 7);
    for (int r = 0; r < rounds; ++r)
        host_pool().run(n_threads, [&](int t) {
            const int r0 = (int)((long long)height * t / n_threads), r1 = (int)((long long)height * (t + 1) / n_threads);
            host_add_rows(film.data(), (size_t)stride_px, src.data(), width, r0, r1);
        });
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < stride_px * 3; ++x) {
            const float want = x < width * 3 ? 1.f + rounds * (float)(((size_t)y * width * 3 + x) % 7) : 1.f;
            if (film[(size_t)y * stride_px * 3 + x] != want) return 1 + y;
        }
    return 0;
}

// Two caller threads x two "devices": each call takes the seam mutexes of its device list in ascending device order (lock_seams, what
// kyhip_render_multi does), runs a pool job inside, releases; the lists overlap in opposite orders.  In between the process forks once and the
// child runs a pool job of its own (a forked child inherits the pool object, not its threads).  Returns 0, or what went wrong.
int kyhostcheck_seam_stress(int iterations) {
    std::mutex seam[2];
    std::atomic<long long> total{0};
    std::atomic<int> bad{0};
    auto call = [&](std::vector<int> devices) {
        std::vector<std::pair<int, std::mutex*>> want;
        for (int d : devices) want.emplace_back(d, &seam[d]);
        auto locks = lock_seams(want);
        long long local[4] = {0, 0, 0, 0};
        host_pool().run(3, [&](int t) { local[t] += t + 1; });
        if (local[0] != 1 || local[1] != 2 || local[2] != 3) bad.fetch_add(1);
        total.fetch_add(local[0] + local[1] + local[2]);
    };
    auto worker = [&](bool flip) {
        for (int i = 0; i < iterations; ++i) call(flip ? std::vector<int>{1, 0} : std::vector<int>{0, 1, 0});
    };
    {
        std::thread a(worker, false), b(worker, true);
        a.join(); b.join();
    }
    const pid_t pid = fork();
    if (pid == 0) {   // the child: its pool must come up again by itself
        long long local[3] = {0, 0, 0};
        host_pool().run(3, [&](int t) { local[t] = t + 1; });
        _exit(local[0] == 1 && local[1] == 2 && local[2] == 3 ? 0 : 7);
    }
    int status = 0;
    if (pid < 0 || waitpid(pid, &status, 0) != pid || !WIFEXITED(status) || WEXITSTATUS(status) != 0) return 100;
    {
        std::thread a(worker, true), b(worker, false);
        a.join(); b.join();
    }
    if (bad.load()) return 200;
    return total.load() == 4LL * iterations * 6 ? 0 : 300;
}

// Several threads ask the code cache for the same and for different instantiations at once, blocking and not (KYHIP_HIPCC points the cache at a
// stand-in compiler: tests/test_sanitize.py); returns the number of requests that ended with an object
int kyhostcheck_jit_stress(int n_threads, int rounds) {
    std::atomic<int> got{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t] {
            for (int r = 0; r < rounds; ++r) {
                char args[96];
                snprintf(args, sizeof args, "false, 48, false, false, %d, 11, false", (t + r) % 3);
                bool pending = false;
                const kyjit::Code* c = kyjit::get_code(args, (t & 1) == 0, &pending);
                for (int spin = 0; !c && pending && spin < 2000; ++spin) { usleep(1000); c = kyjit::get_code(args, false, &pending); }
                if (c && !c->object.empty()) got.fetch_add(1);
            }
        });
    for (auto& x : th) x.join();
    return got.load();
}


// ---- the entry points that need a GPU: absent from this build.  They exist as symbols because the host mirror (ky.hpp) and ctypes resolve every
// symbol when a library is loaded; each validates what the product validates before it touches a device where a CPU test looks at that, and then
// reports that there is no device -- exactly what libkyhip.so reports on a machine without a gfx950 GPU.
static int no_gpu() { return fail(KY_ERR_NO_DEVICE, "sanitizer build of the host-only code: no GPU entry points"); }
int kyhip_device_count(void) { return 0; }
int kyhip_render(int, const ky_scene*, const ky_render_params* p, float*, size_t) { return valid_params(p) ? no_gpu() : fail(KY_ERR_INVALID_VALUE, "invalid render params"); }
int kyhip_render_multi(const int*, int, const ky_scene*, const ky_render_params* p, float*, size_t) { return valid_params(p) ? no_gpu() : fail(KY_ERR_INVALID_VALUE, "invalid render params"); }
int kyhip_render_tiles_device(int, const ky_scene*, const ky_render_params* p, float*, void*, size_t, void*) { return valid_params(p) ? no_gpu() : fail(KY_ERR_INVALID_VALUE, "invalid render params"); }
int kyhip_film_add_tiles_device(int, const ky_render_params*, const float*, float*, size_t, void*) { return no_gpu(); }
int kyhip_film_add_gathered_device(int, const ky_render_params*, int, const float*, size_t, float*, size_t, void*) { return no_gpu(); }
float kyhip_kernel_ms(int) { return -1.f; }
const char* kyhip_last_kernel(int) { return ""; }
const char* kyhip_multi_status(int) { return ""; }
void* kyhip_film_alloc(size_t) { return nullptr; }   // no device: callers fall back to ordinary memory
void kyhip_film_free(void*) {}
int kyhip_kat_intersect(int, const ky_shape*, const float*, int, float*) { return no_gpu(); }
int kyhip_kat_camera(int, const ky_camera*, const float*, int, float*) { return no_gpu(); }
int kyhip_kat_bsdf(int, const ky_material*, const float*, int, float*) { return no_gpu(); }
int kyhip_kat_light(int, const ky_scene*, int, const float*, int, float*) { return no_gpu(); }
int kyhip_kat_scene_intersect(int, const ky_scene*, const float*, int, float*) { return no_gpu(); }
int kyhip_kat_occluded(int, const ky_scene*, const float*, int, float*) { return no_gpu(); }
int kyhip_kat_any_pair(int, const ky_scene*, const float*, int, float*) { return no_gpu(); }
int kyhip_kat_occluded_between(int, const ky_scene*, int, const float*, int, float*) { return no_gpu(); }
int kyhip_kat_li(int, const ky_scene*, const ky_render_params*, int, int, int, int, float*) { return no_gpu(); }
int kyhip_kat_nee(int, const ky_scene*, int, int, const float*, int, float*) { return no_gpu(); }
int kyhip_kat_li_trace(int, const ky_scene*, const ky_render_params*, int, int, int, float*, int, float*) { return no_gpu(); }
int kyhip_smallpt_render(int, const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p, double* image) {
    const int rc = smallpt_check(spheres, n, p);
    if (rc != KY_OK) return rc;
    return image ? no_gpu() : fail(KY_ERR_INVALID_VALUE, "null image");
}
int kyhip_smallpt_kat_radiance(int, const ky_smallpt_sphere* spheres, int n, const ky_smallpt_params* p, int x, int y, int sx, int sy, int s0, int cnt, double* out3) {
    const int rc = smallpt_check(spheres, n, p);
    if (rc != KY_OK) return rc;
    if (!out3 || cnt <= 0 || s0 < 0 || x < 0 || y < 0 || x >= p->width || y >= p->height || (sx | sy) < 0 || sx > 1 || sy > 1) return fail(KY_ERR_INVALID_VALUE, "bad KAT arguments");
    if (p->variant == KY_SP_VARIANT_REWRITE && (sx | sy) != 0) return fail(KY_ERR_INVALID_VALUE, "variant 1 has no subpixels: sx = sy = 0");
    return no_gpu();
}

}  // extern "C"
