/*
 * ky_seam.cpp -- the drop-in boundary's host-film calls: kyhip_render / kyhip_render_multi (include/kyhip.h), what stands behind
 * integrator_t::render(&scene, sampler, &film) (ky.cpp:3689; ky_amd/host/ky.hpp).  Shards of the frame go to the listed devices' own streams
 * (kyhip_render_tiles_device), the tile buffers are gathered on the first device, one kernel adds them into a device film, and the film comes
 * home in row bands that a few parked host threads add into the caller's buffer (film_t::add_color, 1586-1590).  HIP runtime calls only: no kernel
 * is defined here.
 */
#include <algorithm>
#include <string>
#include <vector>

#include "ky_ctx.hpp"

using namespace kyh;

// integrator_t::render on a LIST of devices (the reference spreads the pixel loop over all cores inside render(),
// ky.cpp:3696-3699).  Shard i of the frame goes to devices[i] on that device's own stream; the tile buffers are gathered on
// devices[0] (peer copies over xGMI), de-interleaved by one kernel and added into the caller's film.
//
// What the call owns besides the kernels is kept per device and reused (SeamBuffers): the gather block and the device film on the root, a
// PINNED host staging film, the tile buffers of remote shards.  The film comes back in row bands: band b's download is followed by an
// event, and a few host threads add band b into the caller's film (film_t::add_color, 1586-1590) the moment its event has fired, so the
// host's pass over the film overlaps the rest of the download.  (Round 3 allocated and freed two device buffers per call, downloaded into
// pageable memory and added with one scalar loop afterwards: 1-2 ms on a 9.4 MB film, a third of a 64-spp frame.)
static int seam_reserve(void** p, size_t* have, size_t need, bool pinned) {
    if (*have >= need) return KY_OK;
    if (*p) { HIP_TRY(pinned ? hipHostFree(*p) : hipFree(*p)); *p = nullptr; *have = 0; }
    const size_t bytes = need + need / 4;   // some slack: a caller that alternates frame sizes does not reallocate on every call
    HIP_TRY(pinned ? hipHostMalloc(p, bytes) : hipMalloc(p, bytes < 16 ? 16 : bytes));
    *have = bytes;
    return KY_OK;
}

extern "C" {

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
    std::vector<std::pair<int, std::mutex*>> seams;
    for (int i = 0; i < n_devices; ++i) seams.emplace_back(devices[i], &ctx[i]->seam.m);
    const std::vector<std::unique_lock<std::mutex>> seam_locks = lock_seams(std::move(seams));

    // buffers: one gather block and the film on the root, the pinned staging film; a tile buffer per remote shard on its device
    HIP_TRY(hipSetDevice(root));
    SeamBuffers& sb = ctx[0]->seam;
    // A film in PINNED host memory (kyhip_film_alloc: what ky.hpp's film_t allocates; or memory the caller registered with the runtime) is added to IN PLACE by the
    // root GPU: the add kernel reads and writes the host film over PCIe (12 B in, 12 B out per pixel, both directions at once), and the staging film, its download and
    // the host threads' pass drop out -- the seam then costs the same whatever CPUs the process is granted (round 5; a pageable film takes the banded path below).
    float* film_in_place = nullptr;
    {
        const size_t span = ((size_t)(p->height - 1) * stride_px + (size_t)p->width) * 3 * sizeof(float);
        hipPointerAttribute_t first{}, last{};
        void *dptr = nullptr, *dlast = nullptr;
        //  both ends pinned is not enough: two adjacent registrations, or one whose device mapping is not one piece, would pass.  The film's
        // whole span must lie in ONE mapping: the device alias of its last byte is span - 1 bytes beyond the alias of its first, and where the runtime knows
        // the allocation behind the alias (hipMemGetAddressRange) the span lies inside it.
        if (hipPointerGetAttributes(&first, film_rgb) == hipSuccess && first.type == hipMemoryTypeHost &&
            hipPointerGetAttributes(&last, (const char*)film_rgb + span - 1) == hipSuccess && last.type == hipMemoryTypeHost &&
            hipHostGetDevicePointer(&dptr, film_rgb, 0) == hipSuccess && dptr &&
            hipHostGetDevicePointer(&dlast, (char*)film_rgb + span - 1, 0) == hipSuccess && dlast == (char*)dptr + span - 1) {
            hipDeviceptr_t base = nullptr;
            size_t size = 0;
            const bool known = hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)dptr) == hipSuccess && base && size;
            if (!known || ((char*)dptr >= (char*)base && (char*)dptr + span <= (char*)base + size)) film_in_place = (float*)dptr;
        }
        (void)hipGetLastError();   // "not a registered pointer" is the ordinary answer for a pageable film
    }
    int rcode = seam_reserve(&sb.d_gather, &sb.gather_bytes, rank_stride * n_devices * sizeof(float), false);
    if (rcode == KY_OK && !film_in_place) rcode = seam_reserve(&sb.d_film, &sb.film_bytes, film_floats * sizeof(float), false);
    if (rcode == KY_OK && !film_in_place) rcode = seam_reserve((void**)&sb.h_stage, &sb.stage_bytes, film_floats * sizeof(float), true);
    if (rcode != KY_OK) return rcode;
    for (hipEvent_t& e : sb.band)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    float* const d_gather = (float*)sb.d_gather;
    float* const d_film = (float*)sb.d_film;
    hipStream_t root_stream = ctx[0]->stream;
    if (!film_in_place) HIP_TRY(hipMemsetAsync(d_film, 0, film_floats * sizeof(float), root_stream));
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
    kyjit::frame_begin();   // asynchronous run-time instantiations (kyhip_set_jit(2)): every shard of this frame renders on the same kernel
    struct FrameEnd { ~FrameEnd() { kyjit::frame_end(); } } frame_end_guard;
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
            // the root's mapping of this device's memory: asked for ONCE per (root, device) pair and remembered (round 4 asked the runtime on every call)
            if (sb.peer.size() <= (size_t)devices[i]) sb.peer.resize((size_t)devices[i] + 1, 0);
            if (sb.peer[devices[i]] == 0) {
                int can = 0;
                sb.peer[devices[i]] = 2;
                if (hipDeviceCanAccessPeer(&can, root, devices[i]) == hipSuccess && can) {   // direct xGMI copies; staged by the runtime otherwise
                    HIP_CHECK_BREAK(hipSetDevice(root));
                    const hipError_t pe = hipDeviceEnablePeerAccess(devices[i], 0);
                    (void)hipGetLastError();   // "already enabled" is not an error here
                    if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) sb.peer[devices[i]] = 1;
                }
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
        if (film_in_place) rcode = kyhip_film_add_gathered_device(root, p, n_devices, d_gather, rank_stride, film_in_place, stride_px, root_stream);   // film_t::add_color, by the GPU, in the caller's film
        else rcode = kyhip_film_add_gathered_device(root, p, n_devices, d_gather, rank_stride, d_film, (size_t)p->width, root_stream);
    }
    // 3. the film comes home in row bands, each followed by an event
    const size_t row_bytes = (size_t)p->width * 3 * sizeof(float);
    int n_bands = (int)std::min<size_t>(KY_SEAM_BANDS, std::max<size_t>(1, film_floats * sizeof(float) / (512u << 10)));   // bands of at least 512 KB
    n_bands = std::min(n_bands, p->height);
    if (film_in_place) n_bands = 0;   // nothing comes home: the film was added to where it lies
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
    if (rcode == KY_OK && bands_enqueued == n_bands && n_bands > 0) {
        const int n_threads = std::max(1, std::min(p->height, seam_threads()));
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
    {   // kyhip_multi_status(root)
        std::string st = std::to_string(n_devices) + " shard(s):";
        for (int i = 0; i < n_devices; ++i) {
            const char* how = devices[i] == root ? "local" : ((size_t)devices[i] < sb.peer.size() && sb.peer[devices[i]] == 1 ? "peer" : "staged");
            st += std::string(i ? "," : "") + " device " + std::to_string(devices[i]) + " " + how;
        }
        st += film_in_place ? std::string("; film added in place by the GPU (pinned host film)")
                            : "; film added by " + std::to_string(std::max(1, std::min(p->height, seam_threads()))) + " host thread(s)";
        sb.last_status = st;
    }
    if (rcode != KY_OK) return rcode;
    if (sync_err != hipSuccess) return fail(KY_ERR_DEVICE, "render failed: %s (the caller's film may hold a part of the frame)", hipGetErrorString(sync_err));
    return KY_OK;
}

const char* kyhip_multi_status(int root_device) {
    static thread_local std::string s;
    s.clear();
    DeviceCtx* c = find_ctx(root_device);
    if (c) { std::lock_guard<std::mutex> lock(c->seam.m); s = c->seam.last_status; }
    return s.c_str();
}

// Film memory the root GPU can add to in place (kyhip_render_multi): pinned, mapped host memory.  NULL when there is no device or no memory; kyhip_film_free
// takes what kyhip_film_alloc returned (and NULL).
void* kyhip_film_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void kyhip_film_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int kyhip_render(int device, const ky_scene* scene, const ky_render_params* p, float* film_rgb, size_t stride_px) {
    return kyhip_render_multi(&device, 1, scene, p, film_rgb, stride_px);
}

}  // extern "C"
