/*
 * ky_capi.cpp -- a thin C API over the C++ host layer (ky.hpp) so that Python (ctypes) can build the
 * reference's scenes, drive integrator_t::render() exactly as a C++ caller would, and use the film
 * writers.  Plumbing only: no rendering arithmetic lives here.
 */
#include <cstring>
#include <memory>
#include <string>

#include "ky.hpp"

static thread_local std::string g_host_error;

namespace {
struct scene_box {
    ky::scene_t scene;
};
template <typename F>
int guarded(F f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_host_error = e.what();
        return -1;
    }
}
}  // namespace

extern "C" {

const char* kyhost_last_error(void) { return g_host_error.c_str(); }

// scene_t::create_cornell_box_scene(flags, resolution), ky.cpp:3240.  NULL on error (both large balls: 3268-3271).
void* kyhost_scene_create_cornell_box(int flags, float width, float height) {
    scene_box* box = nullptr;
    guarded([&] { box = new scene_box{ky::scene_t::create_cornell_box_scene((ky::cornell_box_enum_t)flags, {width, height})}; });
    return box;
}

// scene_t::create_mis_scene(resolution), ky.cpp:3434
void* kyhost_scene_create_mis(float width, float height) {
    scene_box* box = nullptr;
    guarded([&] { box = new scene_box{ky::scene_t::create_mis_scene({width, height})}; });
    return box;
}

void kyhost_scene_destroy(void* scene) { delete (scene_box*)scene; }

// flat view (valid until the scene is destroyed)
const ky_scene* kyhost_scene_flatten(void* scene) {
    const ky_scene* flat = nullptr;
    guarded([&] { flat = &((scene_box*)scene)->scene.flatten(); });
    return flat;
}

// create_integrator(enum, depth, direct_sample)->render(&scene, sampler, &film) on a film_t (grid_rows = 0)
// or on cell `cell` of a film_grid_t(grid_rows, grid_cols, width, height): the call shape of every
// reference driver (ky.cpp:4697, 4732, 4770, 4810, 4851, 4899).  film points at the WHOLE film
// (grid_cols*width x grid_rows*height when a grid is used) and is accumulated into.
// Returns 0, -1 on error, -2 when create_integrator returns nullptr.
int kyhost_render(void* scene, int integrator_enum, int depth, int direct_sample_enum, int sampler_kind, int spp, unsigned seed,
                  int width, int height, int grid_rows, int grid_cols, int cell, float* film, int device) {
    int status = 0;
    int rc = guarded([&] {
        std::unique_ptr<ky::integrator_t> integrator;
        const auto ie = (ky::integrator_enum_t)integrator_enum;
        if (ie == ky::integrator_enum_t::position || ie == ky::integrator_enum_t::normal || ie == ky::integrator_enum_t::basecolor)
            integrator = std::make_unique<ky::debug_integrator_t>(ie, device < 0 ? 0 : device);                          // 4730-4731
        else
            integrator = ky::create_integrator(ie, depth, (ky::direct_sample_enum_t)direct_sample_enum, device < 0 ? 0 : device);  // 4621
        if (!integrator) { status = -2; return; }
        // device < 0 selects a device LIST: -1 = every visible GPU; -n (n >= 2) = device 0 listed n times, which drives the
        // multi-device path (shards, gather, one add) on a single GPU
        if (device == -1) integrator->set_devices(ky::integrator_t::all_devices());
        else if (device < -1) integrator->set_devices(std::vector<int>((size_t)-device, 0));
        std::unique_ptr<ky::sampler_t> sampler;
        if (sampler_kind == KY_SAMPLER_DEBUG) sampler = std::make_unique<ky::debug_sampler_t>(spp);
        else sampler = std::make_unique<ky::random_sampler_t>(spp);
        sampler->set_seed(seed);
        std::unique_ptr<ky::film_t> f;
        if (grid_rows > 0) {
            auto grid = std::make_unique<ky::film_grid_t>(grid_rows, grid_cols, width, height);
            for (int i = 0; i < cell; ++i) grid->next_subfilm();
            f = std::move(grid);
        } else {
            f = std::make_unique<ky::film_t>(width, height);
        }
        const size_t n = (size_t)f->get_pixel_num() * 3;
        std::memcpy(f->data(), film, n * sizeof(float));
        integrator->render(&((scene_box*)scene)->scene, sampler.get(), f.get());
        std::memcpy(film, f->data(), n * sizeof(float));
    });
    return rc != 0 ? rc : status;
}

// create_integrator(...)->debug_area(&scene, sampler, &film, {bx, by}, {ex, ey}) on a film_t of width x height (ky.cpp:3733-3777);
// `film` is read, modified and written back.  Returns 0, -1 on error, -2 when create_integrator returns nullptr.
int kyhost_debug_area(void* scene, int integrator_enum, int depth, int direct_sample_enum, int sampler_kind, int spp, unsigned seed,
                      int width, int height, float* film, int bx, int by, int ex, int ey, int device) {
    int status = 0;
    int rc = guarded([&] {
        std::unique_ptr<ky::integrator_t> integrator;
        const auto ie = (ky::integrator_enum_t)integrator_enum;
        if (ie == ky::integrator_enum_t::position || ie == ky::integrator_enum_t::normal || ie == ky::integrator_enum_t::basecolor)
            integrator = std::make_unique<ky::debug_integrator_t>(ie, device);
        else
            integrator = ky::create_integrator(ie, depth, (ky::direct_sample_enum_t)direct_sample_enum, device);
        if (!integrator) { status = -2; return; }
        std::unique_ptr<ky::sampler_t> sampler;
        if (sampler_kind == KY_SAMPLER_DEBUG) sampler = std::make_unique<ky::debug_sampler_t>(spp);
        else sampler = std::make_unique<ky::random_sampler_t>(spp);
        sampler->set_seed(seed);
        ky::film_t f(width, height);
        const size_t n = (size_t)f.get_pixel_num() * 3;
        std::memcpy(f.data(), film, n * sizeof(float));
        if (ex == bx + 1 && ey == by + 1) integrator->debug_pixel(&((scene_box*)scene)->scene, sampler.get(), &f, ky::point2_t((float)bx, (float)by));
        else integrator->debug_area(&((scene_box*)scene)->scene, sampler.get(), &f, ky::point2_t((float)bx, (float)by), ky::point2_t((float)ex, (float)ey));
        std::memcpy(film, f.data(), n * sizeof(float));
    });
    return rc != 0 ? rc : status;
}

// film writers, ky.cpp:1646-1782.  kind: 0 ppm, 1 bmp, 2 hdr (image_enum_t, 1531-1536)
int kyhost_store_image(const char* filename, int kind, int width, int height, const float* rgb) {
    bool ok = false;
    guarded([&] {
        switch (kind) {
        case 0: ok = ky::film_t::store_ppm_impl(filename, width, height, 3, rgb); break;
        case 1: ok = ky::film_t::store_bmp_impl(filename, width, height, 3, rgb); break;
        case 2: ok = ky::film_t::store_hdr_impl(filename, width, height, 3, rgb); break;
        default: g_host_error = "unknown image kind"; break;
        }
    });
    return ok ? 0 : -1;
}

int kyhost_gamma_encoding(float x) { return ky::gamma_encoding(x); }

}  // extern "C"
