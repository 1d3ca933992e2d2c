/*
 * ky_drivers.cpp -- the reference's experiment drivers (ky.cpp:4675-4949) written against ky_amd/host/ky.hpp,
 * i.e. the same call shape `use_devices(*integrator); integrator->render(&scene, sampler.get(), &film); film.next_subfilm();` with the
 * rendering done by the MI355X library.  Workload definitions only: spp, grids and scene flags are the reference's.
 *
 *   ky_drivers single [spp4]     render_single_scene       (4675): 1024x1024 Cornell + environment light
 *   ky_drivers debug             render_debug              (4715): Veach position / normal / basecolor, 1x3 grid
 *   ky_drivers multiple_integrator render_multiple_integrator (4740): 4 Cornell lights x 5 integrators, 4x5 grid
 *   ky_drivers direct_sample     render_direct_sample_enum (4779): 4 Cornell lights x 5 strategies, 4x5 grid
 *   ky_drivers multiple_scene    render_multiple_scene     (4819): 3 strategies x 4 Cornell lights, 3x4 grid
 *   ky_drivers mis               render_mis_scene          (4878): Veach x 6 strategies, 2x3 grid
 *   ky_drivers lighting_enum [spp] [w] [h]  BASELINE.json configs[1]: the scene of the reference's (commented-out) render_lighting_enum
 *                                (4907-4935: Cornell, both small spheres, area light) at 1024 x 768, 1024 spp, path_tracing_iteration d5
 *                                both_mis; writes lighting_enum.bmp
 *   ky_drivers batch [spp] [res] BASELINE.json configs[3]: render_multiple_scene scaled up -- the four Cornell light variants
 *                                (both_mis), Veach (both_mis) and a first-hit AOV pass, each res x res (1024) at spp (2048),
 *                                into a film_grid_t(2, 3, res, res); writes batch.bmp
 *   ky_drivers stress [spp] [res] BASELINE.json configs[4]: Cornell res x res (4096), spp (16384), max depth 16; writes stress.bmp
 * An optional last argument multiplies every spp (the reference's values are tiny because its CPU path is slow).
 * KY_DEVICES=all (or a count n: devices 0 .. n-1) makes every integrator spread its tiles over that many GPUs of the node
 * (integrator_t::set_devices); the images do not depend on it.
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "../ky_amd/host/ky.hpp"

using namespace ky;

static int g_spp_scale = 1;

// KY_DEVICES: the GPUs every integrator of this process renders on
static void use_devices(integrator_t& integrator) {
    const char* e = std::getenv("KY_DEVICES");
    if (!e || !*e) return;
    std::vector<int> all = integrator_t::all_devices();
    if (std::strcmp(e, "all") != 0) {
        const int n = std::atoi(e);
        if (n < 1) return;
        all.resize((size_t)n);
        for (int i = 0; i < n; ++i) all[(size_t)i] = i;
    }
    integrator.set_devices(all);
}

template <typename F>
static double timing_seconds(F f) {  // wall clock (the reference's clock() counts CPU time on Linux, ky.cpp:156-163)
    const auto t0 = std::chrono::steady_clock::now();
    f();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

static void render_single_scene(int spp4) {
    const int width = 1024, height = 1024;
    film_t film(width, height);
    scene_t scene = scene_t::create_cornell_box_scene(cornell_box_enum_t::both_small_spheres | cornell_box_enum_t::light_environment, film.get_resolution());
    const int samples_per_pixel = (spp4 > 0 ? spp4 / 4 : 16) * g_spp_scale;
    std::unique_ptr<sampler_t> sampler = std::make_unique<random_sampler_t>(samples_per_pixel);
    auto integrator = create_integrator(integrator_enum_t::path_tracing_iteration, 5, direct_sample_enum_t::both_mis);
    use_devices(*integrator);
    const double seconds = timing_seconds([&] { integrator->render(&scene, sampler.get(), &film); });
    std::printf("%d spp, %.3f seconds (kernel %.3f ms), %.1f Msamples/s\n", samples_per_pixel, seconds, integrator->last_kernel_ms(),
                (double)width * height * samples_per_pixel / seconds / 1e6);
    film.store_image("single");
}

static void render_debug() {
    film_grid_t film(1, 3, 512, 308);
    std::unique_ptr<sampler_t> sampler = std::make_unique<random_sampler_t>(10 * g_spp_scale);
    scene_t scene = scene_t::create_mis_scene(film.get_resolution());
    for (auto e : {integrator_enum_t::position, integrator_enum_t::normal, integrator_enum_t::basecolor}) {
        std::unique_ptr<integrator_t> integrator = std::make_unique<debug_integrator_t>(e);
        use_devices(*integrator); integrator->render(&scene, sampler.get(), &film);
        film.next_subfilm();
    }
    film.store_image("render_debug");
}

static const std::vector<std::pair<cornell_box_enum_t, int>>& scene_params(bool multiple_scene) {
    static const std::vector<std::pair<cornell_box_enum_t, int>> a{{cornell_box_enum_t::light_point, 1}, {cornell_box_enum_t::light_direction, 10},
                                                                   {cornell_box_enum_t::light_area, 1}, {cornell_box_enum_t::light_environment, 10}};
    static const std::vector<std::pair<cornell_box_enum_t, int>> b{{cornell_box_enum_t::light_point, 10}, {cornell_box_enum_t::light_direction, 40},
                                                                   {cornell_box_enum_t::light_area, 40}, {cornell_box_enum_t::light_environment, 10}};
    return multiple_scene ? b : a;
}

static void render_multiple_integrator() {
    const std::vector<integrator_enum_t> integrator_enums{integrator_enum_t::direct_lighting, integrator_enum_t::simple_path_tracing_recursion,
                                                          integrator_enum_t::path_tracing_recursion, integrator_enum_t::path_tracing_recursion_defered,
                                                          integrator_enum_t::path_tracing_iteration};
    film_grid_t film(4, 5, 256, 256);
    for (auto [scene_enum, spp] : scene_params(false)) {
        scene_t scene = scene_t::create_cornell_box_scene(cornell_box_enum_t::both_small_spheres | scene_enum, film.get_resolution());
        std::unique_ptr<sampler_t> sampler = std::make_unique<random_sampler_t>(spp * g_spp_scale);
        for (auto integrator_enum : integrator_enums) {
            auto integrator = create_integrator(integrator_enum, 5, direct_sample_enum_t::both_mis);
            use_devices(*integrator); integrator->render(&scene, sampler.get(), &film);
            film.next_subfilm();
        }
    }
    film.store_image("direct_sample");   // (sic) the reference writes this driver's mosaic under the same name, 4776
}

static void render_direct_sample_enum() {
    const std::vector<direct_sample_enum_t> sample_enums{direct_sample_enum_t::bsdf, direct_sample_enum_t::light, direct_sample_enum_t::bsdf_mis,
                                                        direct_sample_enum_t::light_mis, direct_sample_enum_t::both_mis};
    film_grid_t film(4, 5, 256, 256);
    for (auto [scene_enum, spp] : scene_params(false)) {
        std::unique_ptr<sampler_t> sampler = std::make_unique<random_sampler_t>(spp * g_spp_scale);
        scene_t scene = scene_t::create_cornell_box_scene(cornell_box_enum_t::both_small_spheres | scene_enum, film.get_resolution());
        for (auto sample_enum : sample_enums) {
            std::unique_ptr<integrator_t> integrator = std::make_unique<path_tracing_iteration_t>(5, sample_enum);
            use_devices(*integrator); integrator->render(&scene, sampler.get(), &film);
            film.next_subfilm();
        }
    }
    film.store_image("direct_sample");
}

static void render_multiple_scene() {
    const std::vector<direct_sample_enum_t> sample_enums{direct_sample_enum_t::bsdf, direct_sample_enum_t::light, direct_sample_enum_t::both_mis};
    film_grid_t film(3, 4, 256, 256);
    for (auto sample_enum : sample_enums) {
        std::unique_ptr<integrator_t> integrator = std::make_unique<path_tracing_iteration_t>(5, sample_enum);
        for (auto [scene_enum, spp] : scene_params(true)) {
            std::unique_ptr<sampler_t> sampler = std::make_unique<random_sampler_t>(spp * g_spp_scale);
            scene_t scene = scene_t::create_cornell_box_scene(cornell_box_enum_t::both_small_spheres | scene_enum, film.get_resolution());
            use_devices(*integrator); integrator->render(&scene, sampler.get(), &film);
            film.next_subfilm();
        }
    }
    film.store_image("light_mis");
}

static void render_mis_scene() {
    film_grid_t film(2, 3, 512, 308);
    std::unique_ptr<sampler_t> sampler = std::make_unique<random_sampler_t>(10 * g_spp_scale);
    scene_t scene = scene_t::create_mis_scene(film.get_resolution());
    for (auto sample_enum : {direct_sample_enum_t::bsdf, direct_sample_enum_t::light, direct_sample_enum_t::idle, direct_sample_enum_t::bsdf_mis,
                             direct_sample_enum_t::light_mis, direct_sample_enum_t::both_mis}) {
        std::unique_ptr<integrator_t> integrator = std::make_unique<path_tracing_iteration_t>(5, sample_enum);
        use_devices(*integrator); integrator->render(&scene, sampler.get(), &film);
        film.next_subfilm();
    }
    film.store_image("veach_mis");
}

// BASELINE.json configs[3]: the batch of render_multiple_scene (4819-4876) at production size -- one film_grid_t, one
// integrator->render() per cell, every frame's tiles interleaved over the GPUs of KY_DEVICES
static void render_batch(int spp, int res) {
    film_grid_t film(2, 3, res, res);
    double samples = 0;
    const double seconds = timing_seconds([&] {
        std::unique_ptr<sampler_t> sampler = std::make_unique<random_sampler_t>(spp);
        for (auto light : {cornell_box_enum_t::light_point, cornell_box_enum_t::light_direction, cornell_box_enum_t::light_area, cornell_box_enum_t::light_environment}) {
            scene_t scene = scene_t::create_cornell_box_scene(cornell_box_enum_t::both_small_spheres | light, film.get_resolution());
            std::unique_ptr<integrator_t> integrator = std::make_unique<path_tracing_iteration_t>(5, direct_sample_enum_t::both_mis);
            use_devices(*integrator); integrator->render(&scene, sampler.get(), &film);
            film.next_subfilm();
            samples += (double)res * res * spp;
        }
        {
            scene_t scene = scene_t::create_mis_scene(film.get_resolution());
            std::unique_ptr<integrator_t> integrator = std::make_unique<path_tracing_iteration_t>(5, direct_sample_enum_t::both_mis);
            use_devices(*integrator); integrator->render(&scene, sampler.get(), &film);
            film.next_subfilm();
            samples += (double)res * res * spp;
            std::unique_ptr<integrator_t> aov = std::make_unique<debug_integrator_t>(integrator_enum_t::normal);   // the "debug" member of the batch
            std::unique_ptr<sampler_t> one = std::make_unique<debug_sampler_t>(1);
            use_devices(*aov); aov->render(&scene, one.get(), &film);
            samples += (double)res * res;
        }
    });
    std::printf("batch: 6 frames %dx%d, %d spp: %.3f seconds, %.1f Msamples/s\n", res, res, spp, seconds, samples / seconds / 1e6);
    film.store_image("batch");
}

// BASELINE.json configs[1]: the headline frame
static void render_lighting_enum(int spp, int width, int height) {
    film_t film(width, height);
    scene_t scene = scene_t::create_cornell_box_scene(cornell_box_enum_t::both_small_spheres | cornell_box_enum_t::light_area, film.get_resolution());
    std::unique_ptr<sampler_t> sampler = std::make_unique<random_sampler_t>(spp);
    auto integrator = create_integrator(integrator_enum_t::path_tracing_iteration, 5, direct_sample_enum_t::both_mis);
    use_devices(*integrator);
    const double seconds = timing_seconds([&] { integrator->render(&scene, sampler.get(), &film); });
    std::printf("lighting_enum: %dx%d, %d spp: %.3f seconds (kernel %.3f ms), %.1f Msamples/s\n", width, height, spp, seconds, integrator->last_kernel_ms(),
                (double)width * height * spp / seconds / 1e6);
    film.store_image("lighting_enum");
}

// BASELINE.json configs[4]: the stress frame
static void render_stress(int spp, int res) {
    film_t film(res, res);
    scene_t scene = scene_t::create_cornell_box_scene(cornell_box_enum_t::both_small_spheres | cornell_box_enum_t::light_area, film.get_resolution());
    std::unique_ptr<sampler_t> sampler = std::make_unique<random_sampler_t>(spp);
    auto integrator = create_integrator(integrator_enum_t::path_tracing_iteration, 16, direct_sample_enum_t::both_mis);
    use_devices(*integrator);
    const double seconds = timing_seconds([&] { integrator->render(&scene, sampler.get(), &film); });
    std::printf("stress: %dx%d, %d spp, depth 16: %.3f seconds, %.1f Msamples/s\n", res, res, spp, seconds, (double)res * res * spp / seconds / 1e6);
    film.store_image("stress");
}

int main(int argc, char* argv[]) {
    const char* which = argc > 1 ? argv[1] : "single";
    try {
        if (!std::strcmp(which, "single")) {
            if (argc > 3) g_spp_scale = std::atoi(argv[3]);
            render_single_scene(argc > 2 ? std::atoi(argv[2]) : 0);
        } else if (!std::strcmp(which, "lighting_enum")) {
            render_lighting_enum(argc > 2 ? std::atoi(argv[2]) : 1024, argc > 3 ? std::atoi(argv[3]) : 1024, argc > 4 ? std::atoi(argv[4]) : 768);
        } else if (!std::strcmp(which, "batch")) {
            render_batch(argc > 2 ? std::atoi(argv[2]) : 2048, argc > 3 ? std::atoi(argv[3]) : 1024);
        } else if (!std::strcmp(which, "stress")) {
            render_stress(argc > 2 ? std::atoi(argv[2]) : 16384, argc > 3 ? std::atoi(argv[3]) : 4096);
        } else {
            if (argc > 2) g_spp_scale = std::atoi(argv[2]);
            if (g_spp_scale < 1) g_spp_scale = 1;
            if (!std::strcmp(which, "debug")) render_debug();
            else if (!std::strcmp(which, "multiple_integrator")) render_multiple_integrator();
            else if (!std::strcmp(which, "direct_sample")) render_direct_sample_enum();
            else if (!std::strcmp(which, "multiple_scene")) render_multiple_scene();
            else if (!std::strcmp(which, "mis")) render_mis_scene();
            else { std::fprintf(stderr, "unknown driver '%s'\n", which); return 2; }
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
