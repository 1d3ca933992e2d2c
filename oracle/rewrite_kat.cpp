// oracle/rewrite_kat.cpp -- TEST INFRASTRUCTURE (never linked into the product).
//
// A harness around the one reference translation unit this image can build: it #includes
// /root/reference/smallpt2pbrt/smallpt_rewrite.cpp UNMODIFIED (only `main` is renamed, by a macro, so that this file can have
// its own) and calls the reference's own classes on inputs read from a file:
//     Frame(n) + ToLocal / ToWorld            smallpt_rewrite.cpp:122-176   -> ancestor of ky.cpp frame_t 526-578
//     Sphere::Intersect                        smallpt_rewrite.cpp:706-786   -> ky.cpp sphere_t::intersect 1336-1393
//     PerspectiveCamera ctor + GenerateRay     smallpt_rewrite.cpp:651-694   -> ky.cpp camera_t 1864-1892
//     LambertionReflection f / Pdf             smallpt_rewrite.cpp:873-905   -> ky.cpp lambertion_reflection_t 2227-2257
//     SpecularReflection Sample_f              smallpt_rewrite.cpp:907-934   -> ky.cpp perfect_specular_reflection_t 2292-2307
//     CosineSampleHemisphere's lift            smallpt_rewrite.cpp:259-265   -> ky.cpp cosine_hemisphere_sample 737-745
//     GammaEncoding                            smallpt_rewrite.cpp:494       -> ky.cpp gamma_encoding 1548
//     AreaLight::Le through Primitive::Intersect   smallpt_rewrite.cpp:1114-1117, 1135-1146 -> ky.cpp areal_radiance 2957-2960 / 2977 + surface_t::intersect 3077-3088
//                                              (one-sided: the light's radiance where dot(normal, wo) > 0, black seen from behind / inside)        [round 5]
//     Scene::Intersect                         smallpt_rewrite.cpp:1184-1197 -> ky.cpp scene_t::intersect 3172-3184: the whole list is scanned, the
//                                              ray's distance shrinks, a later surface at EXACTLY the same distance does not replace an earlier one     [round 5]
// Everything is fp64 there and fp32 in ky.cpp; tests/test_rewrite_kat.py compares oracle/ky_oracle.cpp's fp32 restatement (and the
// HIP path) with these values where ky.cpp kept the formula, and lists where it did not (tests/golden/make_rewrite_kat.py).
// Built by `make -C oracle ref` into oracle/_ref/rewrite_kat, only where /root/reference exists; run by
// tests/golden/make_rewrite_kat.py, whose output tests/golden/rewrite_kat.npz is what travels.
//
// File format (little-endian): input = int64 counts {n_frame, n_sphere, n_camera_sets, n_bsdf, n_lift, n_gamma, n_le, n_scene_sets} followed by the
// records as doubles; output = the results as doubles, in the same order.  Record layouts are at each loop.
#define main smallpt_rewrite_main
#include "smallpt_rewrite.cpp"   // found through -I$(REF)/smallpt2pbrt: the file is compiled where it lies
#undef main

#include <cstdint>
#include <cstdio>
#include <vector>

static std::vector<double> g_in;
static size_t g_pos = 0;
static double next() { return g_in.at(g_pos++); }
static Vector3 next3() { const double x = next(), y = next(), z = next(); return Vector3(x, y, z); }
static void put3(std::vector<double>& out, const Vector3& v) { out.push_back(v.x); out.push_back(v.y); out.push_back(v.z); }

int main(int argc, char** argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: rewrite_kat in.bin out.bin\n"); return 2; }
    FILE* fi = std::fopen(argv[1], "rb");
    if (!fi) return 3;
    int64_t counts[8];
    if (std::fread(counts, sizeof(int64_t), 8, fi) != 8) return 4;
    double d;
    while (std::fread(&d, sizeof d, 1, fi) == 1) g_in.push_back(d);
    std::fclose(fi);
    std::vector<double> out;

    // Frame: in {n[3], v[3]} -> out {s[3], t[3], n[3], ToLocal(v)[3], ToWorld(v)[3]}
    for (int64_t i = 0; i < counts[0]; ++i) {
        const Vector3 n = next3(), v = next3();
        const Frame f(n);
        put3(out, f.Binormal()); put3(out, f.Tangent()); put3(out, f.Normal());
        put3(out, f.ToLocal(v)); put3(out, f.ToWorld(v));
    }
    // Sphere::Intersect: in {center[3], radius, origin[3], direction[3], tmax} -> out {hit, t, position[3], normal[3]}
    for (int64_t i = 0; i < counts[1]; ++i) {
        const Vector3 c = next3();
        const double radius = next();
        const Vector3 o = next3(), dir = next3();
        const double tmax = next();
        const Sphere sphere(radius, c);
        Ray ray(o, dir, tmax);
        Isect isect;
        const bool hit = sphere.Intersect(ray, &isect);
        out.push_back(hit ? 1.0 : 0.0);
        out.push_back(hit ? ray.distance : 0.0);
        put3(out, hit ? isect.position : Vector3()); put3(out, hit ? isect.normal : Vector3());
    }
    // PerspectiveCamera: per set in {position[3], front[3] (unit), up[3], fov, res[2], n_samples, then n_samples x pFilm[2]}
    //                    -> out per sample {direction[3]}  (the rewrite starts its rays 140 units in: only the direction carries over)
    for (int64_t s = 0; s < counts[2]; ++s) {
        const Vector3 position = next3(), front = next3(), up = next3();
        const double fov = next(), rx = next(), ry = next();
        const int64_t n = (int64_t)next();
        const PerspectiveCamera camera(position, front, up, fov, Vector2(rx, ry));
        for (int64_t i = 0; i < n; ++i) {
            CameraSample cs;
            const double px = next(), py = next();   // (two calls in one argument list would be evaluated right to left by GCC)
            cs.pFilm = Vector2(px, py);
            const Ray ray = camera.GenerateRay(cs);
            put3(out, ray.direction);
        }
    }
    // BSDFs: in {normal[3], wo[3], wi[3], R[3]} -> out {lambert f[3], lambert pdf, mirror f[3], mirror wi[3], mirror pdf}
    for (int64_t i = 0; i < counts[3]; ++i) {
        const Vector3 n = next3(), wo = next3(), wi = next3(), R = next3();
        const LambertionReflection lambert(Frame(n), R);
        put3(out, lambert.f(wo, wi));
        out.push_back(lambert.Pdf(wo, wi));
        const SpecularReflection mirror(Frame(n), R);
        const BSDFSample ms = mirror.Sample_f(wo, Float2(0.5, 0.5));
        put3(out, ms.f); put3(out, ms.wi); out.push_back(ms.pdf);
    }
    // the lift of CosineSampleHemisphere, isolated from its disk mapping (ky.cpp maps the disk concentrically, 710-733; the rewrite by
    // polar coordinates, 252-257): in {u[2]} -> out {disk x, disk y, z}: z = sqrt(max(0, 1 - x^2 - y^2)) is what both share
    for (int64_t i = 0; i < counts[4]; ++i) {
        const double u0 = next(), u1 = next();
        put3(out, CosineSampleHemisphere(Float2(u0, u1)));
    }
    // GammaEncoding: in {x} -> out {byte}
    for (int64_t i = 0; i < counts[5]; ++i) out.push_back((double)GammaEncoding(next()));

    // AreaLight::Le as a path ray sees it (Primitive::Intersect sets isect.emission = areaLight->Le(isect, isect.wo)): one emitting sphere per row,
    // rays from outside and from inside.  in {center[3], radius, radiance[3], origin[3], direction[3]} -> out {hit, t, emission[3]}
    const MatteMaterial black{Color()};
    for (int64_t i = 0; i < counts[6]; ++i) {
        const Vector3 c = next3();
        const double radius = next();
        const Vector3 L = next3(), o = next3(), dir = next3();
        const Sphere sphere(radius, c);
        const AreaLight light(Color(L.x, L.y, L.z), &sphere);
        const Primitive prim{&sphere, &black, &light};
        Ray ray(o, dir);
        Isect isect;
        const bool hit = prim.Intersect(ray, &isect);
        out.push_back(hit ? 1.0 : 0.0);
        out.push_back(hit ? ray.distance : 0.0);
        const Color e = hit ? isect.Le() : Color();
        out.push_back(e.x); out.push_back(e.y); out.push_back(e.z);
    }
    // Scene::Intersect over a list of spheres (some of them twice: exact ties).  Every primitive carries an area light of radiance (index + 1, 0, 0), so the
    // emission of the surviving hit names the primitive the scan kept (all origins lie outside every sphere: the nearest hit faces the ray).
    // per set: in {n_spheres, n_spheres x {center[3], radius}, n_rays, n_rays x {origin[3], direction[3], tmax}} -> out per ray {hit, t, index, position[3]}
    for (int64_t s = 0; s < counts[7]; ++s) {
        const int64_t ns = (int64_t)next();
        std::vector<std::shared_ptr<Shape>> shapes;
        std::vector<std::shared_ptr<Material>> materials{std::make_shared<MatteMaterial>(Color())};
        std::vector<std::shared_ptr<Light>> lights;
        std::vector<std::shared_ptr<AreaLight>> area;
        std::vector<Primitive> prims;
        for (int64_t k = 0; k < ns; ++k) {
            const Vector3 c = next3();
            const double radius = next();
            shapes.push_back(std::make_shared<Sphere>(radius, c));
            area.push_back(std::make_shared<AreaLight>(Color((double)(k + 1), 0, 0), shapes.back().get()));
            lights.push_back(area.back());
            prims.push_back(Primitive{shapes.back().get(), materials[0].get(), area.back().get()});
        }
        const Scene scene(shapes, materials, lights, prims);
        const int64_t nr = (int64_t)next();
        for (int64_t i = 0; i < nr; ++i) {
            const Vector3 o = next3(), dir = next3();
            const double tmax = next();
            Ray ray(o, dir, tmax);
            Isect isect;
            const bool hit = scene.Intersect(ray, &isect);
            out.push_back(hit ? 1.0 : 0.0);
            out.push_back(hit ? ray.distance : 0.0);
            out.push_back(hit ? isect.Le().x - 1.0 : -1.0);
            put3(out, hit ? isect.position : Vector3());
        }
    }

    FILE* fo = std::fopen(argv[2], "wb");
    if (!fo) return 5;
    std::fwrite(out.data(), sizeof(double), out.size(), fo);
    std::fclose(fo);
    return g_pos == g_in.size() ? 0 : 6;
}
