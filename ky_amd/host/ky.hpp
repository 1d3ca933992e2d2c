/*
 * ky.hpp -- host-side C++ API with the shape of ky's Scene / Camera / Film / Sampler / Integrator
 *           classes, sitting on top of the C ABI in include/kyhip.h.
 *
 * The classes keep the reference's names, constructor arguments and call shapes
 *   (film_t 1553, film_grid_t 1802, camera_t 1859, the shape/material/light classes, surface_t 3071,
 *    scene_t 3147 incl. create_cornell_box_scene 3240 and create_mis_scene 3434, sampler_t 877,
 *    integrator_t::render 3689, create_integrator 4621 -- all line numbers in /root/reference/ky.cpp)
 * but hold DATA only: every computation of the hot path happens on the GPU behind
 * integrator_t::render().  There is no CPU rendering fallback in this layer.
 */
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/kyhip.h"

namespace ky {

using float_t = float;

// ---------------------------------------------------------------------------------------------
// small value types (ky.cpp:226-388).  Host code only builds scene data with them.
// ---------------------------------------------------------------------------------------------
struct color_t {
    float r{}, g{}, b{};
    color_t() = default;
    color_t(float r, float g, float b) : r(r), g(g), b(b) {}
    color_t operator*(float s) const { return {r * s, g * s, b * s}; }
    color_t operator/(float s) const { return {r / s, g / s, b / s}; }
    color_t operator+(color_t c) const { return {r + c.r, g + c.g, b + c.b}; }
    float luminance() const { return 0.212671f * r + 0.715160f * g + 0.072169f * b; }  // 249-255
};

struct vec2_t {
    float x{}, y{};
    vec2_t() = default;
    vec2_t(float x, float y) : x(x), y(y) {}
};
using point2_t = vec2_t;

struct vec3_t {
    float x{}, y{}, z{};
    vec3_t() = default;
    vec3_t(float x, float y, float z) : x(x), y(y), z(z) {}
    vec3_t operator-() const { return {-x, -y, -z}; }
    vec3_t operator+(vec3_t v) const { return {x + v.x, y + v.y, z + v.z}; }
    vec3_t operator-(vec3_t v) const { return {x - v.x, y - v.y, z - v.z}; }
    vec3_t operator*(float s) const { return {x * s, y * s, z * s}; }
    float dot(vec3_t v) const { return x * v.x + y * v.y + z * v.z; }
    vec3_t cross(vec3_t v) const { return {y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x}; }
    float magnitude() const { return std::sqrt(x * x + y * y + z * z); }
    vec3_t normalize() const { return *this * (1 / std::sqrt(x * x + y * y + z * z)); }  // 314
};
using point3_t = vec3_t;
using normal_t = vec3_t;
inline vec3_t cross(vec3_t a, vec3_t b) { return a.cross(b); }
inline vec3_t normalize(vec3_t v) { return v.normalize(); }
inline void store3(float* d, vec3_t v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline void store3(float* d, color_t c) { d[0] = c.r; d[1] = c.g; d[2] = c.b; }

constexpr float k_pi      = 3.14159265358979323846;
constexpr float k_inv_pi  = 0.318309886183790671538;
constexpr float k_inv_4pi = k_inv_pi / 4.f;
inline float radians(float degree) { return (k_pi / 180.f) * degree; }  // 190

// axis-aligned bounds + bounding sphere (ky.cpp:461-516): only needed by light preprocess.
struct bounds3_t {
    vec3_t min_{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    vec3_t max_{std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()};
    void grow(vec3_t p) {
        min_ = {std::min(min_.x, p.x), std::min(min_.y, p.y), std::min(min_.z, p.z)};
        max_ = {std::max(max_.x, p.x), std::max(max_.y, p.y), std::max(max_.z, p.z)};
    }
    void grow(const bounds3_t& b) { grow(b.min_); grow(b.max_); }
    void bounding_sphere(vec3_t* center, float* radius) const {  // 508-512
        *center = min_ + (max_ - min_) * 0.5f;
        bool inside = center->x >= min_.x && center->x <= max_.x && center->y >= min_.y && center->y <= max_.y &&
                      center->z >= min_.z && center->z <= max_.z;
        *radius = inside ? (*center - max_).magnitude() : 0;
    }
};

// ---------------------------------------------------------------------------------------------
// film (ky.cpp:1531-1836)
// ---------------------------------------------------------------------------------------------
inline float clamp01(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
inline uint8_t gamma_encoding(float x) { return (uint8_t)(std::pow((double)clamp01(x), 1 / 2.2) * 255 + .5); }  // 1548

class film_t {
public:
    // The reference's film owns its pixels (`new color_t[]`, 1559).  Here they come from kyhip_film_alloc -- pinned host memory the GPU adds to in place
    // (include/kyhip.h: no staging copy, no host pass at the end of render()) -- and from ordinary memory where that returns NULL (no device).
    film_t(int width, int height) : width_(width), height_(height), count_((size_t)width * height) {
        pixels_ = static_cast<color_t*>(kyhip_film_alloc(count_ * sizeof(color_t)));
        pinned_ = pixels_ != nullptr;
        if (!pixels_) pixels_ = new color_t[count_];
        for (size_t i = 0; i < count_; ++i) pixels_[i] = color_t{};
    }
    virtual ~film_t() {
        if (pinned_) kyhip_film_free(pixels_);
        else delete[] pixels_;
    }
    film_t(const film_t&) = delete;
    film_t& operator=(const film_t&) = delete;

    int get_width() const { return width_; }
    int get_height() const { return height_; }
    int get_pixel_num() const { return width_ * height_; }
    int get_channels() const { return 3; }

    virtual vec2_t get_resolution() const { return {(float)width_, (float)height_}; }  // 1569
    virtual color_t& operator()(int x, int y) { return pixels_[(size_t)width_ * y + x]; }  // 1570-1575
    void set_color(int x, int y, color_t c) { (*this)(x, y) = c; }
    void clear_color(int x, int y) { set_color(x, y, color_t{}); }
    void add_color(int x, int y, color_t d) { color_t& c = (*this)(x, y); c = c + d; }  // 1586-1590
    void clear(color_t c) { for (size_t i = 0; i < count_; ++i) pixels_[i] = c; }

    // raw view for the C ABI: pointer to the first float of the current render target and its row stride
    float* data() { return &pixels_[0].r; }
    const float* data() const { return &pixels_[0].r; }
    virtual float* target_origin() { return data(); }
    size_t row_stride_px() const { return (size_t)width_; }

    // store_image (1606-1644): BMP by default, like the reference; never spawns a viewer.
    virtual bool store_image(std::string filename) const { return store_bmp_impl(filename + ".bmp", width_, height_, 3, data()); }

    // 1646-1659
    static bool store_ppm_impl(const std::string& filename, int width, int height, int channel, const float* floats) {
        std::ofstream f(filename, std::ios::binary);
        if (!f) return false;
        f << "P3\n" << width << " " << height << "\n255\n";
        const int n = width * height * channel;
        for (int i = 0; i < n; ++i) f << (int)gamma_encoding(floats[i]) << " ";
        return true;
    }
    // 1661-1737: 24-bit bottom-up BGR BMP.  The row padding arithmetic is the reference's: the header
    // advertises padded rows but the body is written unpadded (identical bytes for widths % 4 == 0).
    static bool store_bmp_impl(const std::string& filename, int width, int height, int channel, const float* floats) {
        std::ofstream f(filename, std::ios::binary);
        if (!f) return false;
        const uint32_t line_padded = ((uint32_t)(width * channel) + 3u) & ~3u;
        const uint32_t body = line_padded * (uint32_t)height;
        auto u32 = [&](uint32_t v) { f.write((const char*)&v, 4); };
        auto u16 = [&](uint16_t v) { f.write((const char*)&v, 2); };
        f.write("BM", 2);
        u32(14 + 40 + body); u32(0); u32(14 + 40);                        // file header
        u32(40); u32((uint32_t)width); u32((uint32_t)height); u16(1); u16((uint16_t)(channel * 8));
        u32(0); u32(0); u32(0); u32(0); u32(0); u32(0);                   // compression .. colours
        std::vector<uint8_t> row((size_t)width * 3);
        for (int y = height - 1; y >= 0; --y) {
            const float* src = floats + (size_t)y * width * 3;
            for (int x = 0; x < width; ++x) {
                row[3 * x + 0] = gamma_encoding(src[3 * x + 2]);
                row[3 * x + 1] = gamma_encoding(src[3 * x + 1]);
                row[3 * x + 2] = gamma_encoding(src[3 * x + 0]);
            }
            f.write((const char*)row.data(), (std::streamsize)row.size());
        }
        return true;
    }
    // 1739-1782: Radiance RGBE, uncompressed scanlines
    static bool store_hdr_impl(const std::string& filename, int width, int height, int channel, const float* floats) {
        (void)channel;
        std::ofstream f(filename, std::ios::binary);
        if (!f) return false;
        f << "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " << height << " +X " << width << "\n";
        for (int i = 0; i < width * height; ++i) {
            uint8_t rgbe[4]{};
            const float r = floats[3 * i], g = floats[3 * i + 1], b = floats[3 * i + 2];
            const float v = std::max(r, std::max(g, b));
            if (v >= 1e-32f) {
                int e;
                const float m = (float)(std::frexp(v, &e) * 256.f / v);
                rgbe[0] = (uint8_t)(r * m); rgbe[1] = (uint8_t)(g * m); rgbe[2] = (uint8_t)(b * m); rgbe[3] = (uint8_t)(e + 128);
            }
            f.write((const char*)rgbe, 4);
        }
        return true;
    }

protected:
    int32_t width_{}, height_{};
    size_t count_;
    color_t* pixels_ = nullptr;
    bool pinned_ = false;
};

// mosaic of sub-films (1802-1836): render() targets the current cell
class film_grid_t : public film_t {
public:
    film_grid_t(int row, int column, int sub_width, int sub_height)
        : film_t(column * sub_width, row * sub_height), row_(row), column_(column), sub_width_(sub_width), sub_height_(sub_height) {}
    vec2_t get_resolution() const override { return {(float)sub_width_, (float)sub_height_}; }  // 1815
    color_t& operator()(int x, int y) override {                                                 // 1817-1822
        const int col = subfilm_index_ % column_, row = subfilm_index_ / column_;
        return film_t::operator()(x + col * sub_width_, y + row * sub_height_);
    }
    float* target_origin() override { return &(*this)(0, 0).r; }
    void next_subfilm() { ++subfilm_index_; }                                                    // 1824
private:
    int row_{}, column_{}, subfilm_index_{}, sub_width_{}, sub_height_{};
};

// ---------------------------------------------------------------------------------------------
// camera (ky.cpp:1859-1906)
// ---------------------------------------------------------------------------------------------
class camera_t {
public:
    camera_t(vec3_t position, vec3_t front, vec3_t up, float fov_degree, vec2_t resolution)
        : position_(position), front_(front.normalize()), up_(up.normalize()), resolution_(resolution) {
        const float tan_fov = std::tan(radians(fov_degree) / 2);                                 // 1875
        right_ = up_.cross(front_).normalize() * tan_fov * (resolution_.x / resolution_.y);      // 1878
        up_    = front_.cross(right_).normalize() * tan_fov;                                     // 1879
    }
    ky_camera flatten() const {
        ky_camera c{};
        store3(c.position, position_); store3(c.front, front_); store3(c.right, right_); store3(c.up, up_);
        c.resolution[0] = resolution_.x; c.resolution[1] = resolution_.y;
        return c;
    }
private:
    vec3_t position_, front_, right_, up_;
    vec2_t resolution_;
};
using const_camera_sptr_t = std::shared_ptr<const camera_t>;

// ---------------------------------------------------------------------------------------------
// shapes (ky.cpp:1009-1519): geometry records + the two host-side queries lights need
// ---------------------------------------------------------------------------------------------
class shape_t {
public:
    virtual ~shape_t() = default;
    virtual ky_shape flatten() const = 0;
    virtual bounds3_t world_bound() const = 0;
    virtual float area() const = 0;
};
using shape_sptr_t = std::shared_ptr<shape_t>;
using shape_list_t = std::vector<shape_sptr_t>;

class disk_t : public shape_t {
public:
    disk_t(point3_t position, normal_t normal, float radius) : position_(position), normal_(normalize(normal)), radius_(radius) {}
    ky_shape flatten() const override {
        ky_shape s{}; s.kind = KY_SHAPE_DISK; store3(s.p[0], position_); store3(s.normal, normal_); s.radius = radius_; return s;
    }
    bounds3_t world_bound() const override {  // 1134-1139: box spanned by +-(s + t) * r of the normal's frame
        const vec3_t a = std::abs(normal_.x) > 0.99f ? vec3_t(0, 1, 0) : vec3_t(1, 0, 0);
        const vec3_t t = normalize(cross(normal_, a)), s = normalize(cross(t, normal_));
        const vec3_t off = s * radius_ + t * radius_;
        bounds3_t b; b.grow(position_ - off); b.grow(position_ + off); return b;
    }
    float area() const override { return k_pi * radius_ * radius_; }
private:
    point3_t position_; normal_t normal_; float radius_;
};

class triangle_t : public shape_t {
public:
    triangle_t(point3_t p0, point3_t p1, point3_t p2, bool flip_normal = false) : p0_(p0), p1_(p1), p2_(p2) {
        normal_ = normalize(cross(p1_ - p0_, p2_ - p0_));                                        // 1174
        if (flip_normal) normal_ = -normal_;
    }
    ky_shape flatten() const override {
        ky_shape s{}; s.kind = KY_SHAPE_TRIANGLE; store3(s.p[0], p0_); store3(s.p[1], p1_); store3(s.p[2], p2_); store3(s.normal, normal_); return s;
    }
    bounds3_t world_bound() const override { bounds3_t b; b.grow(p0_); b.grow(p1_); b.grow(p2_); return b; }
    float area() const override { return 0.5f * cross(p1_ - p0_, p2_ - p0_).magnitude(); }
private:
    point3_t p0_, p1_, p2_; normal_t normal_;
};

class rectangle_t : public shape_t {
public:
    rectangle_t(point3_t p0, point3_t p1, point3_t p2, point3_t p3, bool flip_normal = false) : p0_(p0), p1_(p1), p2_(p2), p3_(p3) {
        normal_ = normalize(cross(p1_ - p0_, p2_ - p0_));                                        // 1256
        if (flip_normal) normal_ = -normal_;
    }
    ky_shape flatten() const override {
        ky_shape s{}; s.kind = KY_SHAPE_RECTANGLE;
        store3(s.p[0], p0_); store3(s.p[1], p1_); store3(s.p[2], p2_); store3(s.p[3], p3_); store3(s.normal, normal_);
        return s;
    }
    bounds3_t world_bound() const override { bounds3_t b; b.grow(p0_); b.grow(p1_); b.grow(p2_); b.grow(p3_); return b; }
    float area() const override { return cross(p0_ - p1_, p2_ - p1_).magnitude(); }              // 1304
private:
    point3_t p0_, p1_, p2_, p3_; normal_t normal_;
};

class sphere_t : public shape_t {
public:
    sphere_t(vec3_t center, float radius) : center_(center), radius_(radius) {}
    ky_shape flatten() const override { ky_shape s{}; s.kind = KY_SHAPE_SPHERE; store3(s.p[0], center_); s.radius = radius_; return s; }
    bounds3_t world_bound() const override {
        const vec3_t h(radius_, radius_, radius_);
        bounds3_t b; b.grow(center_ + h); b.grow(center_ - h); return b;
    }
    float area() const override { return 4 * k_pi * radius_ * radius_; }
private:
    vec3_t center_; float radius_;
};

// ---------------------------------------------------------------------------------------------
// materials (ky.cpp:2568-2682)
// ---------------------------------------------------------------------------------------------
class material_t {
public:
    virtual ~material_t() = default;
    virtual ky_material flatten() const = 0;
};
using material_sptr_t = std::shared_ptr<material_t>;
using material_list_t = std::vector<material_sptr_t>;

class matte_material_t : public material_t {
public:
    explicit matte_material_t(color_t diffuse_color) : diffuse_color_(diffuse_color) {}
    ky_material flatten() const override { ky_material m{}; m.kind = KY_MATERIAL_MATTE; store3(m.color0, diffuse_color_); return m; }
private:
    color_t diffuse_color_;
};
class mirror_material_t : public material_t {
public:
    explicit mirror_material_t(color_t specular_color) : specular_color_(specular_color) {}
    ky_material flatten() const override { ky_material m{}; m.kind = KY_MATERIAL_MIRROR; store3(m.color0, specular_color_); return m; }
private:
    color_t specular_color_;
};
class glass_material_t : public material_t {
public:
    explicit glass_material_t(float eta, color_t reflection_color = {1, 1, 1}, color_t transmission_color = {1, 1, 1})
        : eta_(eta), reflection_color_(reflection_color), transmission_color_(transmission_color) {}
    ky_material flatten() const override {
        ky_material m{}; m.kind = KY_MATERIAL_GLASS; m.eta = eta_; store3(m.color0, reflection_color_); store3(m.color1, transmission_color_); return m;
    }
private:
    float eta_; color_t reflection_color_, transmission_color_;
};
class plastic_material_t : public material_t {
public:
    plastic_material_t(color_t diffuse_color, color_t specular_color, float shininess)
        : diffuse_color_(diffuse_color), specular_color_(specular_color), exponent_(shininess) {
        const float diffuse = diffuse_color.luminance(), specular = specular_color.luminance();  // 2653-2658
        const float luminance = diffuse + specular;
        diffuse_probility_  = diffuse / luminance;
        specular_probility_ = specular / luminance;
    }
    ky_material flatten() const override {
        ky_material m{}; m.kind = KY_MATERIAL_PLASTIC; store3(m.color0, diffuse_color_); store3(m.color1, specular_color_);
        m.exponent = exponent_; m.diffuse_probability = diffuse_probility_; m.specular_probability = specular_probility_;
        return m;
    }
private:
    color_t diffuse_color_, specular_color_; float exponent_, diffuse_probility_{}, specular_probility_{};
};

// ---------------------------------------------------------------------------------------------
// lights (ky.cpp:2764-3062)
// ---------------------------------------------------------------------------------------------
class scene_t;
class light_t {
public:
    virtual ~light_t() = default;
    light_t(point3_t world_position, int samples_num = 1) : world_position_(world_position), samples_num_(samples_num) {}
    virtual bool is_delta() const = 0;
    virtual void preprocess(const scene_t&) {}
    virtual ky_light flatten(const scene_t& scene) const = 0;
protected:
    point3_t world_position_; int samples_num_;
};
using light_sptr_t = std::shared_ptr<light_t>;
using light_list_t = std::vector<light_sptr_t>;

class point_light_t : public light_t {
public:
    point_light_t(point3_t world_position, int samples_num, color_t intensity) : light_t(world_position, samples_num), intensity_(intensity) {}
    bool is_delta() const override { return true; }
    ky_light flatten(const scene_t&) const override {
        ky_light l{}; l.kind = KY_LIGHT_POINT; l.shape = -1; store3(l.color, intensity_); store3(l.position, world_position_); return l;
    }
private:
    color_t intensity_;
};
class direction_light_t : public light_t {
public:
    direction_light_t(point3_t world_position, int samples_num, color_t irradiance, vec3_t world_direction)
        : light_t(world_position, samples_num), irradiance_(irradiance), world_direction_(normalize(world_direction)) {}
    bool is_delta() const override { return true; }
    void preprocess(const scene_t& scene) override;                                               // 3555-3563
    ky_light flatten(const scene_t&) const override {
        ky_light l{}; l.kind = KY_LIGHT_DIRECTION; l.shape = -1; store3(l.color, irradiance_); store3(l.direction, world_direction_);
        l.world_radius = world_radius_; return l;
    }
private:
    color_t irradiance_; vec3_t world_direction_; vec3_t world_center_{}; float world_radius_{};
};
class area_light_t : public light_t {
public:
    area_light_t(point3_t world_position, int samples_num, color_t radiance, const shape_t* shape)
        : light_t(world_position, samples_num), radiance_(radiance), shape_(shape) {}
    bool is_delta() const override { return false; }
    ky_light flatten(const scene_t& scene) const override;
private:
    color_t radiance_; const shape_t* shape_;
};
class environment_light_t : public light_t {
public:
    environment_light_t(point3_t world_position, int samples_num, color_t radiance) : light_t(world_position, samples_num), radiance_(radiance) {}
    bool is_delta() const override { return false; }
    void preprocess(const scene_t& scene) override;                                               // 3565-3574
    ky_light flatten(const scene_t&) const override {
        ky_light l{}; l.kind = KY_LIGHT_ENVIRONMENT; l.shape = -1; store3(l.color, radiance_); l.world_radius = world_radius_; return l;
    }
private:
    color_t radiance_; vec3_t world_center_{}; float world_radius_{};
};

// surface_t (3071-3075)
struct surface_t {
    const shape_t* shape{};
    const material_t* material{};
    const area_light_t* area_light{};
};
using surface_list_t = std::vector<surface_t>;

// cornell_box_enum_t (3121-3145)
enum class cornell_box_enum_t {
    none, light_area = 1, light_direction = 2, light_point = 4, light_environment = 8,
    large_mirror_sphere = 16, large_glass_sphere = 32, small_mirror_sphere = 64, small_glass_sphere = 128, glossy_floor = 256,
    both_small_spheres = small_mirror_sphere | small_glass_sphere,
    both_large_spheres = large_mirror_sphere | large_glass_sphere,
    default_scene = both_small_spheres | light_area,
};
constexpr cornell_box_enum_t operator|(cornell_box_enum_t a, cornell_box_enum_t b) { return (cornell_box_enum_t)((int)a | (int)b); }
constexpr bool enum_have(cornell_box_enum_t group, cornell_box_enum_t value) { return ((int)group & (int)value) != 0; }

// ---------------------------------------------------------------------------------------------
// scene (ky.cpp:3147-3547)
// ---------------------------------------------------------------------------------------------
class scene_t {
public:
    scene_t() = default;
    scene_t(const_camera_sptr_t camera, shape_list_t shape_list, material_list_t material_list, light_list_t light_list,
            surface_list_t surface_list, environment_light_t* env_light = nullptr)
        : camera_(std::move(camera)), shape_list_(std::move(shape_list)), material_list_(std::move(material_list)),
          light_list_(std::move(light_list)), environment_light_(env_light), surface_list_(std::move(surface_list)) {
        for (light_sptr_t& light : light_list_) light->preprocess(*this);                         // 3163-3166
    }
    scene_t(scene_t&&) = default;
    scene_t& operator=(scene_t&&) = default;
    scene_t(const scene_t&) = delete;

    bounds3_t world_bound() const {                                                               // 3209-3219
        bounds3_t b;
        for (const surface_t& s : surface_list_) b.grow(s.shape->world_bound());
        return b;
    }
    const camera_t* get_camera() const { return camera_.get(); }
    int light_count() const { return (int)light_list_.size(); }
    const light_list_t& light_list() const { return light_list_; }
    const environment_light_t* environment_light() const { return environment_light_; }

    int shape_index(const shape_t* s) const {
        for (size_t i = 0; i < shape_list_.size(); ++i) if (shape_list_[i].get() == s) return (int)i;
        throw std::runtime_error("ky::scene_t: a surface or light refers to a shape that is not in shape_list");
    }

    // Flat view for the C ABI.  Pointers inside stay valid while this scene_t lives and is not modified.
    const ky_scene& flatten() const {
        flat_shapes_.clear(); flat_materials_.clear(); flat_lights_.clear(); flat_surfaces_.clear();
        for (auto& s : shape_list_) flat_shapes_.push_back(s->flatten());
        for (auto& m : material_list_) flat_materials_.push_back(m->flatten());
        for (auto& l : light_list_) flat_lights_.push_back(l->flatten(*this));
        for (const surface_t& s : surface_list_) {
            ky_surface fs{};
            fs.shape = shape_index(s.shape);
            fs.material = -1;
            for (size_t i = 0; i < material_list_.size(); ++i) if (material_list_[i].get() == s.material) fs.material = (int)i;
            if (fs.material < 0) throw std::runtime_error("ky::scene_t: a surface refers to a material that is not in material_list");
            fs.area_light = -1;
            for (size_t i = 0; i < light_list_.size(); ++i) if (light_list_[i].get() == s.area_light) fs.area_light = (int)i;
            flat_surfaces_.push_back(fs);
        }
        flat_ = ky_scene{};
        flat_.shapes = flat_shapes_.data(); flat_.shape_count = (int)flat_shapes_.size();
        flat_.materials = flat_materials_.data(); flat_.material_count = (int)flat_materials_.size();
        flat_.lights = flat_lights_.data(); flat_.light_count = (int)flat_lights_.size();
        flat_.surfaces = flat_surfaces_.data(); flat_.surface_count = (int)flat_surfaces_.size();
        flat_.environment_light = -1;
        for (size_t i = 0; i < light_list_.size(); ++i) if (light_list_[i].get() == environment_light_) flat_.environment_light = (int)i;
        flat_.camera = camera_->flatten();
        return flat_;
    }

    static scene_t create_cornell_box_scene(cornell_box_enum_t scene_enum, point2_t film_resolution);  // 3240-3432
    static scene_t create_mis_scene(point2_t film_resolution);                                         // 3434-3533

private:
    const_camera_sptr_t camera_;
    shape_list_t shape_list_;
    material_list_t material_list_;
    light_list_t light_list_;
    environment_light_t* environment_light_{};
    surface_list_t surface_list_;
    mutable std::vector<ky_shape> flat_shapes_;
    mutable std::vector<ky_material> flat_materials_;
    mutable std::vector<ky_light> flat_lights_;
    mutable std::vector<ky_surface> flat_surfaces_;
    mutable ky_scene flat_{};
};

inline void direction_light_t::preprocess(const scene_t& scene) { scene.world_bound().bounding_sphere(&world_center_, &world_radius_); }
inline void environment_light_t::preprocess(const scene_t& scene) { scene.world_bound().bounding_sphere(&world_center_, &world_radius_); }
inline ky_light area_light_t::flatten(const scene_t& scene) const {
    ky_light l{}; l.kind = KY_LIGHT_AREA; l.shape = scene.shape_index(shape_); store3(l.color, radiance_); return l;
}

// The SmallVCM-style box (3240-3432).  All literals are the reference's scene data.
inline scene_t scene_t::create_cornell_box_scene(cornell_box_enum_t scene_enum, point2_t film_resolution) {
    using enum_t = cornell_box_enum_t;
    const_camera_sptr_t camera = std::make_shared<camera_t>(
        point3_t{-0.0439815f, 4.12529f, 0.222539f}, vec3_t{0.00688625f, -0.998505f, -0.0542161f},
        vec3_t{3.73896e-4f, -0.0542148f, 0.998529f}, 80.f, film_resolution);

    if (enum_have(scene_enum, enum_t::large_mirror_sphere) && enum_have(scene_enum, enum_t::large_glass_sphere))
        throw std::runtime_error("cannot set both large balls");                                  // 3268-3271

    auto matte = [](float r, float g, float b) { return std::make_shared<matte_material_t>(color_t(r, g, b)); };
    material_sptr_t black = matte(0, 0, 0), white = matte(.8f, .8f, .8f);
    material_sptr_t red = matte(0.803922f, 0.152941f, 0.152941f), green = matte(0.156863f, 0.803922f, 0.172549f),
                    blue = matte(0.156863f, 0.172549f, 0.803922f);
    material_sptr_t glossy = std::make_shared<plastic_material_t>(color_t(.1f, .1f, .1f), color_t(.7f, .7f, .7f), 90.f);
    material_sptr_t mirror_mat = std::make_shared<mirror_material_t>(color_t(1, 1, 1));
    material_sptr_t glass_mat = std::make_shared<glass_material_t>(1.6f);
    material_list_t material_list{black, white, red, green, blue, glossy, mirror_mat, glass_mat};

    // box corners: bit0 = +x, bit1 = +z within the y- face (0..3), then the y+ face (4..7), as 3299-3309
    const float X0 = -1.27029f, X1 = 1.28975f, Y0 = -1.30455f, Y1 = 1.25549f, Z0 = -1.28002f, Z1 = 1.28002f;
    const vec3_t cb[8] = {{X0, Y0, Z0}, {X1, Y0, Z0}, {X1, Y0, Z1}, {X0, Y0, Z1}, {X0, Y1, Z0}, {X1, Y1, Z0}, {X1, Y1, Z1}, {X0, Y1, Z1}};
    auto rect = [](vec3_t a, vec3_t b, vec3_t c, vec3_t d) { return std::make_shared<rectangle_t>(a, b, c, d); };
    shape_sptr_t left = rect(cb[3], cb[0], cb[4], cb[7]), right = rect(cb[1], cb[2], cb[6], cb[5]), back = rect(cb[0], cb[3], cb[2], cb[1]),
                 bottom = rect(cb[0], cb[1], cb[5], cb[4]), top = rect(cb[2], cb[3], cb[7], cb[6]);

    const float large_radius = 0.8f, small_radius = 0.5f;
    const vec3_t large_center = (cb[0] + cb[4] + cb[5] + cb[1]) * (1.f / 4.f) + vec3_t(0, 0, large_radius);      // 3319
    const vec3_t left_wall_center = (cb[0] + cb[4]) * (1.f / 2.f) + vec3_t(0, 0, small_radius);                  // 3323
    const vec3_t right_wall_center = (cb[1] + cb[5]) * (1.f / 2.f) + vec3_t(0, 0, small_radius);                 // 3324
    const float length_x = right_wall_center.x - left_wall_center.x;
    const vec3_t left_center = left_wall_center + vec3_t(2.f * length_x / 7.f, 0, 0);                            // 3327
    const vec3_t right_center = right_wall_center - vec3_t(2.f * length_x / 7.f, 0, 0);                          // 3328
    shape_sptr_t large_ball = std::make_shared<sphere_t>(large_center, large_radius);
    shape_sptr_t left_ball = std::make_shared<sphere_t>(left_center, small_radius);
    shape_sptr_t right_ball = std::make_shared<sphere_t>(right_center, small_radius);

    const float L = 0.25f, LZ0 = 1.26002f, LZ1 = 1.28002f;  // light box under the ceiling, 3336-3346
    const vec3_t lb[8] = {{-L, -L, LZ0}, {L, -L, LZ0}, {L, -L, LZ1}, {-L, -L, LZ1}, {-L, L, LZ0}, {L, L, LZ0}, {L, L, LZ1}, {-L, L, LZ1}};
    shape_sptr_t left2 = rect(lb[3], lb[7], lb[4], lb[0]), right2 = rect(lb[1], lb[5], lb[6], lb[2]), front2 = rect(lb[4], lb[7], lb[6], lb[5]),
                 back2 = rect(lb[0], lb[1], lb[2], lb[3]), bottom2 = rect(lb[0], lb[4], lb[5], lb[1]);

    shape_list_t shape_list{left, right, back, bottom, top, large_ball, left_ball, right_ball, left2, right2, front2, back2, bottom2};

    light_list_t light_list;
    if (enum_have(scene_enum, enum_t::light_area))
        light_list.push_back(std::make_shared<area_light_t>(point3_t(), 1, color_t(25, 25, 25), bottom2.get()));
    if (enum_have(scene_enum, enum_t::light_direction))
        light_list.push_back(std::make_shared<direction_light_t>(point3_t(), 1, color_t(10, 4, 0), vec3_t(-1, -1.5f, -1)));
    if (enum_have(scene_enum, enum_t::light_point)) {
        const float I = 70 * k_inv_4pi;
        light_list.push_back(std::make_shared<point_light_t>(point3_t(0.0f, 0.5f, 1.0f), 1, color_t(I, I, I)));
    }
    environment_light_t* environment_light{};
    if (enum_have(scene_enum, enum_t::light_environment)) {
        auto light = std::make_shared<environment_light_t>(point3_t(), 1, color_t((float)(135. / 255), (float)(206. / 255), (float)(250. / 255)));
        light_list.push_back(light);
        environment_light = light.get();
    }

    surface_list_t surface_list{
        {left.get(), green.get(), nullptr}, {right.get(), red.get(), nullptr}, {top.get(), white.get(), nullptr},
        {bottom.get(), glossy.get(), nullptr}, {back.get(), blue.get(), nullptr}};
    if (enum_have(scene_enum, enum_t::large_mirror_sphere)) surface_list.push_back({large_ball.get(), mirror_mat.get(), nullptr});
    else if (enum_have(scene_enum, enum_t::large_glass_sphere)) surface_list.push_back({large_ball.get(), glass_mat.get(), nullptr});
    if (enum_have(scene_enum, enum_t::small_mirror_sphere)) surface_list.push_back({left_ball.get(), mirror_mat.get(), nullptr});
    if (enum_have(scene_enum, enum_t::small_glass_sphere)) surface_list.push_back({right_ball.get(), glass_mat.get(), nullptr});
    if (enum_have(scene_enum, enum_t::light_area)) {
        for (auto& s : {left2, right2, front2, back2}) surface_list.push_back({s.get(), white.get(), nullptr});
        surface_list.push_back({bottom2.get(), black.get(), (area_light_t*)light_list[0].get()});
    }
    return scene_t{camera, shape_list, material_list, light_list, surface_list, environment_light};
}

// Veach's MIS scene (3434-3533).  Lights 1 and 2 sample each other's spheres.
inline scene_t scene_t::create_mis_scene(point2_t film_resolution) {
    const_camera_sptr_t camera = std::make_shared<camera_t>(point3_t{0, 2, -15}, vec3_t{0, -4, 12.5f}, vec3_t{0, 1, 0}, 50.f, film_resolution);

    material_sptr_t black = std::make_shared<matte_material_t>(color_t());
    material_sptr_t gray = std::make_shared<matte_material_t>(color_t(.4f, .4f, .4f));
    material_sptr_t silver = std::make_shared<plastic_material_t>(color_t(0.07f, 0.09f, 0.13f), color_t(1, 1, 1), 5000.f);
    material_list_t material_list{black, gray, silver};

    auto rect = [](vec3_t a, vec3_t b, vec3_t c, vec3_t d) { return std::make_shared<rectangle_t>(a, b, c, d, true); };
    auto plank = [&](float y0, float z0, float y1, float z1) { return rect({4, y0, z0}, {4, y1, z1}, {-4, y1, z1}, {-4, y0, z0}); };
    shape_sptr_t bottom = rect({-10, -4.14615f, 10}, {-10, -4.14615f, -10}, {10, -4.14615f, -10}, {10, -4.14615f, 10});
    shape_sptr_t back = rect({-10, -10, 2}, {-10, 10, 2}, {10, 10, 2}, {10, -10, 2});
    shape_sptr_t plank0 = plank(-2.70651f, -0.25609f, -2.08375f, 0.526323f);
    shape_sptr_t plank1 = plank(-3.28825f, -1.36972f, -2.83856f, -0.476536f);
    shape_sptr_t plank2 = plank(-3.73096f, -2.70046f, -3.43378f, -1.74564f);
    shape_sptr_t plank3 = plank(-3.99615f, -4.0667f, -3.82069f, -3.08221f);
    auto ball = [](float x, float y, float z, float r) { return std::make_shared<sphere_t>(point3_t(x, y, z), r); };
    shape_sptr_t ball0 = ball(10, 10, -4, 0.5f), ball1 = ball(-3.75f, 0, 0, 0.03333f), ball2 = ball(-1.25f, 0, 0, 0.1f),
                 ball3 = ball(1.25f, 0, 0, 0.3f), ball4 = ball(3.75f, 0, 0, 0.9f);
    shape_list_t shape_list{bottom, back, plank0, plank1, plank2, plank3, ball0, ball1, ball2, ball3, ball4};

    auto area = [](float L, const shape_sptr_t& s) { return std::make_shared<area_light_t>(point3_t(), 1, color_t(L, L, L), s.get()); };
    auto light0 = area(800, ball0), light1 = area(901.803f, ball2), light2 = area(100, ball1), light3 = area(11.1111f, ball3),
         light4 = area(1.23457f, ball4);                                                          // 3497-3501
    light_list_t light_list{light0, light1, light2, light3, light4};

    surface_list_t surface_list{
        {bottom.get(), gray.get(), nullptr}, {back.get(), gray.get(), nullptr},
        {plank0.get(), silver.get(), nullptr}, {plank1.get(), silver.get(), nullptr}, {plank2.get(), silver.get(), nullptr}, {plank3.get(), silver.get(), nullptr},
        {ball0.get(), black.get(), light0.get()}, {ball1.get(), black.get(), light1.get()}, {ball2.get(), black.get(), light2.get()},
        {ball3.get(), black.get(), light3.get()}, {ball4.get(), black.get(), light4.get()}};      // 3514-3529
    return scene_t{camera, shape_list, material_list, light_list, surface_list};
}

// ---------------------------------------------------------------------------------------------
// sampler (ky.cpp:877-975): on this path only spp, kind and seed are read
// ---------------------------------------------------------------------------------------------
class sampler_t {
public:
    virtual ~sampler_t() = default;
    explicit sampler_t(int samples_per_pixel) : samples_per_pixel_(samples_per_pixel) {}
    virtual std::unique_ptr<sampler_t> clone() = 0;
    virtual int ge_samples_per_pixel() { return samples_per_pixel_; }                             // 890 (sic)
    virtual void set_samples_per_pixel(int spp) { samples_per_pixel_ = spp; }
    virtual ky_sampler_kind kind() const = 0;
    uint32_t seed() const { return seed_; }
    void set_seed(uint32_t seed) { seed_ = seed; }
protected:
    int samples_per_pixel_{};
    uint32_t seed_ = 1234;                                                                        // 833
};
class debug_sampler_t : public sampler_t {
public:
    using sampler_t::sampler_t;
    std::unique_ptr<sampler_t> clone() override { return std::make_unique<debug_sampler_t>(samples_per_pixel_); }
    ky_sampler_kind kind() const override { return KY_SAMPLER_DEBUG; }
};
class random_sampler_t : public sampler_t {
public:
    using sampler_t::sampler_t;
    std::unique_ptr<sampler_t> clone() override { auto s = std::make_unique<random_sampler_t>(samples_per_pixel_); s->set_seed(seed_); return s; }
    ky_sampler_kind kind() const override { return KY_SAMPLER_RANDOM; }
};

// ---------------------------------------------------------------------------------------------
// integrators (ky.cpp:3608-3654, 3679-3729, 4621-4639)
// ---------------------------------------------------------------------------------------------
enum class direct_sample_enum_t {
    idle, sample_single_light = 1, sample_all_light = 2, bsdf = 4, light = 8, bsdf_mis = 16, light_mis = 32,
    both_mis = bsdf_mis | light_mis, default_stragtgy = sample_all_light | both_mis
};
enum class integrator_enum_t {
    position, normal, basecolor, delta_bsdf, delta_light, direct_lighting_point, direct_lighting, stochastic_raytracing,
    simple_path_tracing_recursion, path_tracing_recursion, path_tracing_recursion_defered, path_tracing_iteration
};

class integrator_t {
public:
    virtual ~integrator_t() = default;

    // integrator_t::render(scene, sampler, film), ky.cpp:3689: adds clamp01(mean radiance) into the film's
    // current target.  Errors of the C ABI surface as exceptions, like the reference's LOG_ERROR (75-82).
    void render(scene_t* scene, sampler_t* original_sampler, film_t* film) {
        const ky_render_params p = params_for(original_sampler, film);
        // the reference spreads this loop over all cores (3696-3699); here the frame's tiles are spread over devices_
        const int rc = kyhip_render_multi(devices_.data(), (int)devices_.size(), &scene->flatten(), &p, film->target_origin(), film->row_stride_px());
        if (rc != KY_OK) throw std::runtime_error(std::string("kyhip_render_multi: ") + kyhip_last_error());
    }
    // integrator_t::debug_area / debug_pixel (ky.cpp:3733-3787), the reference's single-pixel replay: a red frame is ADDED around
    // [begin, end) (color_t{1.f} = (1, 0, 0) on the pixels of [begin - 1, end], 3739-3746), then every pixel of the area is cleared and
    // rendered again on its own (3756-3776): the caller's sampler, the samples summed one after the other in float, one clamp, add_color.
    // Here each pixel's samples are the device's replay of exactly the streams render() uses (kyhip_kat_li: per-sample radiance of pixel
    // (x, y)), so the area shows what render() computed there, to the rounding of a different summation order.  The reference writes the
    // frame without bounds checks (CHECK_DEBUG is compiled out in release, 108-115); here pixels outside the film are skipped.
    void debug_area(scene_t* scene, sampler_t* original_sampler, film_t* film, point2_t begin, point2_t end) {
        const vec2_t res = film->get_resolution();
        const int w = (int)res.x, h = (int)res.y;
        const int bx = (int)begin.x, by = (int)begin.y, ex = (int)end.x, ey = (int)end.y;
        for (int y = by - 1; y <= ey; ++y)
            for (int x = bx - 1; x <= ex; ++x)
                if (x >= 0 && y >= 0 && x < w && y < h) film->add_color(x, y, color_t{1.f, 0.f, 0.f});
        ky_render_params p = params_for(original_sampler, film);
        const int spp = p.samples_per_pixel;
        const float inv_spp = (float)(1. / spp);   // 3764: a double quotient narrowed by color_t::operator*(float_t)
        std::vector<float> li((size_t)spp * 3);
        for (int y = by; y < ey; ++y) {
            for (int x = bx; x < ex; ++x) {
                if (x < 0 || y < 0 || x >= w || y >= h) continue;
                film->clear_color(x, y);
                const int rc = kyhip_kat_li(devices_[0], &scene->flatten(), &p, x, y, 0, spp, li.data());
                if (rc != KY_OK) throw std::runtime_error(std::string("kyhip_kat_li: ") + kyhip_last_error());
                color_t L{};
                for (int s = 0; s < spp; ++s) L = L + color_t(li[3 * (size_t)s], li[3 * (size_t)s + 1], li[3 * (size_t)s + 2]) * inv_spp;
                film->add_color(x, y, color_t(clamp01_keep_nan(L.r), clamp01_keep_nan(L.g), clamp01_keep_nan(L.b)));
            }
        }
    }
    void debug_area(scene_t* scene, sampler_t* original_sampler, film_t* film, point2_t begin, float_t width, float_t height) {
        debug_area(scene, original_sampler, film, begin, point2_t(begin.x + width, begin.y + height));   // 3779-3782
    }
    void debug_pixel(scene_t* scene, sampler_t* original_sampler, film_t* film, point2_t pixel_position) {
        debug_area(scene, original_sampler, film, pixel_position, point2_t(pixel_position.x + 1, pixel_position.y + 1));   // 3784-3787
    }

    // duration of the integrator kernel of the last render() on the first device, milliseconds (hipEvents on the launch stream)
    float last_kernel_ms() const { return kyhip_kernel_ms(devices_[0]); }

    // The GPUs render() uses (default: the one device given to the constructor).  Tiles are interleaved over the list and
    // gathered on its first entry; the image does not depend on the list.  all_devices() = every GPU the process sees.
    void set_devices(std::vector<int> devices) {
        if (devices.empty()) throw std::runtime_error("integrator_t::set_devices: empty device list");
        devices_ = std::move(devices);
    }
    const std::vector<int>& devices() const { return devices_; }
    static std::vector<int> all_devices() {
        std::vector<int> d(std::max(1, kyhip_device_count()));
        for (size_t i = 0; i < d.size(); ++i) d[i] = (int)i;
        return d;
    }

protected:
    integrator_t(integrator_enum_t kind, int max_path_depth, direct_sample_enum_t direct_sample_enum, int device)
        : kind_(kind), max_path_depth_(max_path_depth), direct_sample_enum_(direct_sample_enum), devices_{device} {}
    // what the path reads from its sampler, its film and itself (SURVEY 8(b) "inputs read by the path")
    ky_render_params params_for(sampler_t* original_sampler, film_t* film) const {
        const vec2_t res = film->get_resolution();
        ky_render_params p{};
        p.integrator = (int)kind_;
        p.max_path_depth = max_path_depth_;
        p.direct_sample = (int)direct_sample_enum_;
        p.samples_per_pixel = original_sampler->ge_samples_per_pixel();
        p.sampler = original_sampler->kind();
        p.seed = original_sampler->seed();
        p.width = (int)res.x; p.height = (int)res.y;
        p.tile_w = 16; p.tile_h = 16; p.tile_first = 0; p.tile_step = 1;
        return p;
    }
    static float clamp01_keep_nan(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }   // std::clamp keeps a NaN (1545)
    integrator_enum_t kind_;
    int max_path_depth_;
    direct_sample_enum_t direct_sample_enum_;
    std::vector<int> devices_;
};

// debug_integrator_t(position | normal | basecolor), ky.cpp:4094-4123
class debug_integrator_t : public integrator_t {
public:
    explicit debug_integrator_t(integrator_enum_t integrator_enum, int device = 0)
        : integrator_t(integrator_enum, 0, direct_sample_enum_t::idle, device) {
        if ((int)integrator_enum > (int)integrator_enum_t::basecolor) throw std::runtime_error("debug_integrator_t: not a debug integrator enum");
    }
};
// direct_lighting_t, ky.cpp:4125-4155
class direct_lighting_t : public integrator_t {
public:
    explicit direct_lighting_t(direct_sample_enum_t direct_sample_enum, int device = 0)
        : integrator_t(integrator_enum_t::direct_lighting, 0, direct_sample_enum, device) {}
};
// path_integrator_t, ky.cpp:4172-4184
class path_integrator_t : public integrator_t {
protected:
    path_integrator_t(integrator_enum_t kind, int max_path_depth, direct_sample_enum_t direct_sample_enum, int device)
        : integrator_t(kind, max_path_depth, direct_sample_enum, device) {}
};
// path_tracing_iteration_t, ky.cpp:4523-4618 -- the hot path
class path_tracing_iteration_t : public path_integrator_t {
public:
    path_tracing_iteration_t(int max_path_depth, direct_sample_enum_t direct_sample_enum, int device = 0)
        : path_integrator_t(integrator_enum_t::path_tracing_iteration, max_path_depth, direct_sample_enum, device) {}
};

// the three recursive integrators (ky.cpp:4191-4514): the same estimators, run as modes of the device kernel
class simple_path_tracing_recursion_t : public path_integrator_t {
public:
    simple_path_tracing_recursion_t(int max_path_depth, direct_sample_enum_t direct_sample_enum, int device = 0)
        : path_integrator_t(integrator_enum_t::simple_path_tracing_recursion, max_path_depth, direct_sample_enum, device) {}
};
class path_tracing_recursion_t : public path_integrator_t {
public:
    path_tracing_recursion_t(int max_path_depth, direct_sample_enum_t direct_sample_enum, int device = 0)
        : path_integrator_t(integrator_enum_t::path_tracing_recursion, max_path_depth, direct_sample_enum, device) {}
};
enum class lighting_enum_t { emit = 1, direct = 2, indirect = 4, all_lighting = 7, diffuse = 8, specular = 16, all_scattering = 24, all = 31 };  // 3591-3603
class path_tracing_recursion_defered_t : public path_integrator_t {
public:
    // lighting_enum is stored and never read by the reference either (4412)
    path_tracing_recursion_defered_t(int max_path_depth, direct_sample_enum_t direct_sample_enum, lighting_enum_t = lighting_enum_t::all, int device = 0)
        : path_integrator_t(integrator_enum_t::path_tracing_recursion_defered, max_path_depth, direct_sample_enum, device) {}
};

// create_integrator, ky.cpp:4621-4639.  nullptr for enums the reference's switch does not handle (4638).
inline std::unique_ptr<integrator_t> create_integrator(integrator_enum_t integrator_enum, int depth, direct_sample_enum_t direct_sample_enum, int device = 0) {
    switch (integrator_enum) {
    case integrator_enum_t::direct_lighting: return std::make_unique<direct_lighting_t>(direct_sample_enum, device);
    case integrator_enum_t::simple_path_tracing_recursion: return std::make_unique<simple_path_tracing_recursion_t>(depth, direct_sample_enum, device);
    case integrator_enum_t::path_tracing_recursion: return std::make_unique<path_tracing_recursion_t>(depth, direct_sample_enum, device);
    case integrator_enum_t::path_tracing_recursion_defered:
        return std::make_unique<path_tracing_recursion_defered_t>(depth, direct_sample_enum, lighting_enum_t::all, device);
    case integrator_enum_t::path_tracing_iteration: return std::make_unique<path_tracing_iteration_t>(depth, direct_sample_enum, device);
    default: return nullptr;
    }
}

// The same factory for a list of GPUs (kyhip_render_multi): the frame's tiles are interleaved over `devices`.
inline std::unique_ptr<integrator_t> create_integrator(integrator_enum_t integrator_enum, int depth, direct_sample_enum_t direct_sample_enum, std::vector<int> devices) {
    if (devices.empty()) throw std::runtime_error("create_integrator: empty device list");
    auto integrator = create_integrator(integrator_enum, depth, direct_sample_enum, devices[0]);
    if (integrator) integrator->set_devices(std::move(devices));
    return integrator;
}

}  // namespace ky
