// rtiow.hpp -- host-side scene API of the MI355X path tracer.
//
// The classes carry the names, constructor signatures and error behaviour of the
// reference's scene API (hgiddens/ray-tracing-in-one-weekend, c++/*.hpp) so that a
// program written against the reference builds its world the same way.  What stays on
// the host is what the reference also does once per scene: value-type arithmetic,
// bounding boxes, the BVH build and the camera basis.  Everything on the per-ray path
// -- hit(), scatter(), value(), get_ray(), the bounce loop -- runs on the GPU through
// the C ABI of rtiow_hip.h; there is no CPU implementation behind those calls.
//
// All names live in namespace rtiow.
#pragma once

#include <cstdint>
#include <memory>
#include <optional>
#include <random>
#include <span>
#include <vector>

#include "rtiow_hip.h"

namespace rtiow {

// ---- value types (vec3.hpp:3-24, ray.hpp:5-15, colour.hpp:3-17, aabb.hpp:6-20) --------
class vec3 final {
    double elements[3];
public:
    vec3(double x, double y, double z) : elements{x, y, z} {}
    double x() const { return elements[0]; }
    double y() const { return elements[1]; }
    double z() const { return elements[2]; }
    vec3 operator-() const { return {-x(), -y(), -z()}; }
    vec3 cross(vec3 that) const;
    double dot(vec3 that) const;
    double length() const;
    double squared_length() const;
    vec3 unit_vector() const;
    friend vec3 operator+(vec3 a, vec3 b);
    friend vec3 operator-(vec3 a, vec3 b);
    friend vec3 operator*(vec3 v, double d);
    friend vec3 operator*(double d, vec3 v);
    friend vec3 operator/(vec3 v, double d);
};

class ray final {
    vec3 a, b;
    double t0;
public:
    ray(vec3 origin, vec3 direction, double time) : a(origin), b(direction), t0(time) {}
    vec3 origin() const { return a; }
    vec3 direction() const { return b; }
    double time() const { return t0; }
    vec3 point_at_parameter(double t) const { return a + t * b; }
};

class colour final {
    double elements[3];
public:
    explicit colour(double white) : elements{white, white, white} {}
    colour(double red, double green, double blue) : elements{red, green, blue} {}
    double red() const { return elements[0]; }
    double green() const { return elements[1]; }
    double blue() const { return elements[2]; }
    colour gamma2() const;
    friend colour operator+(colour a, colour b);
    friend colour operator*(colour c, double d);
    friend colour operator*(double d, colour c);
};

inline double ffmin(double a, double b) { return a < b ? a : b; }
inline double ffmax(double a, double b) { return a > b ? a : b; }

class aabb final {
    vec3 _min, _max;
public:
    static aabb surrounding_box(aabb box0, aabb box1);
    aabb(vec3 a, vec3 b) : _min(a), _max(b) {}
    vec3 min() const { return _min; }
    vec3 max() const { return _max; }
    // the slab test runs on the GPU (rt_aabb_hit_batch)
    bool hit(ray const& r, double tmin, double tmax) const;
};

// ---- flattening ------------------------------------------------------------------------
class material;
class texture;
class object;

// Collects the linear tables handed to rt_scene_create.
class scene_flattener {
public:
    std::vector<rt_texture_desc> textures;
    std::vector<rt_material_desc> materials;
    std::vector<rt_prim_desc> prims;
    std::vector<rt_node_desc> nodes;
    std::vector<rt_item_desc> items;
    std::vector<material const*> material_of_prim;  // hit_record.material for each primitive

    rt_scene_desc desc() const;
    int32_t append(scene_flattener const& other, int32_t ref);
};

// The GPU used by the per-ray calls that bring no scene of their own -- aabb::hit,
// camera::get_ray, and the one-object scenes behind object::hit, material::scatter and
// texture::value.  -1, the default, is the calling thread's current HIP device (device 0
// unless the program chose another with hipSetDevice), so a process bound to one GPU of a
// node stays on it.
int probe_device();
void set_probe_device(int device);

// A world resident on one GPU (-1: the calling thread's current HIP device).
class device_scene final {
    rt_scene* handle = nullptr;
    scene_flattener flat;
public:
    explicit device_scene(object const& world, int device = -1);
    ~device_scene();
    device_scene(device_scene const&) = delete;
    device_scene& operator=(device_scene const&) = delete;
    rt_scene* get() const { return handle; }
    scene_flattener const& tables() const { return flat; }
    double flatten_seconds = 0, create_seconds = 0;   // how long the two halves of the constructor took
};

// ---- textures (texture.hpp:9-33) -------------------------------------------------------
class texture {
    mutable std::unique_ptr<device_scene> probe;  // one-primitive scene used by value()
public:
    texture();
    virtual ~texture();
    // evaluated on the GPU (rt_texture_batch)
    virtual colour value(double u, double v, vec3 const& p) const;
    virtual uint32_t flatten(scene_flattener& out) const = 0;
};

class constant_texture : public texture {
    colour c;
public:
    explicit constant_texture(colour c) : c(c) {}
    uint32_t flatten(scene_flattener& out) const override;
};

class chequer_texture : public texture {
    std::unique_ptr<texture const> even;
    std::unique_ptr<texture const> odd;
public:
    chequer_texture(std::unique_ptr<texture const> even, std::unique_ptr<texture const> odd);
    uint32_t flatten(scene_flattener& out) const override;
};

class perlin_texture : public texture {
public:
    uint32_t flatten(scene_flattener& out) const override;
};

// perlin.hpp:8-14.  The tables equal the ones the reference obtains from glibc's
// drand48() in its default state (perlin.cpp:8-31,50-53); noise() runs on the GPU.
class perlin {
public:
    static std::vector<double> const ranfloat;
    static std::vector<uint8_t> const perm_x, perm_y, perm_z;
    double noise(vec3 const& p) const;
};

// ---- materials (material.hpp:11-53) ----------------------------------------------------
struct hit_record;

struct scatter_record {
    vec3 attenuation;
    ray scattered;
};

class material {
    mutable std::unique_ptr<device_scene> probe;
protected:
    std::mt19937& mt;
    explicit material(std::mt19937& mt);
public:
    virtual ~material();
    // Runs on the GPU (rt_scatter_batch).  The uniforms are drawn from the shared
    // generator exactly as the reference would draw them, and the generator advances
    // by exactly the number consumed.
    virtual std::optional<scatter_record> scatter(ray const& ray, hit_record const& hit) const;
    virtual uint32_t flatten(scene_flattener& out) const = 0;
};

class lambertian final : public material {
    std::unique_ptr<texture const> t;
public:
    lambertian(std::mt19937& mt, std::unique_ptr<texture const> texture);
    uint32_t flatten(scene_flattener& out) const override;
};

class metal final : public material {
    vec3 albedo;
    double fuzziness;
public:
    metal(std::mt19937& mt, vec3 albedo, double fuzziness);
    uint32_t flatten(scene_flattener& out) const override;
};

class dielectric final : public material {
    double refractive_index;
public:
    dielectric(std::mt19937& mt, double refractive_index);
    uint32_t flatten(scene_flattener& out) const override;
};

// ---- objects (object.hpp:10-22 and the five implementations) ---------------------------
struct hit_record {
    double t;
    vec3 p;
    vec3 normal;
    class material const* material;
};

class object {
    mutable std::unique_ptr<device_scene> self_scene;  // built on the first hit() call
public:
    object();
    virtual ~object();
    // Runs on the GPU (rt_trace_batch) over a device copy of this object.
    virtual std::optional<hit_record> hit(ray const& r, double t_min, double t_max) const;
    virtual std::optional<aabb> bounding_box(double t0, double t1) const = 0;
    // Appends this object to the tables as a bvh child: returns a node index (>= 0) or
    // ~primitive index.  `parent_box` is the box of the bvh_node whose child it is.
    virtual int32_t flatten_child(scene_flattener& out, aabb const& parent_box) const = 0;
    // The objects a list scan visits in this object's place, in order: the object itself, or --
    // for a list -- its members (a list in a list scans like its members written out in place).
    virtual void scan_members(std::vector<object const*>& out) const;
    // Appends this object to the world's item list.
    virtual void flatten_items(scene_flattener& out) const;
};

class sphere final : public object {
    vec3 centre;
    double radius;
    std::unique_ptr<material const> mat;
public:
    sphere(vec3 centre, double radius, std::unique_ptr<material const> mat);
    std::optional<aabb> bounding_box(double t0, double t1) const override;
    int32_t flatten_child(scene_flattener& out, aabb const& parent_box) const override;
};

class moving_sphere final : public object {
    vec3 centre0, centre1;
    double time0, time1;
    double radius;
    std::unique_ptr<material const> mat;
    vec3 centre(double time) const;
public:
    moving_sphere(vec3 centre0, vec3 centre1, double time0, double time1, double radius, std::unique_ptr<material const> mat);
    std::optional<aabb> bounding_box(double t0, double t1) const override;
    int32_t flatten_child(scene_flattener& out, aabb const& parent_box) const override;
};

class object_list final : public object {
    std::vector<std::unique_ptr<object const>> objects;
public:
    explicit object_list(std::unique_ptr<object const> object);
    explicit object_list(std::vector<std::unique_ptr<object const>> objects);
    std::optional<aabb> bounding_box(double t0, double t1) const override;
    int32_t flatten_child(scene_flattener& out, aabb const& parent_box) const override;
    void flatten_items(scene_flattener& out) const override;
    void scan_members(std::vector<object const*>& out) const override;
};

class bvh_node final : public object {
    static std::uniform_int_distribution<int> dist;
    // (the reference holds shared_ptrs because a node over one object points at it twice,
    // bvh_node.hpp:14-15; here the node owns each child once and points at it once or twice)
    std::unique_ptr<object const> own_left, own_right;
    object const* left = nullptr;
    object const* right = nullptr;
    std::optional<aabb> box;
    size_t n_objects = 0;   // objects under this node
    static constexpr size_t kParallelFlatten = size_t(1) << 16;   // a node over this many objects flattens its halves on two threads
    // the axis of every node, drawn before the build (see rtiow.cpp)
    struct build_plan;
    static build_plan draw_axes(std::mt19937& mt, size_t n, double t0, double t1);
    static build_plan prepare(std::mt19937& mt, std::span<std::unique_ptr<object const>> objects, double t0, double t1);
    static bool arrange(build_plan const& plan, void* keys, size_t n, size_t index, int depth);
    bvh_node(build_plan const& plan, std::span<std::unique_ptr<object const>> objects, size_t index, int depth);
public:
    // Same build as the reference (bvh_node.cpp:8-50): random axis, std::ranges::sort on
    // the box minimum, median split; takes ownership of the objects in the span.  Throws
    // a string literal when an object has no bounding box.
    bvh_node(std::mt19937& mt, std::span<std::unique_ptr<object const>> objects, double t0, double t1);
    std::optional<aabb> bounding_box(double t0, double t1) const override;
    int32_t flatten_child(scene_flattener& out, aabb const& parent_box) const override;
};

// ---- camera, supersampler (camera.hpp:7-28, supersampler.hpp:5-14) ---------------------
class camera final {
    static std::uniform_real_distribution<double> dist;
    vec3 origin, w, u, v, lower_left_corner, horizontal, vertical;
    double lens_radius;
    double t0, t1;
    std::mt19937& mt;
public:
    // vfov = vertical field of view in degrees
    camera(std::mt19937& mt, vec3 from, vec3 at, vec3 up, double vfov, double aspect, double aperture, double focus_dist, double t0, double t1);
    // Runs on the GPU (rt_camera_batch); consumes the shared generator as the reference does.
    ray get_ray(double s, double t) const;
    rt_camera describe() const;
};

class supersampler final {
    int count;
    double r, g, b;
public:
    supersampler();
    void add_sample(colour colour);
    colour value() const;
};

// ---- the frame (main.cpp:107-151) -------------------------------------------------------
struct render_options {
    int nx = 400, ny = 200, ns = 10;   // main.cpp:108
    int max_depth = 50;                // main.cpp:34
    double t_min = 0.001;              // main.cpp:23
    uint64_t seed = 0;
    int device = -1;                   // -1: the calling thread's current HIP device
    // Several GPUs: the frame's ns samples are split between them by sample index (README.md:21-23), each
    // renders its share of every pixel, and the sums are merged with one RCCL reduction.  Empty: `device`.
    std::vector<int> devices;
};

struct frame {
    int nx = 0, ny = 0, ns = 0;
    std::vector<float> sums;   // nx*ny*3 per-pixel sums over the samples, top row first
    rt_stats stats{};
    colour pixel(int i, int row) const;       // supersampler::value().gamma2() of that pixel
    void write_ppm(std::FILE* out) const;     // the P3 text of main.cpp:145-150
    void write_ppm_binary(std::FILE* out) const;   // the same values as binary P6
    void write_png(std::FILE* out) const;          // the same values as an 8-bit RGB PNG
};

// The pixel/sample loop of main.cpp:126-141 on one GPU, or on opt.devices.
frame render(object const& world, camera const& cam, render_options const& opt);
frame render(device_scene const& scene, camera const& cam, render_options const& opt);
// ... on the GPUs that hold `scenes` (the same world uploaded to each: upload()), sample ranges in list order
frame render(std::vector<std::unique_ptr<device_scene>> const& scenes, camera const& cam, render_options const& opt);
// the world on each of `devices`, uploaded by one host thread per device
std::vector<std::unique_ptr<device_scene>> upload(object const& world, std::vector<int> const& devices);

// Scene builders of main.cpp:52-99.  The argument evaluation order that the reference
// leaves to the compiler (main.cpp:89) is fixed here: albedo red, green, blue, then
// fuzziness.
object_list two_perlin_spheres(std::mt19937& mt);
bvh_node build_world(std::mt19937& mt);

}  // namespace rtiow
