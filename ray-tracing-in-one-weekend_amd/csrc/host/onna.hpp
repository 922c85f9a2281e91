// onna.hpp -- host scene API for the features of the reference's Common Lisp tracer
// (hgiddens/ray-tracing-in-one-weekend, onna/): the constructors of scene.lisp, material.lisp,
// texture.lisp and camera.lisp under their own names (make-quad -> make_quad, ...), the scene
// builders of scenes.lisp, and render.  The host does what the Lisp constructors do once per object
// -- derived quad constants, bounding boxes, the bvh -- and flattens the graph into the tables of
// include/onna_hip.h; every ray is traced on the GPU behind that C ABI (libonna_hip.so).  There is no
// CPU renderer here.  Citations are file:line under onna/.
#pragma once

#include <cstdint>
#include <cstdio>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "onna_hip.h"

namespace onna {

struct vec3 { double x = 0, y = 0, z = 0; };
using point3 = vec3;                       // vec.lisp:86-92
struct colour { double r = 0, g = 0, b = 0; };   // colour.lisp:3-10
struct interval { double min, max; };      // vec.lisp:110-120 (the default is empty: +inf, -inf)
struct aabb { interval x, y, z; };         // aabb.lisp:12-36

// Lisp's (random x): the scene builders draw from one generator.  A double argument gives a double
// in [0, x); a single-float argument (the literals 1.0, 0.5 of scenes.lisp) a single in [0, x).
struct random_state {
    std::mt19937_64 engine;
    explicit random_state(uint64_t seed = 0) : engine(seed) {}
    double random(double x);
    float random(float x);
    int random(int n);
};

struct texture;
struct material;
struct object;
using texture_ptr = std::shared_ptr<texture const>;
using material_ptr = std::shared_ptr<material const>;
using object_ptr = std::shared_ptr<object const>;

// an image as image.lisp:10-28 holds it; rows top first, 8-bit RGB (a channel's value is (x / 256)^2)
struct image8 {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
};
image8 read_png(std::string const& path);                                     // image.lisp:10-28
void write_png(std::string const& path, image8 const& image);                 // image.lisp:30-40 (the bytes; gamma is render's business)

// ---- the flat tables of include/onna_hip.h, filled by object::flatten -------------------------
struct flat_scene {
    std::vector<on_object_desc> objects;
    std::vector<int32_t> children;
    std::vector<on_material_desc> materials;
    std::vector<on_texture_desc> textures;
    std::vector<on_image_desc> images;
    std::vector<uint8_t> pixels;
    std::vector<on_perlin_desc> perlins;
    int32_t world = -1, lights = -1;
    on_scene_desc desc() const;
    // shared materials and textures are entered once
    std::vector<std::pair<void const*, int32_t>> seen_materials, seen_textures, seen_objects;
};

// texture.lisp
struct texture {
    virtual ~texture() = default;
    virtual int32_t flatten(flat_scene& out) const = 0;
};
texture_ptr make_solid(colour c);                                             // texture.lisp:8-10: a colour is a texture
texture_ptr make_chequer(double scale, texture_ptr even, texture_ptr odd);   // texture.lisp:14-20
texture_ptr make_image_texture(image8 image);                                // texture.lisp:36-37
texture_ptr make_perlin(random_state& rs, double scale = 1.0);               // texture.lisp:50-69

// material.lisp
struct material {
    virtual ~material() = default;
    virtual int32_t flatten(flat_scene& out) const = 0;
};
material_ptr make_lambertian(texture_ptr t);                                  // material.lisp:49-50
material_ptr make_lambertian(colour c);
material_ptr make_metal(colour albedo, double fuzz = 0.0);                   // material.lisp:68-70
material_ptr make_dielectric(double refraction_index);                       // material.lisp:82-85
material_ptr make_diffuse_light(texture_ptr t);                              // material.lisp:110-111
material_ptr make_diffuse_light(colour c);
material_ptr make_isotropic(texture_ptr t);                                  // material.lisp:128-129

// scene.lisp
struct object {
    virtual ~object() = default;
    virtual aabb bounding_box() const = 0;                                    // scene.lisp:28
    virtual int32_t flatten(flat_scene& out) const = 0;
};
object_ptr make_sphere(point3 centre, double radius, material_ptr m = nullptr);             // scene.lisp:129-148 (:centre)
object_ptr make_sphere(point3 from, point3 to, double radius, material_ptr m);              // (:from :to)
object_ptr make_quad(point3 q, vec3 u, vec3 v, material_ptr m = nullptr);                   // scene.lisp:215-238
object_ptr make_list(std::vector<object_ptr> objects);                                      // a Lisp vector of objects (scene.lisp:92-115)
object_ptr make_box(point3 a, point3 b, material_ptr m);                                    // scene.lisp:292-315
object_ptr make_translate(object_ptr o, vec3 offset);                                       // scene.lisp:319-326
object_ptr make_rotate_y(object_ptr o, double angle_degrees);                               // scene.lisp:340-377
object_ptr make_constant_medium(object_ptr boundary, double density, texture_ptr t);        // scene.lisp:412-420
object_ptr make_bvh_node(std::vector<object_ptr> objects);                                  // scene.lisp:52-75

// camera.lisp:19-90
struct camera_options {
    double aspect_ratio = 16.0 / 9.0;
    uint32_t image_width = 400;
    uint32_t samples_per_pixel = 9;
    uint32_t max_depth = 10;
    colour background_colour{(double)0.7f, (double)0.8f, 1};
    double vertical_fov = 90;
    point3 look_from{0, 0, 0}, look_at{0, 0, -1};
    vec3 up{0, 1, 0};
    double defocus_angle = 0;
    double focus_distance = 10;
};
on_camera make_camera(camera_options const& o);

// scenes.lisp
struct scene { object_ptr world; object_ptr lights; camera_options camera; };
scene two_spheres();                       // scenes.lisp:3-7   (its spheres have no material: not renderable, kept for hit tests)
scene four_spheres();                      // :9-30
scene bouncing_spheres(random_state& rs);  // :32-94
scene chequered_spheres();                 // :96-107
scene earth(image8 image);                 // :109-114
scene perlin_spheres(random_state& rs);    // :116-121
scene quads();                             // :123-145
scene simple_light(random_state& rs);      // :147-155
scene cornell_box();                       // :157-182
scene cornell_smoke();                     // :184-206
scene final_scene(random_state& rs, image8 image);   // :208-294
scene scene_by_name(std::string const& name, random_state& rs, image8 const& image);
std::vector<std::string> scene_names();

flat_scene flatten(object_ptr const& world, object_ptr const& lights);

// a world resident on one GPU
class device_scene {
  public:
    device_scene(object_ptr const& world, object_ptr const& lights, int device = -1);
    ~device_scene();
    device_scene(device_scene const&) = delete;
    device_scene& operator=(device_scene const&) = delete;
    on_scene* handle() const { return scene_; }
    flat_scene const& tables() const { return flat_; }
  private:
    flat_scene flat_;
    on_scene* scene_ = nullptr;
};

// render (camera.lisp:165-181): the per-pixel sums, and the 8-bit image write-png would store
struct frame {
    uint32_t width = 0, height = 0, samples = 0;
    std::vector<double> sums;      // height x width x 3
    image8 resolve() const;        // blend-colours, gamma-2, colour-8bit (colour.lisp:12-37)
    rt_stats stats{};
};
frame render(on_camera const& camera, device_scene const& scene, uint64_t seed = 0);
frame render(on_camera const& camera, object_ptr const& world, object_ptr const& lights, uint64_t seed = 0, int device = -1);
// The same frame from several GPUs of the node (the reference's README.md:21-23 proposes the split for threads): GPU g
// of the list renders the samples [g S / n, (g + 1) S / n) of every pixel on a host thread of its own, and the per-GPU
// sums are added on the host in the order of the list.  The random streams are keyed by (pixel, sample): the paths are
// the ones a single GPU traces.  A GPU may be listed more than once.
frame render(on_camera const& camera, object_ptr const& world, object_ptr const& lights, uint64_t seed, std::vector<int> const& devices);

// hit-test of one ray (scene.lisp:22): flattens `o`, uploads it and asks the GPU
on_hit hit_test(rt_ray const& ray, object_ptr const& o, interval ray_interval, int device = -1);

}  // namespace onna
