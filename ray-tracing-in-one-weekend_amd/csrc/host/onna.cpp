// onna.cpp -- host scene API of the Lisp tracer's features (see onna.hpp).  Citations are file:line
// under onna/ of hgiddens/ray-tracing-in-one-weekend.
//
// Arithmetic follows the Lisp expressions' operand order (vec3+ sums from zero, n-ary + and *
// associate to the left) and their numeric contagion: the decimal literals of the Lisp files are
// single-floats unless written with a d exponent, and widen to double exactly where they meet one.
#include "onna.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace onna {

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double sf(float x) { return (double)x; }   // a single-float literal where it meets a double

// vec.lisp:27-60
vec3 add2(vec3 a, vec3 b) { return {(0.0 + a.x) + b.x, (0.0 + a.y) + b.y, (0.0 + a.z) + b.z}; }
vec3 add3(vec3 a, vec3 b, vec3 c) { return {((0.0 + a.x) + b.x) + c.x, ((0.0 + a.y) + b.y) + c.y, ((0.0 + a.z) + b.z) + c.z}; }
vec3 neg(vec3 a) { return {-a.x, -a.y, -a.z}; }
vec3 sub(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
vec3 scaled(vec3 a, double d) { return {a.x * d, a.y * d, a.z * d}; }
double dot(vec3 u, vec3 v) { return (u.x * v.x + u.y * v.y) + u.z * v.z; }
vec3 cross(vec3 u, vec3 v) { return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }
double length(vec3 v) { return std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z); }
vec3 unit(vec3 v) { return scaled(v, 1.0 / length(v)); }
// vec.lisp:94-99
point3 padd1(point3 p, vec3 a) { return {p.x + (0.0 + a.x), p.y + (0.0 + a.y), p.z + (0.0 + a.z)}; }
point3 padd2(point3 p, vec3 a, vec3 b) { vec3 const s = add2(a, b); return {p.x + s.x, p.y + s.y, p.z + s.z}; }

// vec.lisp:110-153
interval empty_interval() { return {HUGE_VAL, -HUGE_VAL}; }
interval combine(interval a, interval b) { return {std::min(a.min, b.min), std::max(a.max, b.max)}; }
double size(interval i) { return i.max - i.min; }
// aabb.lisp:3-10
interval pad_to_minimum(interval i) {
    double const delta = sf(0.0001f);
    if (size(i) < delta) { double const padding = delta / 2.0; return {i.min - padding, i.max + padding}; }
    return i;
}
// aabb.lisp:19-33
aabb empty_aabb() { return {empty_interval(), empty_interval(), empty_interval()}; }
aabb aabb_from_points(point3 a, point3 b) {
    return {pad_to_minimum({std::min(a.x, b.x), std::max(a.x, b.x)}), pad_to_minimum({std::min(a.y, b.y), std::max(a.y, b.y)}),
            pad_to_minimum({std::min(a.z, b.z), std::max(a.z, b.z)})};
}
aabb aabb_from_aabbs(aabb const& a, aabb const& b) { return {combine(a.x, b.x), combine(a.y, b.y), combine(a.z, b.z)}; }

void put3(double* out, vec3 v) { out[0] = v.x; out[1] = v.y; out[2] = v.z; }
void put_box(double* out, aabb const& b) { out[0] = b.x.min; out[1] = b.x.max; out[2] = b.y.min; out[3] = b.y.max; out[4] = b.z.min; out[5] = b.z.max; }

template <class T> int32_t find_seen(std::vector<std::pair<void const*, int32_t>> const& seen, T const* p) {
    for (auto const& e : seen)
        if (e.first == p) return e.second;
    return -1;
}

// ---- textures ------------------------------------------------------------------------------------
struct solid : texture {
    colour c;
    explicit solid(colour c_) : c(c_) {}
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_textures, this); k >= 0) return k;
        on_texture_desc d{};
        d.kind = ON_TEX_SOLID; d.even = d.odd = d.table = -1;
        d.colour[0] = c.r; d.colour[1] = c.g; d.colour[2] = c.b;
        out.textures.push_back(d);
        out.seen_textures.push_back({this, (int32_t)out.textures.size() - 1});
        return (int32_t)out.textures.size() - 1;
    }
};

struct chequer : texture {
    double inv_scale;
    texture_ptr even, odd;
    chequer(double scale, texture_ptr e, texture_ptr o) : inv_scale(1.0 / scale), even(std::move(e)), odd(std::move(o)) {}
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_textures, this); k >= 0) return k;
        on_texture_desc d{};
        d.kind = ON_TEX_CHEQUER; d.table = -1;
        d.even = even->flatten(out); d.odd = odd->flatten(out);
        d.scale = inv_scale;
        out.textures.push_back(d);
        out.seen_textures.push_back({this, (int32_t)out.textures.size() - 1});
        return (int32_t)out.textures.size() - 1;
    }
};

struct image_texture : texture {
    image8 image;
    explicit image_texture(image8 im) : image(std::move(im)) {}
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_textures, this); k >= 0) return k;
        on_image_desc im{};
        im.width = image.width; im.height = image.height; im.offset = out.pixels.size();
        out.pixels.insert(out.pixels.end(), image.rgb.begin(), image.rgb.end());
        out.images.push_back(im);
        on_texture_desc d{};
        d.kind = ON_TEX_IMAGE; d.even = d.odd = -1; d.table = (int32_t)out.images.size() - 1;
        out.textures.push_back(d);
        out.seen_textures.push_back({this, (int32_t)out.textures.size() - 1});
        return (int32_t)out.textures.size() - 1;
    }
};

struct perlin : texture {
    on_perlin_desc tables;
    double scale;
    // texture.lisp:50-69: 256 random unit vectors (vec.lisp:66-72), three shuffles of 0..255
    perlin(random_state& rs, double scale_) : scale(scale_) {
        for (auto& v : tables.vec) {
            for (;;) {
                vec3 const p{-1.0 + rs.random(2.0), -1.0 + rs.random(2.0), -1.0 + rs.random(2.0)};
                double const l2 = (p.x * p.x + p.y * p.y) + p.z * p.z;
                if (1e-160 < l2 && l2 <= 1.0) { put3(v, scaled(p, 1.0 / std::sqrt(l2))); break; }
            }
        }
        for (auto& perm : tables.perm) {
            for (int i = 0; i < 256; ++i) perm[i] = i;
            for (int i = 255; i > 0; --i) std::swap(perm[i], perm[rs.random(i + 1)]);   // alexandria:shuffle
        }
    }
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_textures, this); k >= 0) return k;
        out.perlins.push_back(tables);
        on_texture_desc d{};
        d.kind = ON_TEX_PERLIN; d.even = d.odd = -1; d.table = (int32_t)out.perlins.size() - 1;
        d.scale = scale;
        out.textures.push_back(d);
        out.seen_textures.push_back({this, (int32_t)out.textures.size() - 1});
        return (int32_t)out.textures.size() - 1;
    }
};

// ---- materials -----------------------------------------------------------------------------------
struct plain_material : material {
    uint32_t kind;
    texture_ptr tex;
    colour albedo;
    double fuzz = 0, refraction_index = 0;
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_materials, this); k >= 0) return k;
        on_material_desc d{};
        d.kind = kind;
        d.texture = tex ? tex->flatten(out) : -1;
        d.albedo[0] = albedo.r; d.albedo[1] = albedo.g; d.albedo[2] = albedo.b;
        d.fuzz = fuzz; d.refraction_index = refraction_index;
        out.materials.push_back(d);
        out.seen_materials.push_back({this, (int32_t)out.materials.size() - 1});
        return (int32_t)out.materials.size() - 1;
    }
};

material_ptr textured(uint32_t kind, texture_ptr t) {
    if (!t) throw std::invalid_argument("onna: material without a texture");
    auto m = std::make_shared<plain_material>();
    m->kind = kind; m->tex = std::move(t);
    return m;
}

// ---- objects -------------------------------------------------------------------------------------
struct base_object : object {
    // enters the record once; shared objects (a boundary that is also in the world) keep one number
    template <class F> int32_t enter(flat_scene& out, F&& fill) const {
        if (int32_t k = find_seen(out.seen_objects, this); k >= 0) return k;
        on_object_desc d{};
        d.material = -1; d.a = d.b = -1;
        fill(d);
        put_box(d.box, bounding_box());
        out.objects.push_back(d);
        out.seen_objects.push_back({this, (int32_t)out.objects.size() - 1});
        return (int32_t)out.objects.size() - 1;
    }
};

struct sphere : base_object {
    point3 origin; vec3 direction; double radius; material_ptr mat; aabb box;
    sphere(point3 o, vec3 d, double r, material_ptr m) : origin(o), direction(d), radius(r), mat(std::move(m)) {
        // scene.lisp:135-143
        vec3 const rv{radius, radius, radius};
        point3 const c0 = padd1(origin, scaled(direction, 0.0)), c1 = padd1(origin, scaled(direction, 1.0));
        box = aabb_from_aabbs(aabb_from_points(padd1(c0, neg(rv)), padd1(c0, rv)), aabb_from_points(padd1(c1, neg(rv)), padd1(c1, rv)));
    }
    aabb bounding_box() const override { return box; }
    int32_t flatten(flat_scene& out) const override {
        return enter(out, [&](on_object_desc& d) {
            d.kind = ON_OBJ_SPHERE;
            d.material = mat ? mat->flatten(out) : -1;
            put3(d.v, origin); put3(d.v + 3, direction); d.v[6] = radius;
        });
    }
};

struct quad : base_object {
    point3 q; vec3 u, v, w, normal; double d, area; material_ptr mat; aabb box;
    quad(point3 q_, vec3 u_, vec3 v_, material_ptr m) : q(q_), u(u_), v(v_), mat(std::move(m)) {
        // scene.lisp:219-228
        vec3 const n = cross(u, v);
        w = scaled(n, 1.0 / dot(n, n));
        box = aabb_from_aabbs(aabb_from_points(q, padd2(q, u, v)), aabb_from_points(padd1(q, u), padd1(q, v)));
        normal = unit(n);
        d = dot(normal, q);
        area = length(n);
    }
    aabb bounding_box() const override { return box; }
    int32_t flatten(flat_scene& out) const override {
        return enter(out, [&](on_object_desc& o) {
            o.kind = ON_OBJ_QUAD;
            o.material = mat ? mat->flatten(out) : -1;
            put3(o.v, q); put3(o.v + 3, u); put3(o.v + 6, v); put3(o.v + 9, w); put3(o.v + 12, normal);
            o.v[15] = d; o.v[16] = area;
        });
    }
};

struct list : base_object {
    std::vector<object_ptr> members;
    explicit list(std::vector<object_ptr> m) : members(std::move(m)) {}
    aabb bounding_box() const override {   // scene.lisp:101-104
        aabb b = empty_aabb();
        for (auto const& m : members) b = aabb_from_aabbs(b, m->bounding_box());
        return b;
    }
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_objects, this); k >= 0) return k;
        std::vector<int32_t> ids;
        for (auto const& m : members) ids.push_back(m->flatten(out));
        return enter(out, [&](on_object_desc& d) {
            d.kind = ON_OBJ_LIST;
            d.a = (int32_t)out.children.size(); d.b = (int32_t)ids.size();
            out.children.insert(out.children.end(), ids.begin(), ids.end());
        });
    }
};

struct translate : base_object {
    object_ptr inner; vec3 offset; aabb box;
    translate(object_ptr o, vec3 off) : inner(std::move(o)), offset(off) {
        aabb const b = inner->bounding_box();   // aabb.lisp:70-74
        box = {{b.x.min + off.x, b.x.max + off.x}, {b.y.min + off.y, b.y.max + off.y}, {b.z.min + off.z, b.z.max + off.z}};
    }
    aabb bounding_box() const override { return box; }
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_objects, this); k >= 0) return k;
        int32_t const child = inner->flatten(out);
        return enter(out, [&](on_object_desc& d) { d.kind = ON_OBJ_TRANSLATE; d.a = child; put3(d.v, offset); });
    }
};

struct rotate_y : base_object {
    object_ptr inner; double sin_theta, cos_theta; aabb box;
    rotate_y(object_ptr o, double angle) : inner(std::move(o)) {
        // scene.lisp:344-373
        double const radians = (angle * kPi) / 180.0;
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        aabb const b = inner->bounding_box();
        point3 lo{HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi{-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    double const x = b.x.max * i + b.x.min * (1 - i), y = b.y.max * j + b.y.min * (1 - j), z = b.z.max * k + b.z.min * (1 - k);
                    double const nx = cos_theta * x + sin_theta * z, nz = (-sin_theta) * x + cos_theta * z;
                    lo = {std::min(lo.x, nx), std::min(lo.y, y), std::min(lo.z, nz)};
                    hi = {std::max(hi.x, nx), std::max(hi.y, y), std::max(hi.z, nz)};
                }
        box = aabb_from_points(lo, hi);
    }
    aabb bounding_box() const override { return box; }
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_objects, this); k >= 0) return k;
        int32_t const child = inner->flatten(out);
        return enter(out, [&](on_object_desc& d) { d.kind = ON_OBJ_ROTATE_Y; d.a = child; d.v[0] = sin_theta; d.v[1] = cos_theta; });
    }
};

struct constant_medium : base_object {
    object_ptr boundary; double negative_inverse_density; material_ptr phase;
    constant_medium(object_ptr b, double density, texture_ptr t)
        : boundary(std::move(b)), negative_inverse_density(-1.0 / density), phase(make_isotropic(std::move(t))) {}
    aabb bounding_box() const override { return boundary->bounding_box(); }   // scene.lisp:454-455
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_objects, this); k >= 0) return k;
        int32_t const child = boundary->flatten(out);
        return enter(out, [&](on_object_desc& d) {
            d.kind = ON_OBJ_MEDIUM; d.a = child; d.material = phase->flatten(out); d.v[0] = negative_inverse_density;
        });
    }
};

struct bvh_node : base_object {
    object_ptr left, right; aabb box;
    // scene.lisp:52-75.  The Lisp `sort` is not stable and its order among equal keys is SBCL's
    // business; this one is stable.
    explicit bvh_node(std::vector<object_ptr> objects) {
        if (objects.empty()) throw std::invalid_argument("onna: make_bvh_node over no objects");
        box = empty_aabb();
        for (auto const& o : objects) box = aabb_from_aabbs(box, o->bounding_box());
        size_t const span = objects.size();
        if (span == 1) {
            left = right = objects[0];
        } else if (span == 2) {
            left = objects[0]; right = objects[1];
        } else {
            // aabb.lisp:61-68
            double const x = size(box.x), y = size(box.y), z = size(box.z);
            int const axis = x > y ? (x > z ? 0 : 2) : (y > z ? 1 : 2);
            auto key = [axis](object_ptr const& o) { aabb const b = o->bounding_box(); return axis == 0 ? b.x.min : (axis == 1 ? b.y.min : b.z.min); };
            std::vector<std::pair<double, object_ptr>> keyed;
            keyed.reserve(span);
            for (auto& o : objects) keyed.emplace_back(key(o), o);
            std::stable_sort(keyed.begin(), keyed.end(), [](auto const& a, auto const& b) { return a.first < b.first; });
            size_t const mid = span / 2;
            std::vector<object_ptr> lo, hi;
            for (size_t i = 0; i < span; ++i) (i < mid ? lo : hi).push_back(keyed[i].second);
            left = std::make_shared<bvh_node>(std::move(lo));
            right = std::make_shared<bvh_node>(std::move(hi));
        }
    }
    aabb bounding_box() const override { return box; }
    int32_t flatten(flat_scene& out) const override {
        if (int32_t k = find_seen(out.seen_objects, this); k >= 0) return k;
        int32_t const l = left->flatten(out), r = right == left ? l : right->flatten(out);
        return enter(out, [&](on_object_desc& d) { d.kind = ON_OBJ_BVH; d.a = l; d.b = r; });
    }
};

}  // namespace

// ---- random ----------------------------------------------------------------------------------------
double random_state::random(double x) { return (double)(engine() >> 11) * 0x1p-53 * x; }
float random_state::random(float x) { return (float)(engine() >> 40) * 0x1p-24f * x; }
int random_state::random(int n) { return n > 0 ? (int)(engine() % (uint64_t)n) : 0; }

// ---- constructors ----------------------------------------------------------------------------------
texture_ptr make_solid(colour c) { return std::make_shared<solid>(c); }
texture_ptr make_chequer(double scale, texture_ptr even, texture_ptr odd) {
    if (!even || !odd) throw std::invalid_argument("onna: make_chequer needs both textures");
    return std::make_shared<chequer>(scale, std::move(even), std::move(odd));
}
texture_ptr make_image_texture(image8 image) {
    if (image.width == 0 || image.height == 0 || image.rgb.size() != (size_t)image.width * image.height * 3) throw std::invalid_argument("onna: make_image_texture: empty or ragged image");
    return std::make_shared<image_texture>(std::move(image));
}
texture_ptr make_perlin(random_state& rs, double scale) { return std::make_shared<perlin>(rs, scale); }

material_ptr make_lambertian(texture_ptr t) { return textured(ON_MAT_LAMBERTIAN, std::move(t)); }
material_ptr make_lambertian(colour c) { return make_lambertian(make_solid(c)); }
material_ptr make_metal(colour albedo, double fuzz) {
    auto m = std::make_shared<plain_material>();
    m->kind = ON_MAT_METAL; m->albedo = albedo; m->fuzz = fuzz;
    return m;
}
material_ptr make_dielectric(double refraction_index) {
    auto m = std::make_shared<plain_material>();
    m->kind = ON_MAT_DIELECTRIC; m->refraction_index = refraction_index;
    return m;
}
material_ptr make_diffuse_light(texture_ptr t) { return textured(ON_MAT_DIFFUSE_LIGHT, std::move(t)); }
material_ptr make_diffuse_light(colour c) { return make_diffuse_light(make_solid(c)); }
material_ptr make_isotropic(texture_ptr t) { return textured(ON_MAT_ISOTROPIC, std::move(t)); }

object_ptr make_sphere(point3 centre, double radius, material_ptr m) { return std::make_shared<sphere>(centre, vec3{0, 0, 0}, radius, std::move(m)); }
object_ptr make_sphere(point3 from, point3 to, double radius, material_ptr m) { return std::make_shared<sphere>(from, sub(to, from), radius, std::move(m)); }   // scene.lisp:119-127
object_ptr make_quad(point3 q, vec3 u, vec3 v, material_ptr m) { return std::make_shared<quad>(q, u, v, std::move(m)); }
object_ptr make_list(std::vector<object_ptr> objects) {
    for (auto const& o : objects)
        if (!o) throw std::invalid_argument("onna: make_list: null object");
    return std::make_shared<list>(std::move(objects));
}
object_ptr make_box(point3 a, point3 b, material_ptr m) {
    // scene.lisp:292-315
    point3 const lo{std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}, hi{std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)};
    vec3 const dx{hi.x - lo.x, 0, 0}, dy{0, hi.y - lo.y, 0}, dz{0, 0, hi.z - lo.z};
    return make_list({
        make_quad({lo.x, lo.y, hi.z}, dx, dy, m),        // front
        make_quad({hi.x, lo.y, hi.z}, neg(dz), dy, m),   // right
        make_quad({hi.x, lo.y, lo.z}, neg(dx), dy, m),   // back
        make_quad({lo.x, lo.y, lo.z}, dz, dy, m),        // left
        make_quad({lo.x, hi.y, hi.z}, dx, neg(dz), m),   // top
        make_quad({lo.x, lo.y, lo.z}, dx, dz, m),        // bottom
    });
}
object_ptr make_translate(object_ptr o, vec3 offset) {
    if (!o) throw std::invalid_argument("onna: make_translate: null object");
    return std::make_shared<translate>(std::move(o), offset);
}
object_ptr make_rotate_y(object_ptr o, double angle_degrees) {
    if (!o) throw std::invalid_argument("onna: make_rotate_y: null object");
    return std::make_shared<rotate_y>(std::move(o), angle_degrees);
}
object_ptr make_constant_medium(object_ptr boundary, double density, texture_ptr t) {
    if (!boundary || !t) throw std::invalid_argument("onna: make_constant_medium needs a boundary and a texture");
    return std::make_shared<constant_medium>(std::move(boundary), density, std::move(t));
}
object_ptr make_bvh_node(std::vector<object_ptr> objects) { return std::make_shared<bvh_node>(std::move(objects)); }

// ---- camera (camera.lisp:19-90) --------------------------------------------------------------------
on_camera make_camera(camera_options const& o) {
    on_camera c{};
    uint32_t const root = (uint32_t)std::floor(std::sqrt((double)o.samples_per_pixel));
    if (root == 0 || o.image_width == 0) throw std::invalid_argument("onna: make_camera: no samples or no pixels");
    double const height = std::nearbyint((double)o.image_width / o.aspect_ratio);   // (round w aspect): to even on ties
    c.image_width = o.image_width;
    c.image_height = (uint32_t)std::max(height, 1.0);
    c.sqrt_spp = root;
    c.recip_sqrt_spp = 1.0 / (double)root;
    c.max_depth = o.max_depth;
    double const theta = (o.vertical_fov * kPi) / 180;
    double const h = std::tan(theta / 2.0);
    double const viewport_height = (2.0 * h) * o.focus_distance;
    double const viewport_width = viewport_height * o.aspect_ratio;
    vec3 const w = unit(sub(o.look_from, o.look_at)), u = unit(cross(o.up, w)), v = cross(w, u);
    vec3 const viewport_u = scaled(u, viewport_width), viewport_v = scaled(v, -viewport_height);
    vec3 const du = scaled(viewport_u, 1.0 / (double)c.image_width), dv = scaled(viewport_v, 1.0 / (double)c.image_height);
    vec3 const offset = add3(scaled(w, -o.focus_distance), scaled(viewport_u, 1.0 / -2.0), scaled(viewport_v, 1.0 / -2.0));
    point3 const upper_left = padd1(o.look_from, offset);   // (point3+ centre offset)
    point3 const pixel00 = padd2(upper_left, scaled(du, 0.5), scaled(dv, 0.5));
    double const defocus_radius = o.focus_distance * std::tan(((o.defocus_angle / 2) * kPi) / 180);
    put3(c.centre, o.look_from); put3(c.pixel_delta_u, du); put3(c.pixel_delta_v, dv); put3(c.pixel00, pixel00);
    put3(c.defocus_u, scaled(u, defocus_radius)); put3(c.defocus_v, scaled(v, defocus_radius));
    c.background[0] = o.background_colour.r; c.background[1] = o.background_colour.g; c.background[2] = o.background_colour.b;
    return c;
}

// ---- flatten ---------------------------------------------------------------------------------------
on_scene_desc flat_scene::desc() const {
    on_scene_desc d{};
    d.objects = objects.data(); d.n_objects = (uint32_t)objects.size();
    d.children = children.data(); d.n_children = (uint32_t)children.size();
    d.materials = materials.data(); d.n_materials = (uint32_t)materials.size();
    d.textures = textures.data(); d.n_textures = (uint32_t)textures.size();
    d.images = images.data(); d.n_images = (uint32_t)images.size();
    d.pixels = pixels.data(); d.n_pixel_bytes = pixels.size();
    d.perlins = perlins.data(); d.n_perlins = (uint32_t)perlins.size();
    d.world = world; d.lights = lights;
    return d;
}

flat_scene flatten(object_ptr const& world, object_ptr const& lights) {
    if (!world) throw std::invalid_argument("onna: flatten: no world");
    flat_scene f;
    f.world = world->flatten(f);
    f.lights = lights ? lights->flatten(f) : -1;
    return f;
}

// ---- scenes (scenes.lisp) --------------------------------------------------------------------------
namespace {
colour col(float r, float g, float b) { return {sf(r), sf(g), sf(b)}; }
// the cameras of the book chapters the scene functions name; the Lisp file sets one for cornell-box only (onna.lisp:9-17)
camera_options book_camera(double vfov, point3 from, point3 at, colour background, double aspect = 16.0 / 9.0, double defocus = 0) {
    camera_options c;
    c.aspect_ratio = aspect; c.image_width = 400; c.samples_per_pixel = 100; c.max_depth = 50; c.background_colour = background;
    c.vertical_fov = vfov; c.look_from = from; c.look_at = at; c.defocus_angle = defocus;
    return c;
}
colour const kSky = {sf(0.7f), sf(0.8f), 1};
}  // namespace

scene two_spheres() {
    return {make_list({make_sphere({0, 0, -1}, sf(0.5f)), make_sphere({0, sf(-100.5f), -1}, 100)}), nullptr, book_camera(90, {0, 0, 0}, {0, 0, -1}, kSky)};
}

scene four_spheres() {
    return {make_list({
                make_sphere({0, sf(-100.5f), -1}, 100, make_lambertian(col(0.8f, 0.8f, 0))),
                make_sphere({0, 0, sf(-1.2f)}, sf(0.5f), make_lambertian(col(0.1f, 0.2f, 0.5f))),
                make_sphere({-1, 0, -1}, sf(0.5f), make_dielectric(1.5)),
                make_sphere({-1, 0, -1}, sf(0.4f), make_dielectric(1.0 / 1.5)),
                make_sphere({1, 0, -1}, sf(0.5f), make_metal(col(0.8f, 0.6f, 0.2f), 1.0)),
            }),
            nullptr, book_camera(20, {-2, 2, 1}, {0, 0, -1}, kSky)};
}

scene bouncing_spheres(random_state& rs) {
    std::vector<object_ptr> world;
    // ground (scenes.lisp:60-66)
    world.push_back(make_sphere({0, -1000, 0}, 1000, make_lambertian(make_chequer(sf(0.32f), make_solid(col(0.2f, 0.3f, 0.1f)), make_solid(col(0.9f, 0.9f, 0.9f))))));
    // scenes.lisp:69-78
    point3 const exclusion{4, sf(0.2f), 0};
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            double const x = a + rs.random(0.9), z = b + rs.random(0.9);
            point3 const centre{x, sf(0.2f), z};
            if (!(length(sub(centre, exclusion)) > sf(0.9f))) continue;
            float const choose = rs.random(1.0f);
            if (choose < 0.8f) {
                vec3 const offset{0, rs.random(0.5), 0};
                float const r = rs.random(1.0f) * rs.random(1.0f), g = rs.random(1.0f) * rs.random(1.0f), bl = rs.random(1.0f) * rs.random(1.0f);
                world.push_back(make_sphere(centre, padd1(centre, offset), sf(0.2f), make_lambertian(col(r, g, bl))));
            } else if (choose < 0.95f) {
                float const r = 0.5f + rs.random(0.5f), g = 0.5f + rs.random(0.5f), bl = 0.5f + rs.random(0.5f);
                double const fuzz = rs.random(0.5);
                world.push_back(make_sphere(centre, sf(0.2f), make_metal(col(r, g, bl), fuzz)));
            } else {
                world.push_back(make_sphere(centre, sf(0.2f), make_dielectric(1.5)));
            }
        }
    // the big spheres are pushed to the front of the list one after the other (scenes.lisp:81-92)
    world.insert(world.begin(), make_sphere({0, 1, 0}, 1, make_dielectric(1.5)));
    world.insert(world.begin(), make_sphere({-4, 1, 0}, 1, make_lambertian(col(0.4f, 0.2f, 0.1f))));
    world.insert(world.begin(), make_sphere({4, 1, 0}, 1, make_metal(col(0.7f, 0.6f, 0.5f))));
    camera_options cam = book_camera(20, {13, 2, 3}, {0, 0, 0}, kSky, 16.0 / 9.0, 0.6);
    cam.focus_distance = 10;
    return {make_bvh_node(std::move(world)), nullptr, cam};
}

scene chequered_spheres() {
    texture_ptr const t = make_chequer(sf(0.32f), make_solid(col(0.2f, 0.3f, 0.1f)), make_solid(col(0.9f, 0.9f, 0.9f)));
    return {make_list({make_sphere({0, -10, 0}, 10, make_lambertian(t)), make_sphere({0, 10, 0}, 10, make_lambertian(t))}), nullptr,
            book_camera(20, {13, 2, 3}, {0, 0, 0}, kSky)};
}

scene earth(image8 image) {
    return {make_sphere({0, 0, 0}, 2, make_lambertian(make_image_texture(std::move(image)))), nullptr, book_camera(20, {0, 0, 12}, {0, 0, 0}, kSky)};
}

scene perlin_spheres(random_state& rs) {
    material_ptr const m = make_lambertian(make_perlin(rs, 4.0));
    return {make_list({make_sphere({0, -1000, 0}, 1000, m), make_sphere({0, 2, 0}, 2, m)}), nullptr, book_camera(20, {13, 2, 3}, {0, 0, 0}, kSky)};
}

scene quads() {
    return {make_list({
                make_quad({-3, -2, 5}, {0, 0, -4}, {0, 4, 0}, make_lambertian(col(1, 0.2f, 0.2f))),
                make_quad({-2, -2, 0}, {4, 0, 0}, {0, 4, 0}, make_lambertian(col(0.2f, 1, 0.2f))),
                make_quad({3, -2, 1}, {0, 0, 4}, {0, 4, 0}, make_lambertian(col(0.2f, 0.2f, 1))),
                make_quad({-2, 3, 1}, {4, 0, 0}, {0, 0, 4}, make_lambertian(col(1, 0.5f, 0))),
                make_quad({-2, -3, 5}, {4, 0, 0}, {0, 0, -4}, make_lambertian(col(0.2f, 0.8f, 0.8f))),
            }),
            nullptr, book_camera(80, {0, 0, 9}, {0, 0, 0}, kSky, 1.0)};
}

scene simple_light(random_state& rs) {
    material_ptr const m = make_lambertian(make_perlin(rs, 4.0)), light = make_diffuse_light(colour{4, 4, 4});
    return {make_list({make_sphere({0, -1000, 0}, 1000, m), make_sphere({0, 2, 0}, 2, m), make_sphere({0, 7, 0}, 2, light),
                       make_quad({3, 1, -2}, {2, 0, 0}, {0, 2, 0}, light)}),
            nullptr, book_camera(20, {26, 3, 6}, {0, 2, 0}, {0, 0, 0})};
}

namespace {
camera_options cornell_camera() {   // onna.lisp:9-17
    camera_options c;
    c.aspect_ratio = 1; c.samples_per_pixel = 10; c.image_width = 600; c.max_depth = 50; c.background_colour = {0, 0, 0};
    c.vertical_fov = 40; c.look_from = {278, 278, -800}; c.look_at = {278, 278, 0}; c.defocus_angle = 0;
    return c;
}
}  // namespace

scene cornell_box() {
    material_ptr const red = make_lambertian(col(0.65f, 0.05f, 0.05f)), white = make_lambertian(col(0.73f, 0.73f, 0.73f)),
                       green = make_lambertian(col(0.12f, 0.45f, 0.15f)), light = make_diffuse_light(colour{15, 15, 15});
    object_ptr const world = make_list({
        make_quad({555, 0, 0}, {0, 555, 0}, {0, 0, 555}, green),
        make_quad({0, 0, 0}, {0, 555, 0}, {0, 0, 555}, red),
        make_quad({343, 554, 332}, {-130, 0, 0}, {0, 0, -105}, light),
        make_quad({0, 0, 0}, {555, 0, 0}, {0, 0, 555}, white),
        make_quad({555, 555, 555}, {-555, 0, 0}, {0, 0, -555}, white),
        make_quad({0, 0, 555}, {555, 0, 0}, {0, 555, 0}, white),
        make_translate(make_rotate_y(make_box({0, 0, 0}, {165, 330, 165}, white), 15), {265, 0, 295}),
        make_sphere({190, 90, 190}, 90, make_dielectric(1.5)),
    });
    object_ptr const lights = make_list({make_quad({343, 554, 332}, {-130, 0, 0}, {0, 0, -105}), make_sphere({190, 90, 190}, 90)});
    return {world, lights, cornell_camera()};
}

scene cornell_smoke() {
    material_ptr const red = make_lambertian(col(0.65f, 0.05f, 0.05f)), white = make_lambertian(col(0.73f, 0.73f, 0.73f)),
                       green = make_lambertian(col(0.12f, 0.45f, 0.15f)), light = make_diffuse_light(colour{7, 7, 7});
    object_ptr box1 = make_translate(make_rotate_y(make_box({0, 0, 0}, {165, 330, 165}, white), 15), {265, 0, 295});
    object_ptr box2 = make_translate(make_rotate_y(make_box({0, 0, 0}, {165, 165, 165}, white), -18), {130, 0, 65});
    object_ptr const world = make_list({
        make_quad({555, 0, 0}, {0, 555, 0}, {0, 0, 555}, green),
        make_quad({0, 0, 0}, {0, 555, 0}, {0, 0, 555}, red),
        make_quad({113, 554, 127}, {330, 0, 0}, {0, 0, 305}, light),
        make_quad({0, 0, 0}, {555, 0, 0}, {0, 0, 555}, white),
        make_quad({555, 555, 555}, {-555, 0, 0}, {0, 0, -555}, white),
        make_quad({0, 0, 555}, {555, 0, 0}, {0, 555, 0}, white),
        make_constant_medium(box1, 0.01, make_solid({0, 0, 0})),
        make_constant_medium(box2, 0.01, make_solid({1, 1, 1})),
    });
    // (the Lisp function returns no lights; the light's own quad is the natural list and is what the camera of onna.lisp would be given)
    return {world, make_list({make_quad({113, 554, 127}, {330, 0, 0}, {0, 0, 305})}), cornell_camera()};
}

scene final_scene(random_state& rs, image8 image) {
    // scenes.lisp:208-294; the world is a list built by pushes, so it ends up in the reverse order of the pushes
    std::vector<object_ptr> pushed;
    {   // ground boxes (scenes.lisp:210-226)
        material_ptr const ground = make_lambertian(col(0.48f, 0.83f, 0.53f));
        std::vector<object_ptr> boxes;
        for (int i = 0; i < 20; ++i)
            for (int j = 0; j < 20; ++j) {
                double const w = 100, x0 = -1000 + i * w, y0 = 0, z0 = -1000 + j * w;
                double const x1 = x0 + w, y1 = 1 + rs.random(100.0), z1 = z0 + w;
                boxes.push_back(make_box({x0, y0, z0}, {x1, y1, z1}, ground));
            }
        pushed.push_back(make_bvh_node(std::move(boxes)));
    }
    pushed.push_back(make_quad({123, 554, 147}, {300, 0, 0}, {0, 0, 265}, make_diffuse_light(colour{7, 7, 7})));
    {
        point3 const from{400, 400, 200}, to = padd1(from, {30, 0, 0});
        pushed.push_back(make_sphere(from, to, 50, make_lambertian(col(0.7f, 0.3f, 0.1f))));
    }
    pushed.push_back(make_sphere({260, 150, 45}, 50, make_dielectric(1.5)));
    pushed.push_back(make_sphere({0, 150, 145}, 50, make_metal(col(0.8f, 0.8f, 0.9f), 1.0)));
    {
        object_ptr const boundary = make_sphere({360, 150, 145}, 70, make_dielectric(1.5));
        pushed.push_back(boundary);
        pushed.push_back(make_constant_medium(boundary, sf(0.2f), make_solid(col(0.2f, 0.4f, 0.9f))));
    }
    pushed.push_back(make_constant_medium(make_sphere({0, 0, 0}, 5000, make_dielectric(1.5)), sf(0.0001f), make_solid({1, 1, 1})));
    pushed.push_back(make_sphere({400, 200, 400}, 100, make_lambertian(make_image_texture(std::move(image)))));
    pushed.push_back(make_sphere({220, 280, 300}, 80, make_lambertian(make_perlin(rs, 0.2))));
    {   // scenes.lisp:227-240
        material_ptr const white = make_lambertian(col(0.73f, 0.73f, 0.73f));
        std::vector<object_ptr> spheres;
        for (int j = 0; j < 1000; ++j) {
            double const x = rs.random(165.0), y = rs.random(165.0), z = rs.random(165.0);
            spheres.push_back(make_sphere({x, y, z}, 10, white));
        }
        pushed.push_back(make_translate(make_rotate_y(make_bvh_node(std::move(spheres)), 15), {-100, 270, 395}));
    }
    std::reverse(pushed.begin(), pushed.end());
    camera_options cam = book_camera(40, {478, 278, -600}, {278, 278, 0}, {0, 0, 0}, 1.0);
    cam.image_width = 800;
    return {make_list(std::move(pushed)), make_list({make_quad({123, 554, 147}, {300, 0, 0}, {0, 0, 265})}), cam};
}

std::vector<std::string> scene_names() {
    return {"two-spheres", "four-spheres", "bouncing-spheres", "chequered-spheres", "earth", "perlin-spheres", "quads", "simple-light", "cornell-box", "cornell-smoke", "final-scene"};
}

scene scene_by_name(std::string const& name, random_state& rs, image8 const& image) {
    auto need_image = [&]() -> image8 {
        if (image.rgb.empty()) throw std::invalid_argument("onna: scene " + name + " needs an image (earthmap.png in the reference)");
        return image;
    };
    if (name == "two-spheres") return two_spheres();
    if (name == "four-spheres") return four_spheres();
    if (name == "bouncing-spheres") return bouncing_spheres(rs);
    if (name == "chequered-spheres") return chequered_spheres();
    if (name == "earth") return earth(need_image());
    if (name == "perlin-spheres") return perlin_spheres(rs);
    if (name == "quads") return quads();
    if (name == "simple-light") return simple_light(rs);
    if (name == "cornell-box") return cornell_box();
    if (name == "cornell-smoke") return cornell_smoke();
    if (name == "final-scene") return final_scene(rs, need_image());
    throw std::invalid_argument("onna: unknown scene " + name);
}

// ---- PNG (image.lisp) ------------------------------------------------------------------------------
namespace {
uint32_t be32(unsigned char const* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(std::vector<unsigned char>& v, uint32_t x) { v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x); }
}  // namespace

image8 read_png(std::string const& path) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("onna: cannot open " + path);
    std::vector<unsigned char> bytes;
    unsigned char buf[65536];
    for (size_t n; (n = std::fread(buf, 1, sizeof buf, f)) > 0;) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    static unsigned char const signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    if (bytes.size() < 8 || std::memcmp(bytes.data(), signature, 8) != 0) throw std::runtime_error("onna: " + path + " is not a PNG file");
    uint32_t width = 0, height = 0;
    int channels = 0;
    std::vector<unsigned char> idat;
    for (size_t at = 8; at + 12 <= bytes.size();) {
        uint32_t const n = be32(&bytes[at]);
        if (at + 12 + (size_t)n > bytes.size()) throw std::runtime_error("onna: " + path + ": truncated chunk");
        char const* type = (char const*)&bytes[at + 4];
        unsigned char const* data = &bytes[at + 8];
        if (std::memcmp(type, "IHDR", 4) == 0 && n >= 13) {
            width = be32(data); height = be32(data + 4);
            if (data[8] != 8 || (data[9] != 2 && data[9] != 6 && data[9] != 0) || data[12] != 0)
                throw std::runtime_error("onna: " + path + ": only 8-bit grey, RGB or RGBA PNGs without interlacing are read");
            channels = data[9] == 2 ? 3 : (data[9] == 6 ? 4 : 1);
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), data, data + n);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        at += 12 + (size_t)n;
    }
    if (!width || !height || !channels || (uint64_t)width * height > ((uint64_t)1 << 28)) throw std::runtime_error("onna: " + path + ": no usable image header");
    size_t const stride = (size_t)width * channels;
    std::vector<unsigned char> raw((stride + 1) * height);
    uLongf got = (uLongf)raw.size();
    if (uncompress(raw.data(), &got, idat.data(), (uLong)idat.size()) != Z_OK || got != raw.size()) throw std::runtime_error("onna: " + path + ": bad image data");
    image8 im;
    im.width = width; im.height = height;
    im.rgb.resize((size_t)width * height * 3);
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    for (uint32_t y = 0; y < height; ++y) {
        unsigned char const* row = &raw[(stride + 1) * y];
        int const filter = row[0];
        for (size_t x = 0; x < stride; ++x) {
            int const a = x >= (size_t)channels ? cur[x - channels] : 0, b = prev[x], c = x >= (size_t)channels ? prev[x - channels] : 0;
            int pred = 0;
            if (filter == 1) pred = a;
            else if (filter == 2) pred = b;
            else if (filter == 3) pred = (a + b) / 2;
            else if (filter == 4) { int const p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (filter != 0) throw std::runtime_error("onna: " + path + ": unknown scanline filter");
            cur[x] = (unsigned char)(row[1 + x] + pred);
        }
        for (uint32_t x = 0; x < width; ++x)
            for (int k = 0; k < 3; ++k) im.rgb[((size_t)y * width + x) * 3 + k] = cur[(size_t)x * channels + (channels == 1 ? 0 : k)];
        prev = cur;
    }
    return im;
}

void write_png(std::string const& path, image8 const& image) {
    if (image.rgb.size() != (size_t)image.width * image.height * 3) throw std::invalid_argument("onna: write_png: ragged image");
    size_t const stride = (size_t)image.width * 3;
    std::vector<unsigned char> raw((stride + 1) * image.height);
    for (uint32_t y = 0; y < image.height; ++y) {
        raw[(stride + 1) * y] = 0;
        std::memcpy(&raw[(stride + 1) * y + 1], &image.rgb[stride * y], stride);
    }
    uLongf zn = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zn);
    if (compress2(z.data(), &zn, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw std::runtime_error("onna: write_png: compression failed");
    z.resize(zn);
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("onna: cannot write " + path);
    auto chunk = [&](char const* type, std::vector<unsigned char> const& data) {
        std::vector<unsigned char> out;
        put_be32(out, (uint32_t)data.size());
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        put_be32(out, (uint32_t)crc32(0, out.data() + 4, (uInt)(out.size() - 4)));
        std::fwrite(out.data(), 1, out.size(), f);
    };
    static unsigned char const signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::fwrite(signature, 1, 8, f);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, image.width); put_be32(ihdr, image.height);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    std::fclose(f);
}

// ---- the GPU ---------------------------------------------------------------------------------------
device_scene::device_scene(object_ptr const& world, object_ptr const& lights, int device) : flat_(flatten(world, lights)) {
    on_scene_desc const d = flat_.desc();
    if (on_scene_create(&d, device, &scene_)) throw std::runtime_error(std::string("onna: on_scene_create: ") + on_last_error());
}
device_scene::~device_scene() { on_scene_destroy(scene_); }

frame render(on_camera const& camera, device_scene const& scene, uint64_t seed) {
    frame f;
    f.width = camera.image_width; f.height = camera.image_height; f.samples = camera.sqrt_spp * camera.sqrt_spp;
    f.sums.assign((size_t)f.width * f.height * 3, 0.0);
    on_render_params const p{0, f.samples, seed};
    if (on_render_host(scene.handle(), &camera, &p, f.sums.data(), &f.stats)) throw std::runtime_error(std::string("onna: on_render_host: ") + on_last_error());
    return f;
}

frame render(on_camera const& camera, object_ptr const& world, object_ptr const& lights, uint64_t seed, int device) {
    device_scene const scene(world, lights, device);
    return render(camera, scene, seed);
}

frame render(on_camera const& camera, object_ptr const& world, object_ptr const& lights, uint64_t seed, std::vector<int> const& devices) {
    if (devices.empty()) return render(camera, world, lights, seed, -1);
    size_t const n = devices.size();
    uint32_t const samples = camera.sqrt_spp * camera.sqrt_spp;
    std::vector<frame> parts(n);
    std::vector<std::string> errors(n);
    auto share = [&](size_t g) {
        try {
            device_scene const scene(world, lights, devices[g]);
            frame& f = parts[g];
            f.width = camera.image_width; f.height = camera.image_height; f.samples = samples;
            f.sums.assign((size_t)f.width * f.height * 3, 0.0);
            on_render_params const p{(uint32_t)((uint64_t)samples * g / n), (uint32_t)((uint64_t)samples * (g + 1) / n), seed};
            if (p.sample_end > p.sample_begin && on_render_host(scene.handle(), &camera, &p, f.sums.data(), &f.stats))
                throw std::runtime_error(std::string("on_render_host: ") + on_last_error());
        } catch (std::exception const& e) {
            errors[g] = e.what();
        }
    };
    std::vector<std::thread> threads;
    for (size_t g = 1; g < n; ++g) threads.emplace_back(share, g);
    share(0);
    for (auto& t : threads) t.join();
    for (size_t g = 0; g < n; ++g)
        if (!errors[g].empty()) throw std::runtime_error("onna: render on device " + std::to_string(devices[g]) + ": " + errors[g]);
    frame out = std::move(parts[0]);
    for (size_t g = 1; g < n; ++g) {
        for (size_t i = 0; i < out.sums.size(); ++i) out.sums[i] += parts[g].sums[i];
        out.stats.rays += parts[g].stats.rays;
        out.stats.kernel_ms = std::max(out.stats.kernel_ms, parts[g].stats.kernel_ms);
    }
    out.stats.paths = (uint64_t)out.width * out.height * samples;
    return out;
}

// colour.lisp:12-18,20-37; image.lisp:30-40 (the same arithmetic as on_resolve; a frame's worth on the host)
image8 frame::resolve() const {
    image8 im;
    im.width = width; im.height = height;
    im.rgb.resize(sums.size());
    for (size_t i = 0; i < sums.size(); ++i) {
        double x = std::sqrt(sums[i] / (double)samples);
        x = x < 0 ? 0 : (x > sf(0.999f) ? sf(0.999f) : x);
        if (!(x == x)) x = 0;
        im.rgb[i] = (uint8_t)std::floor(256 * x);
    }
    return im;
}

on_hit hit_test(rt_ray const& ray, object_ptr const& o, interval ray_interval, int device) {
    device_scene const scene(o, nullptr, device);
    on_hit h{};
    if (on_hit_batch(scene.handle(), scene.tables().world, &ray, 1, ray_interval.min, ray_interval.max, nullptr, 0, &h))
        throw std::runtime_error(std::string("onna: on_hit_batch: ") + on_last_error());
    return h;
}

}  // namespace onna

// ---- C entry points for callers that cannot use the classes (tests, bench.py, tools) ----------------
namespace {
thread_local std::string g_onh_error;
template <class F> int onh_guarded(F&& f) {
    try { f(); return 0; }
    catch (std::exception const& e) { g_onh_error = e.what(); }
    catch (...) { g_onh_error = "unknown error"; }
    return 1;
}
}  // namespace

struct onh_scene {
    onna::scene built;
    onna::flat_scene flat;
    on_scene_desc desc{};
};

struct onh_builder {
    std::vector<onna::texture_ptr> textures;
    std::vector<onna::material_ptr> materials;
    std::vector<onna::object_ptr> objects;
    template <class T> T const& at(std::vector<T> const& v, int32_t i, char const* what) const {
        if (i < 0 || (size_t)i >= v.size()) throw std::invalid_argument(std::string("onh builder: ") + what + " handle out of range");
        return v[(size_t)i];
    }
    std::vector<onna::object_ptr> many(int32_t const* ids, uint32_t n) const {
        std::vector<onna::object_ptr> out;
        for (uint32_t i = 0; i < n; ++i) out.push_back(at(objects, ids[i], "object"));
        return out;
    }
};

extern "C" {

char const* onh_last_error(void) { return g_onh_error.c_str(); }

// a scene of scenes.lisp by name, built with the generator seeded `seed`; image_path may be null
int onh_scene_build(char const* name, uint64_t seed, char const* image_path, onh_scene** out) {
    return onh_guarded([&] {
        onna::random_state rs(seed);
        onna::image8 image;
        if (image_path && *image_path) image = onna::read_png(image_path);
        auto s = std::make_unique<onh_scene>();
        s->built = onna::scene_by_name(name, rs, image);
        s->flat = onna::flatten(s->built.world, s->built.lights);
        s->desc = s->flat.desc();
        *out = s.release();
    });
}
on_scene_desc const* onh_scene_desc(onh_scene const* s) { return &s->desc; }
void onh_scene_free(onh_scene* s) { delete s; }

// the scene's camera with an image width, sample count and depth limit of the caller's (0 keeps the scene's own)
int onh_scene_camera(onh_scene const* s, uint32_t image_width, uint32_t samples_per_pixel, uint32_t max_depth, on_camera* out) {
    return onh_guarded([&] {
        onna::camera_options o = s->built.camera;
        if (image_width) o.image_width = image_width;
        if (samples_per_pixel) o.samples_per_pixel = samples_per_pixel;
        if (max_depth) o.max_depth = max_depth;
        *out = onna::make_camera(o);
    });
}

// make-camera from loose values: aspect, width, spp, depth, background(3), vfov, from(3), at(3), up(3), defocus angle, focus distance
int onh_make_camera(double const* v, on_camera* out) {
    return onh_guarded([&] {
        onna::camera_options o;
        o.aspect_ratio = v[0]; o.image_width = (uint32_t)v[1]; o.samples_per_pixel = (uint32_t)v[2]; o.max_depth = (uint32_t)v[3];
        o.background_colour = {v[4], v[5], v[6]}; o.vertical_fov = v[7];
        o.look_from = {v[8], v[9], v[10]}; o.look_at = {v[11], v[12], v[13]}; o.up = {v[14], v[15], v[16]};
        o.defocus_angle = v[17]; o.focus_distance = v[18];
        *out = onna::make_camera(o);
    });
}

// ---- a scene put together call by call (the constructors of onna.hpp behind integer handles) ----
onh_builder* onh_builder_new(void) { return new (std::nothrow) onh_builder; }
void onh_builder_free(onh_builder* b) { delete b; }

#define ONH_ADD(vec, expr) do { int32_t id_ = -1; if (onh_guarded([&] { b->vec.push_back(expr); id_ = (int32_t)b->vec.size() - 1; })) return -1; return id_; } while (0)

int32_t onh_solid(onh_builder* b, double r, double g, double bl) { ONH_ADD(textures, onna::make_solid({r, g, bl})); }
int32_t onh_chequer(onh_builder* b, double scale, int32_t even, int32_t odd) { ONH_ADD(textures, onna::make_chequer(scale, b->at(b->textures, even, "texture"), b->at(b->textures, odd, "texture"))); }
int32_t onh_perlin(onh_builder* b, uint64_t seed, double scale) { ONH_ADD(textures, [&] { onna::random_state rs(seed); return onna::make_perlin(rs, scale); }()); }
int32_t onh_image(onh_builder* b, uint32_t width, uint32_t height, uint8_t const* rgb) {
    ONH_ADD(textures, [&] { onna::image8 im; im.width = width; im.height = height; im.rgb.assign(rgb, rgb + (size_t)width * height * 3); return onna::make_image_texture(std::move(im)); }());
}
// kind: ON_MAT_*; tex for lambertian, diffuse light and isotropic; albedo and fuzz for metal; refraction index for dielectric
int32_t onh_material(onh_builder* b, uint32_t kind, int32_t tex, double const* albedo, double fuzz, double refraction_index) {
    ONH_ADD(materials, [&]() -> onna::material_ptr {
        switch (kind) {
        case ON_MAT_LAMBERTIAN: return onna::make_lambertian(b->at(b->textures, tex, "texture"));
        case ON_MAT_METAL: return onna::make_metal({albedo[0], albedo[1], albedo[2]}, fuzz);
        case ON_MAT_DIELECTRIC: return onna::make_dielectric(refraction_index);
        case ON_MAT_DIFFUSE_LIGHT: return onna::make_diffuse_light(b->at(b->textures, tex, "texture"));
        case ON_MAT_ISOTROPIC: return onna::make_isotropic(b->at(b->textures, tex, "texture"));
        }
        throw std::invalid_argument("onh_material: unknown kind");
    }());
}
static onna::material_ptr onh_optional_material(onh_builder* b, int32_t m) { return m < 0 ? nullptr : b->at(b->materials, m, "material"); }
// to = null: a sphere that stays where it is
int32_t onh_sphere(onh_builder* b, double const* centre, double const* to, double radius, int32_t material) {
    ONH_ADD(objects, to ? onna::make_sphere({centre[0], centre[1], centre[2]}, {to[0], to[1], to[2]}, radius, onh_optional_material(b, material))
                        : onna::make_sphere({centre[0], centre[1], centre[2]}, radius, onh_optional_material(b, material)));
}
int32_t onh_quad(onh_builder* b, double const* q, double const* u, double const* v, int32_t material) {
    ONH_ADD(objects, onna::make_quad({q[0], q[1], q[2]}, {u[0], u[1], u[2]}, {v[0], v[1], v[2]}, onh_optional_material(b, material)));
}
int32_t onh_box(onh_builder* b, double const* p, double const* q, int32_t material) {
    ONH_ADD(objects, onna::make_box({p[0], p[1], p[2]}, {q[0], q[1], q[2]}, onh_optional_material(b, material)));
}
int32_t onh_list(onh_builder* b, int32_t const* ids, uint32_t n) { ONH_ADD(objects, onna::make_list(b->many(ids, n))); }
int32_t onh_bvh(onh_builder* b, int32_t const* ids, uint32_t n) { ONH_ADD(objects, onna::make_bvh_node(b->many(ids, n))); }
int32_t onh_translate(onh_builder* b, int32_t object, double const* offset) { ONH_ADD(objects, onna::make_translate(b->at(b->objects, object, "object"), {offset[0], offset[1], offset[2]})); }
int32_t onh_rotate_y(onh_builder* b, int32_t object, double angle) { ONH_ADD(objects, onna::make_rotate_y(b->at(b->objects, object, "object"), angle)); }
int32_t onh_medium(onh_builder* b, int32_t boundary, double density, int32_t tex) {
    ONH_ADD(objects, onna::make_constant_medium(b->at(b->objects, boundary, "object"), density, b->at(b->textures, tex, "texture")));
}
#undef ONH_ADD

// the flattened scene of `world` (and `lights`, or -1); the camera is the default one of make-camera
int onh_builder_finish(onh_builder const* b, int32_t world, int32_t lights, onh_scene** out) {
    return onh_guarded([&] {
        auto s = std::make_unique<onh_scene>();
        s->built.world = b->at(b->objects, world, "object");
        s->built.lights = lights < 0 ? nullptr : b->at(b->objects, lights, "object");
        s->flat = onna::flatten(s->built.world, s->built.lights);
        s->desc = s->flat.desc();
        *out = s.release();
    });
}

int onh_read_png(char const* path, uint32_t* width, uint32_t* height, uint8_t* rgb, size_t capacity) {
    return onh_guarded([&] {
        onna::image8 const im = onna::read_png(path);
        *width = im.width; *height = im.height;
        if (rgb) {
            if (capacity < im.rgb.size()) throw std::invalid_argument("onh_read_png: buffer too small");
            std::memcpy(rgb, im.rgb.data(), im.rgb.size());
        }
    });
}

int onh_write_png(char const* path, uint32_t width, uint32_t height, uint8_t const* rgb) {
    return onh_guarded([&] {
        onna::image8 im;
        im.width = width; im.height = height;
        im.rgb.assign(rgb, rgb + (size_t)width * height * 3);
        onna::write_png(path, im);
    });
}

}  // extern "C"
