// rtiow.cpp -- host-side scene API: value types, bounding boxes, the BVH build, the
// camera basis, flattening into the C ABI's tables, and the thin calls into the GPU
// library for everything on the per-ray path.  Reference citations are file:line under
// c++/ of hgiddens/ray-tracing-in-one-weekend.
#include "rtiow.hpp"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <future>
#include <map>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <exception>

namespace rtiow {

namespace {

[[noreturn]] void device_failure(char const* what) {
    throw std::runtime_error(std::string(what) + ": " + rt_last_error());
}

rt_ray to_abi(ray const& r) {
    return {{r.origin().x(), r.origin().y(), r.origin().z()}, {r.direction().x(), r.direction().y(), r.direction().z()}, r.time()};
}

ray from_abi(rt_ray const& r) {
    return {vec3{r.origin[0], r.origin[1], r.origin[2]}, vec3{r.direction[0], r.direction[1], r.direction[2]}, r.time};
}

// Uniforms the GPU call will consume, taken from a COPY of the shared generator as
// std::uniform_real_distribution would take them (two 32-bit draws per value).
std::vector<double> peek_uniforms(std::mt19937 const& mt, size_t k) {
    std::mt19937 copy = mt;
    std::vector<double> tape(k);
    for (double& u : tape) u = std::generate_canonical<double, 53>(copy);
    return tape;
}

void advance(std::mt19937& mt, int used) { mt.discard(2ull * (unsigned long long)used); }

int g_probe_device = -1;

}  // namespace

int probe_device() { return g_probe_device; }
void set_probe_device(int device) { g_probe_device = device; }

// ---------------------------------------------------------------- value types ----
// vec3.cpp:16-74: component-wise, scalar division divides each component
vec3 vec3::cross(vec3 that) const {
    return {y() * that.z() - z() * that.y(), -(x() * that.z() - z() * that.x()), x() * that.y() - y() * that.x()};
}
double vec3::dot(vec3 that) const { return x() * that.x() + y() * that.y() + z() * that.z(); }
double vec3::length() const { return std::sqrt(squared_length()); }
double vec3::squared_length() const { return x() * x() + y() * y() + z() * z(); }
vec3 vec3::unit_vector() const { return *this / length(); }
vec3 operator+(vec3 a, vec3 b) { return {a.x() + b.x(), a.y() + b.y(), a.z() + b.z()}; }
vec3 operator-(vec3 a, vec3 b) { return {a.x() - b.x(), a.y() - b.y(), a.z() - b.z()}; }
vec3 operator*(vec3 v, double d) { return {v.x() * d, v.y() * d, v.z() * d}; }
vec3 operator*(double d, vec3 v) { return {v.x() * d, v.y() * d, v.z() * d}; }
vec3 operator/(vec3 v, double d) { return {v.x() / d, v.y() / d, v.z() / d}; }

// colour.cpp:19-46
colour colour::gamma2() const { return {std::sqrt(red()), std::sqrt(green()), std::sqrt(blue())}; }
colour operator+(colour a, colour b) { return {a.red() + b.red(), a.green() + b.green(), a.blue() + b.blue()}; }
colour operator*(colour c, double d) { return {c.red() * d, c.green() * d, c.blue() * d}; }
colour operator*(double d, colour c) { return {c.red() * d, c.green() * d, c.blue() * d}; }

// aabb.cpp:59-64
aabb aabb::surrounding_box(aabb box0, aabb box1) {
    vec3 const small{std::fmin(box0.min().x(), box1.min().x()), std::fmin(box0.min().y(), box1.min().y()),
                     std::fmin(box0.min().z(), box1.min().z())};
    vec3 const big{std::fmax(box0.max().x(), box1.max().x()), std::fmax(box0.max().y(), box1.max().y()),
                   std::fmax(box0.max().z(), box1.max().z())};
    return aabb(small, big);
}

bool aabb::hit(ray const& r, double tmin, double tmax) const {
    double const box[6] = {_min.x(), _min.y(), _min.z(), _max.x(), _max.y(), _max.z()};
    rt_ray const rr = to_abi(r);
    uint8_t out = 0;
    if (rt_aabb_hit_batch(probe_device(), box, &rr, &tmin, &tmax, 1, &out)) device_failure("aabb::hit");
    return out != 0;
}

// ------------------------------------------------------------------ flattening ----
// Appends another flattener's tables (a subtree flattened on its own); `ref` is a reference into them --
// a node index, or ~(primitive index) -- and comes back as the reference into this one.
int32_t scene_flattener::append(scene_flattener const& o, int32_t ref) {
    int32_t const node_base = (int32_t)nodes.size(), prim_base = (int32_t)prims.size();
    uint32_t const mat_base = (uint32_t)materials.size(), tex_base = (uint32_t)textures.size();
    auto moved = [&](int32_t x) { return x >= 0 ? x + node_base : x - prim_base; };
    nodes.reserve(nodes.size() + o.nodes.size());
    for (rt_node_desc n : o.nodes) { n.left = moved(n.left); n.right = moved(n.right); nodes.push_back(n); }
    prims.reserve(prims.size() + o.prims.size());
    for (rt_prim_desc q : o.prims) { q.material += mat_base; prims.push_back(q); }
    materials.reserve(materials.size() + o.materials.size());
    for (rt_material_desc m : o.materials) { if (m.kind == RT_MAT_LAMBERTIAN) m.texture += tex_base; materials.push_back(m); }
    textures.reserve(textures.size() + o.textures.size());
    for (rt_texture_desc t : o.textures) { if (t.kind == RT_TEX_CHEQUER) { t.even += tex_base; t.odd += tex_base; } textures.push_back(t); }
    material_of_prim.insert(material_of_prim.end(), o.material_of_prim.begin(), o.material_of_prim.end());
    for (rt_item_desc it : o.items) { it.index = it.is_bvh ? it.index + node_base : it.index + prim_base; items.push_back(it); }
    return moved(ref);
}

rt_scene_desc scene_flattener::desc() const {
    rt_scene_desc d{};
    d.textures = textures.data(); d.n_textures = (uint32_t)textures.size();
    d.materials = materials.data(); d.n_materials = (uint32_t)materials.size();
    d.prims = prims.data(); d.n_prims = (uint32_t)prims.size();
    d.nodes = nodes.data(); d.n_nodes = (uint32_t)nodes.size();
    d.items = items.data(); d.n_items = (uint32_t)items.size();
    d.perlin_ranfloat = perlin::ranfloat.data();
    d.perlin_perm_x = perlin::perm_x.data();
    d.perlin_perm_y = perlin::perm_y.data();
    d.perlin_perm_z = perlin::perm_z.data();
    return d;
}

device_scene::device_scene(object const& world, int device) {
    auto const t0 = std::chrono::steady_clock::now();
    world.flatten_items(flat);
    auto const t1 = std::chrono::steady_clock::now();
    rt_scene_desc const d = flat.desc();
    if (rt_scene_create(&d, device, &handle)) device_failure("device_scene");
    flatten_seconds = std::chrono::duration<double>(t1 - t0).count();
    create_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
}

device_scene::~device_scene() { rt_scene_destroy(handle); }

// -------------------------------------------------------------------- textures ----
texture::texture() = default;
texture::~texture() = default;

uint32_t constant_texture::flatten(scene_flattener& out) const {
    rt_texture_desc t{};
    t.kind = RT_TEX_CONSTANT;
    t.colour[0] = c.red(); t.colour[1] = c.green(); t.colour[2] = c.blue();
    out.textures.push_back(t);
    return (uint32_t)out.textures.size() - 1;
}

chequer_texture::chequer_texture(std::unique_ptr<texture const> even, std::unique_ptr<texture const> odd)
    : even(std::move(even)), odd(std::move(odd)) {}

uint32_t chequer_texture::flatten(scene_flattener& out) const {
    rt_texture_desc t{};
    t.kind = RT_TEX_CHEQUER;
    t.even = even->flatten(out);
    t.odd = odd->flatten(out);
    out.textures.push_back(t);
    return (uint32_t)out.textures.size() - 1;
}

uint32_t perlin_texture::flatten(scene_flattener& out) const {
    rt_texture_desc t{};
    t.kind = RT_TEX_PERLIN;
    out.textures.push_back(t);
    return (uint32_t)out.textures.size() - 1;
}

namespace {

// A unit sphere carrying one material: the smallest scene through which a lone texture
// or material can be evaluated on the GPU.
class probe_object final : public object {
    material const& m;
public:
    explicit probe_object(material const& m) : m(m) {}
    std::optional<aabb> bounding_box(double, double) const override { return aabb(vec3{-1, -1, -1}, vec3{1, 1, 1}); }
    int32_t flatten_child(scene_flattener& out, aabb const&) const override {
        rt_prim_desc p{};
        p.kind = RT_PRIM_SPHERE;
        p.material = m.flatten(out);
        p.radius = 1; p.time1 = 1;
        out.prims.push_back(p);
        out.material_of_prim.push_back(&m);
        return ~(int32_t)(out.prims.size() - 1);
    }
};

class probe_material final : public material {
    texture const& t;
    static std::mt19937 unused;
public:
    explicit probe_material(texture const& t) : material(unused), t(t) {}
    uint32_t flatten(scene_flattener& out) const override {
        rt_material_desc m{};
        m.kind = RT_MAT_LAMBERTIAN;
        m.texture = t.flatten(out);
        out.materials.push_back(m);
        return (uint32_t)out.materials.size() - 1;
    }
};
std::mt19937 probe_material::unused;

}  // namespace

colour texture::value(double, double, vec3 const& p) const {
    if (!probe) {
        probe_material pm(*this);
        probe_object po(pm);
        probe = std::make_unique<device_scene>(po, probe_device());
    }
    double const in[3] = {p.x(), p.y(), p.z()};
    double out[3];
    // the texture itself is the last entry of the probe's table
    uint32_t const self = (uint32_t)probe->tables().textures.size() - 1;
    if (rt_texture_batch(probe->get(), self, in, 1, out)) device_failure("texture::value");
    return {out[0], out[1], out[2]};
}

// perlin.cpp:8-31,50-53 with glibc's drand48() in its default state written out:
// X <- (0x5DEECE66D X + 0xB) mod 2^48 from X = 0 (glibc zero-initialises the state),
// value X / 2^48.
namespace {
struct lcg48 {
    uint64_t x = 0;
    double next() {
        x = (x * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL;
        return (double)x / 281474976710656.0;
    }
};
lcg48& perlin_source() { static lcg48 g; return g; }

std::vector<double> perlin_generate() {
    std::vector<double> p(256);
    for (double& v : p) v = perlin_source().next();
    return p;
}

std::vector<uint8_t> perlin_generate_perm() {
    std::vector<uint8_t> p(256);
    for (size_t i = 0; i < p.size(); ++i) p[i] = (uint8_t)i;
    for (size_t i = p.size() - 1; i > 0; --i) {
        size_t const target = (size_t)int(perlin_source().next() * (i + 1));
        std::swap(p[i], p[target]);
    }
    return p;
}
}  // namespace

std::vector<double> const perlin::ranfloat = perlin_generate();
std::vector<uint8_t> const perlin::perm_x = perlin_generate_perm();
std::vector<uint8_t> const perlin::perm_y = perlin_generate_perm();
std::vector<uint8_t> const perlin::perm_z = perlin_generate_perm();

double perlin::noise(vec3 const& p) const { return perlin_texture{}.value(0, 0, p).red(); }

// ------------------------------------------------------------------- materials ----
material::material(std::mt19937& mt) : mt(mt) {}
material::~material() = default;

std::optional<scatter_record> material::scatter(ray const& r, hit_record const& hit) const {
    if (!probe) {
        probe_object po(*this);
        probe = std::make_unique<device_scene>(po, probe_device());
    }
    rt_ray const in = to_abi(r);
    rt_hit h{};
    h.hit = 1; h.prim = 0; h.t = hit.t;
    h.p[0] = hit.p.x(); h.p[1] = hit.p.y(); h.p[2] = hit.p.z();
    h.normal[0] = hit.normal.x(); h.normal[1] = hit.normal.y(); h.normal[2] = hit.normal.z();
    int32_t const prim = 0;
    for (uint32_t k = 64;; k *= 4) {
        std::vector<double> const tape = peek_uniforms(mt, k);
        uint8_t scattered = 0;
        double att[3];
        rt_ray out;
        int32_t used = 0;
        int const rc = rt_scatter_batch(probe->get(), &prim, &in, &h, tape.data(), k, 1, &scattered, att, &out, &used);
        if (rc == RT_E_TAPE && k < (1u << 20)) continue;
        if (rc) device_failure("material::scatter");
        advance(mt, used);
        if (!scattered) return std::nullopt;
        return scatter_record{vec3{att[0], att[1], att[2]}, from_abi(out)};
    }
}

lambertian::lambertian(std::mt19937& mt, std::unique_ptr<texture const> texture) : material(mt), t(std::move(texture)) {}

uint32_t lambertian::flatten(scene_flattener& out) const {
    rt_material_desc m{};
    m.kind = RT_MAT_LAMBERTIAN;
    m.texture = t->flatten(out);
    out.materials.push_back(m);
    return (uint32_t)out.materials.size() - 1;
}

metal::metal(std::mt19937& mt, vec3 const albedo, double const fuzziness) : material(mt), albedo(albedo), fuzziness(fuzziness) {
    assert(fuzziness <= 1);
}

uint32_t metal::flatten(scene_flattener& out) const {
    rt_material_desc m{};
    m.kind = RT_MAT_METAL;
    m.albedo[0] = albedo.x(); m.albedo[1] = albedo.y(); m.albedo[2] = albedo.z();
    m.fuzziness = fuzziness;
    out.materials.push_back(m);
    return (uint32_t)out.materials.size() - 1;
}

dielectric::dielectric(std::mt19937& mt, double const refractive_index) : material(mt), refractive_index(refractive_index) {
    assert(refractive_index >= 0);
}

uint32_t dielectric::flatten(scene_flattener& out) const {
    rt_material_desc m{};
    m.kind = RT_MAT_DIELECTRIC;
    m.refractive_index = refractive_index;
    out.materials.push_back(m);
    return (uint32_t)out.materials.size() - 1;
}

// --------------------------------------------------------------------- objects ----
object::object() = default;
object::~object() = default;

void object::scan_members(std::vector<object const*>& out) const { out.push_back(this); }

void object::flatten_items(scene_flattener& out) const {
    // (only a list looks at its parent's box, and a list appends its members as items instead)
    int32_t const ref = flatten_child(out, aabb(vec3{0, 0, 0}, vec3{0, 0, 0}));
    rt_item_desc it{};
    it.is_bvh = ref >= 0;
    it.index = ref >= 0 ? ref : ~ref;
    out.items.push_back(it);
}

std::optional<hit_record> object::hit(ray const& r, double t_min, double t_max) const {
    if (!self_scene) self_scene = std::make_unique<device_scene>(*this, probe_device());
    rt_ray const in = to_abi(r);
    rt_hit h;
    if (rt_trace_batch(self_scene->get(), &in, 1, t_min, t_max, &h)) device_failure("object::hit");
    if (!h.hit) return std::nullopt;
    return hit_record{h.t, vec3{h.p[0], h.p[1], h.p[2]}, vec3{h.normal[0], h.normal[1], h.normal[2]},
                      self_scene->tables().material_of_prim[(size_t)h.prim]};
}

sphere::sphere(vec3 const centre, double radius, std::unique_ptr<material const> mat)
    : centre(centre), radius(radius), mat(std::move(mat)) {}

// sphere.cpp:47-50
std::optional<aabb> sphere::bounding_box(double, double) const {
    vec3 const rv = vec3(radius, radius, radius);
    return aabb(centre - rv, centre + rv);
}

int32_t sphere::flatten_child(scene_flattener& out, aabb const&) const {
    rt_prim_desc p{};
    p.kind = RT_PRIM_SPHERE;
    p.material = mat->flatten(out);
    p.centre0[0] = p.centre1[0] = centre.x();
    p.centre0[1] = p.centre1[1] = centre.y();
    p.centre0[2] = p.centre1[2] = centre.z();
    p.time0 = 0; p.time1 = 1; p.radius = radius;
    out.prims.push_back(p);
    out.material_of_prim.push_back(mat.get());
    return ~(int32_t)(out.prims.size() - 1);
}

moving_sphere::moving_sphere(vec3 const centre0, vec3 const centre1, double const time0, double const time1,
                             double radius, std::unique_ptr<material const> mat)
    : centre0(centre0), centre1(centre1), time0(time0), time1(time1), radius(radius), mat(std::move(mat)) {}

// moving_sphere.cpp:3-5
vec3 moving_sphere::centre(double const time) const {
    return centre0 + ((time - time0) / (time1 - time0)) * (centre1 - centre0);
}

// moving_sphere.cpp:51-60: the second box's far corner is taken at the START centre,
// exactly as the reference writes it; the tree the GPU walks must cull what the
// reference's tree culls.
std::optional<aabb> moving_sphere::bounding_box(double const t0, double const t1) const {
    vec3 const rv = vec3(radius, radius, radius), c0 = centre(t0), c1 = centre(t1);
    aabb const box0{c0 - rv, c0 + rv}, box1{c1 - rv, c0 + rv};
    return aabb::surrounding_box(box0, box1);
}

int32_t moving_sphere::flatten_child(scene_flattener& out, aabb const&) const {
    rt_prim_desc p{};
    p.kind = RT_PRIM_MOVING_SPHERE;
    p.material = mat->flatten(out);
    p.centre0[0] = centre0.x(); p.centre0[1] = centre0.y(); p.centre0[2] = centre0.z();
    p.centre1[0] = centre1.x(); p.centre1[1] = centre1.y(); p.centre1[2] = centre1.z();
    p.time0 = time0; p.time1 = time1; p.radius = radius;
    out.prims.push_back(p);
    out.material_of_prim.push_back(mat.get());
    return ~(int32_t)(out.prims.size() - 1);
}

object_list::object_list(std::unique_ptr<object const> object) { objects.push_back(std::move(object)); }
object_list::object_list(std::vector<std::unique_ptr<object const>> objects) : objects(std::move(objects)) {}

// object_list.cpp:28-48
std::optional<aabb> object_list::bounding_box(double t0, double t1) const {
    auto it = objects.begin();
    if (it == objects.end()) return std::nullopt;
    std::optional<aabb> result = (**it).bounding_box(t0, t1);
    if (!result) return std::nullopt;
    while (++it != objects.end()) {
        std::optional<aabb> next = (**it).bounding_box(t0, t1);
        if (!next) return std::nullopt;
        *result = aabb::surrounding_box(*result, *next);
    }
    return result;
}

// A list as the child of a bvh_node (bvh_node.cpp:34-40 accepts any object with a box, and
// object_list.cpp:28-48 gives a list one).  The reference tests the node's box (bvh_node.cpp:53) and
// then scans the list (object_list.cpp:12-26): every member is tried -- a list tests no box of its
// own -- the nearest hit wins, and of equal hits the first member's, because a later one must be
// strictly nearer.  In the tables that is a subtree of nodes that all carry the PARENT's box (testing
// it again decides nothing new) over the members in REVERSE order: a node prefers its right child on
// equal hits (bvh_node.cpp:60-61), so the member written last in the subtree -- the list's first --
// wins among equals, and against the rest of the tree the subtree stands where the list stood.
namespace {
int32_t list_subtree(scene_flattener& out, aabb const& box, std::span<object const* const> members) {
    if (members.size() == 1) return members[0]->flatten_child(out, box);
    int32_t const me = (int32_t)out.nodes.size();
    out.nodes.emplace_back();
    size_t const half = members.size() / 2;
    int32_t const l = list_subtree(out, box, members.subspan(0, half));
    int32_t const r = list_subtree(out, box, members.subspan(half));
    rt_node_desc n{};
    n.box_min[0] = box.min().x(); n.box_min[1] = box.min().y(); n.box_min[2] = box.min().z();
    n.box_max[0] = box.max().x(); n.box_max[1] = box.max().y(); n.box_max[2] = box.max().z();
    n.left = l; n.right = r;
    out.nodes[(size_t)me] = n;
    return me;
}
}  // namespace

int32_t object_list::flatten_child(scene_flattener& out, aabb const& parent_box) const {
    std::vector<object const*> members;
    scan_members(members);
    if (members.empty()) throw std::runtime_error("an empty object_list below a bvh_node has no bounding box");
    std::reverse(members.begin(), members.end());
    return list_subtree(out, parent_box, members);
}

void object_list::scan_members(std::vector<object const*>& out) const {
    for (auto const& o : objects) o->scan_members(out);
}

// A list scans its members in order, each seeing the interval narrowed by the ones
// before (object_list.cpp:15-23); a list inside a list therefore scans like its
// members written out in place.
void object_list::flatten_items(scene_flattener& out) const {
    for (auto const& o : objects) o->flatten_items(out);
}

std::uniform_int_distribution<int> bvh_node::dist{0, 2};

// How many bvh_node objects the reference's constructor creates over n objects
// (bvh_node.cpp:31-41: one over one or two objects, else one plus the two halves).
namespace {
size_t nodes_over(size_t n, std::map<size_t, size_t>& memo) {
    if (n <= 2) return 1;
    auto it = memo.find(n);
    if (it != memo.end()) return it->second;
    size_t const c = 1 + nodes_over(n / 2, memo) + nodes_over(n - n / 2, memo);
    memo[n] = c;
    return c;
}
constexpr size_t kParallelBuildMin = size_t(1) << 14;   // objects under a node worth a thread of its own
constexpr int kParallelBuildDepth = 4;                   // ... down to this depth: at most 16 threads (RTIOW_BUILD_DEPTH=d overrides, 0 = serial)
}  // namespace

// The reference draws one axis per node from the shared generator, a node's own before its left
// subtree's, the whole left subtree before the right (bvh_node.cpp:9,39-40): node k in pre-order
// takes draw k, whatever the objects are.  Drawing all of them up front leaves the generator where
// the reference leaves it and lets the two halves of a large node be built at the same time.
struct bvh_node::build_plan {
    std::vector<uint8_t> axis;            // by pre-order node number
    std::map<size_t, size_t> node_count;  // nodes_over, for the sizes that occur
    double t0, t1;
    int parallel_depth;                   // halves of large nodes are built by threads of their own above this depth
    bool arranged = false;                // the objects already stand in the tree's left-to-right order (see arrange)
};

// A large tree without the sorts.  What a node's sort decides is which objects go left: the n / 2 with the
// smallest keys.  When the largest key of that half is smaller than the smallest of the other, the two sets
// are the same whatever the order the objects arrived in and whatever the algorithm -- a selection
// (std::nth_element, linear) finds them as well as the reference's sort, and the order inside a half is the
// business of the half's own node, by its own axis.  arrange() permutes the objects' keys that way, node by
// node, into the left-to-right order of the tree's leaves, and reports failure the moment a node's split is
// NOT decided by the keys alone (equal keys across the cut, or in a node of two): the build then takes the
// reference's road from the start, sorts and all, since the outcome of an unstable sort over equal keys
// depends on the order of its input, all the way up.  O(n log n) instead of O(n log^2 n) comparisons, one
// bounding_box call per object instead of one per object and level.
namespace {
struct keyed3 { double lo[3]; object const* ptr; };
constexpr size_t kArrangeMin = size_t(1) << 16;   // groups at least this large are arranged (smaller ones: the sorts are cheap)
}  // namespace

bool bvh_node::arrange(build_plan const& plan, void* keys, size_t const n, size_t const index, int const depth) {
    keyed3* const a = static_cast<keyed3*>(keys);
    int const axis = plan.axis[index];
    if (n <= 1) return true;
    if (n == 2) {
        if (a[0].lo[axis] == a[1].lo[axis]) return false;
        if (a[1].lo[axis] < a[0].lo[axis]) std::swap(a[0], a[1]);
        return true;
    }
    size_t const half = n / 2;
    std::nth_element(a, a + half, a + n, [axis](keyed3 const& l, keyed3 const& r) { return l.lo[axis] < r.lo[axis]; });
    double const pivot = a[half].lo[axis];
    for (size_t i = 0; i < half; ++i)
        if (!(a[i].lo[axis] < pivot)) return false;
    size_t const left_index = index + 1, right_index = left_index + (half <= 2 ? 1 : plan.node_count.at(half));
    if (n >= kParallelBuildMin && depth < plan.parallel_depth) {
        bool ok_left = false;
        bool started = false;
        std::thread other;
        try {
            other = std::thread([&] { ok_left = arrange(plan, a, half, left_index, depth + 1); });
            started = true;
        } catch (std::system_error const&) {}
        bool const ok_right = arrange(plan, a + half, n - half, right_index, depth + 1);
        if (started) other.join(); else ok_left = arrange(plan, a, half, left_index, depth + 1);
        return ok_left && ok_right;
    }
    return arrange(plan, a, half, left_index, depth + 1) && arrange(plan, a + half, n - half, right_index, depth + 1);
}

bvh_node::build_plan bvh_node::draw_axes(std::mt19937& mt, size_t n, double t0, double t1) {
    build_plan plan;
    plan.t0 = t0; plan.t1 = t1;
    char const* const depth_env = std::getenv("RTIOW_BUILD_DEPTH");
    plan.parallel_depth = depth_env ? std::atoi(depth_env) : kParallelBuildDepth;
    size_t const nodes = nodes_over(n, plan.node_count);
    plan.axis.resize(nodes);
    for (uint8_t& a : plan.axis) a = (uint8_t)dist(mt);
    return plan;
}

// the plan for `objects`, which are put into the tree's leaf order here when the keys decide every split
bvh_node::build_plan bvh_node::prepare(std::mt19937& mt, std::span<std::unique_ptr<object const>> objects, double const t0, double const t1) {
    build_plan plan = draw_axes(mt, objects.size(), t0, t1);
    char const* const off = std::getenv("RTIOW_BUILD_ARRANGE");     // "0": always the reference's sorts (tests)
    if (objects.size() < kArrangeMin || (off && std::atoi(off) == 0)) return plan;
    std::vector<keyed3> keys(objects.size());
    {   // one bounding_box call per object, shared out over the cores
        int const slices = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<char> unusable((size_t)slices, 0);
        auto slice = [&](int t) {
            for (size_t i = objects.size() * (size_t)t / (size_t)slices; i < objects.size() * (size_t)(t + 1) / (size_t)slices; ++i) {
                object const* const o = objects[i].get();
                std::optional<aabb> const b = o != nullptr ? o->bounding_box(t0, t1) : std::nullopt;
                if (!b) { unusable[(size_t)t] = 1; return; }             // (the reference's road throws the reference's literal)
                vec3 const lo = b->min();
                keys[i] = keyed3{{lo.x(), lo.y(), lo.z()}, o};
                if (std::isnan(lo.x()) || std::isnan(lo.y()) || std::isnan(lo.z())) { unusable[(size_t)t] = 1; return; }
            }
        };
        std::vector<std::thread> threads;
        try {
            for (int t = 1; t < slices; ++t) threads.emplace_back(slice, t);
        } catch (std::system_error const&) {}
        int const started = (int)threads.size() + 1;
        slice(0);
        for (int t = started; t < slices; ++t) slice(t);
        for (std::thread& t : threads) t.join();
        for (char u : unusable) if (u) return plan;
    }
    if (!arrange(plan, keys.data(), keys.size(), 0, 0)) return plan;
    for (auto& o : objects) (void)o.release();   // every pointer is in `keys` now
    for (size_t i = 0; i < keys.size(); ++i) objects[i].reset(keys[i].ptr);
    plan.arranged = true;
    return plan;
}

// bvh_node.cpp:8-50
bvh_node::bvh_node(std::mt19937& mt, std::span<std::unique_ptr<object const>> objects, double const t0, double const t1)
    : bvh_node(prepare(mt, objects, t0, t1), objects, 0, 0) {}

bvh_node::bvh_node(build_plan const& plan, std::span<std::unique_ptr<object const>> objects, size_t const index, int const depth)
    : n_objects(objects.size()) {
    int const axis = plan.axis[index];
    double const t0 = plan.t0, t1 = plan.t1;
    // The reference sorts the pointers with a comparator that asks both objects for their
    // boxes at every comparison (bvh_node.cpp:9-29).  The outcome of a comparison depends
    // only on the two objects' box minima along the axis, and the permutation a sort produces
    // depends only on the outcomes, so sorting (key, pointer) pairs with the same algorithm
    // gives the same order -- ties and all -- at a fraction of the cost for large scenes.
    if (objects.size() > 1 && !plan.arranged) {
        struct keyed { double key; object const* ptr; };
        constexpr size_t kSmall = 16;
        keyed small[kSmall];
        std::vector<keyed> large;
        if (objects.size() > kSmall) large.resize(objects.size());
        std::span<keyed> const order(objects.size() > kSmall ? large.data() : small, objects.size());
        for (size_t i = 0; i < objects.size(); ++i) {
            object const* const o = objects[i].get();
            std::optional<aabb> const b = o != nullptr ? o->bounding_box(t0, t1) : std::nullopt;
            if (!b) throw "no bounding box in bvh_node constructor";
            vec3 const lo = b->min();
            order[i] = keyed{axis == 0 ? lo.x() : (axis == 1 ? lo.y() : lo.z()), o};
        }
        auto const less = [](keyed const& l, keyed const& r) -> bool { return l.key < r.key; };
        // Near the root of a large tree a node has threads to spare (its sibling subtrees take the others).
        // When no two keys are equal, every sorting algorithm must arrive at the same order, so slices sorted
        // in parallel and merged give what std::ranges::sort gives; if the merged order shows a tie (or a
        // NaN), the reference's own algorithm decides it, on the original sequence.
        int const threads = depth < plan.parallel_depth ? (1 << (plan.parallel_depth - depth)) : 1;
        bool sorted = false;
        if (threads > 1 && order.size() >= (size_t(1) << 16)) {
            std::vector<keyed> const original(order.begin(), order.end());
            size_t const n = order.size();
            auto bound = [&](int k) { return n * (size_t)k / (size_t)threads; };
            {
                std::vector<std::future<void>> jobs;
                for (int k = 0; k < threads; ++k)
                    jobs.push_back(std::async(std::launch::async, [&, k] { std::sort(order.begin() + (long)bound(k), order.begin() + (long)bound(k + 1), less); }));
                for (auto& j : jobs) j.get();
            }
            for (int width = 1; width < threads; width *= 2) {
                std::vector<std::future<void>> jobs;
                for (int k = 0; k + width < threads; k += 2 * width)
                    jobs.push_back(std::async(std::launch::async, [&, k, width] {
                        std::inplace_merge(order.begin() + (long)bound(k), order.begin() + (long)bound(k + width),
                                           order.begin() + (long)bound(std::min(k + 2 * width, threads)), less);
                    }));
                for (auto& j : jobs) j.get();
            }
            sorted = true;
            for (size_t i = 0; i + 1 < n && sorted; ++i) sorted = order[i].key < order[i + 1].key;
            if (!sorted) std::copy(original.begin(), original.end(), order.begin());
        }
        if (!sorted) std::ranges::sort(order, less);
        for (auto& o : objects) (void)o.release();   // every pointer is in `order` now
        for (size_t i = 0; i < order.size(); ++i) objects[i].reset(order[i].ptr);
    }

    assert(objects.size() > 0);
    if (objects.size() == 1) {
        own_left = std::move(objects[0]);           // one object, both children (bvh_node.cpp:31-34)
        left = right = own_left.get();
    } else if (objects.size() == 2) {
        own_left = std::move(objects[0]);
        own_right = std::move(objects[1]);
        left = own_left.get(); right = own_right.get();
    } else {
        size_t const half = objects.size() / 2;
        size_t const left_index = index + 1, right_index = left_index + (half <= 2 ? 1 : plan.node_count.at(half));
        auto build = [&plan, depth](std::span<std::unique_ptr<object const>> part, size_t at) {
            return std::unique_ptr<bvh_node const>(new bvh_node(plan, part, at, depth + 1));
        };
        if (objects.size() >= kParallelBuildMin && depth < plan.parallel_depth) {
            // (an exception from either half -- the reference's string literal -- comes out of here
            // after both have ended)
            auto other = std::async(std::launch::async, build, objects.subspan(0, half), left_index);
            std::unique_ptr<bvh_node const> r;
            try {
                r = build(objects.subspan(half), right_index);
            } catch (...) {
                other.wait();
                throw;
            }
            own_left = other.get();
            own_right = std::move(r);
        } else {
            own_left = build(objects.subspan(0, half), left_index);
            own_right = build(objects.subspan(half), right_index);
        }
        left = own_left.get(); right = own_right.get();
    }

    std::optional<aabb> const box_left = left->bounding_box(t0, t1), box_right = right->bounding_box(t0, t1);
    if (!box_left || !box_right) throw "no bounding box in bvh_node constructor";
    box = aabb::surrounding_box(*box_left, *box_right);
}

std::optional<aabb> bvh_node::bounding_box(double, double) const { return box; }

// Nodes are numbered in pre-order: a node, its whole left subtree, its right subtree.
int32_t bvh_node::flatten_child(scene_flattener& out, aabb const&) const {
    // room for the whole subtree at once: growing the tables of a 10^6-sphere world by doubling
    // copies them several times over (a sphere brings one primitive, one material and at most a
    // few textures; only the root of a large tree finds the capacity short)
    if (n_objects >= 4096 && out.prims.capacity() < out.prims.size() + n_objects) {
        out.nodes.reserve(out.nodes.size() + n_objects);
        out.prims.reserve(out.prims.size() + n_objects);
        out.material_of_prim.reserve(out.material_of_prim.size() + n_objects);
        out.materials.reserve(out.materials.size() + n_objects);
        out.textures.reserve(out.textures.size() + n_objects);
    }
    int32_t const me = (int32_t)out.nodes.size();
    out.nodes.emplace_back();
    int32_t l, r;
    // The two halves of a large node flatten side by side: the right one into tables of its own, which are
    // then appended with their references moved along.  Entry for entry what one thread writes (the tables
    // list a subtree's entries in the order of a left-to-right walk either way).
    bool split = false;
    static std::atomic<int> helpers{0};   // threads flattening right halves at the moment, over all trees: at most 15
    struct helper_slot {
        bool taken;
        helper_slot() : taken(helpers.fetch_add(1) < 15) { if (!taken) helpers.fetch_sub(1); }
        ~helper_slot() { if (taken) helpers.fetch_sub(1); }
    };
    if (n_objects >= kParallelFlatten && right != left) {
        helper_slot const slot;
        scene_flattener local;
        int32_t r_local = 0;
        std::exception_ptr failed;
        try {
            if (!slot.taken) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
            std::thread other([&] {
                try { r_local = right->flatten_child(local, *box); } catch (...) { failed = std::current_exception(); }
            });
            split = true;
            try { l = left->flatten_child(out, *box); } catch (...) { other.join(); throw; }
            other.join();
        } catch (std::system_error const&) {
            if (split) throw;     // (it came out of a subtree, not out of starting the thread)
        }
        if (split) {
            if (failed) std::rethrow_exception(failed);
            r = out.append(local, r_local);
        }
    }
    if (!split) {
        l = left->flatten_child(out, *box);
        r = (right == left) ? l : right->flatten_child(out, *box);
    }
    rt_node_desc n{};
    n.box_min[0] = box->min().x(); n.box_min[1] = box->min().y(); n.box_min[2] = box->min().z();
    n.box_max[0] = box->max().x(); n.box_max[1] = box->max().y(); n.box_max[2] = box->max().z();
    n.left = l; n.right = r;
    out.nodes[(size_t)me] = n;
    return me;
}

// ---------------------------------------------------------------------- camera ----
std::uniform_real_distribution<double> camera::dist{-1, 1};

namespace {
// camera.cpp:6-14
double half_height(double const vfov) {
    double const theta = vfov * M_PI / 180;
    return std::tan(theta / 2);
}
double half_width(double const vfov, double const aspect) { return aspect * half_height(vfov); }
}  // namespace

// camera.cpp:25-36
camera::camera(std::mt19937& mt, vec3 const from, vec3 const at, vec3 const up, double const vfov, double const aspect,
               double const aperture, double const focus_dist, double const t0, double const t1)
    : origin{from},
      w{(from - at).unit_vector()},
      u{up.cross(w).unit_vector()},
      v{w.cross(u)},
      lower_left_corner{from - half_width(vfov, aspect) * focus_dist * u - half_height(vfov) * focus_dist * v - focus_dist * w},
      horizontal{2 * half_width(vfov, aspect) * focus_dist * u},
      vertical{2 * half_height(vfov) * focus_dist * v},
      lens_radius{aperture / 2},
      t0(t0),
      t1(t1),
      mt(mt) {}

rt_camera camera::describe() const {
    rt_camera c{};
    auto put = [](double* dst, vec3 const& s) { dst[0] = s.x(); dst[1] = s.y(); dst[2] = s.z(); };
    put(c.origin, origin); put(c.u, u); put(c.v, v); put(c.w, w);
    put(c.lower_left_corner, lower_left_corner); put(c.horizontal, horizontal); put(c.vertical, vertical);
    c.lens_radius = lens_radius; c.time0 = t0; c.time1 = t1;
    return c;
}

ray camera::get_ray(double const s, double const t) const {
    rt_camera const c = describe();
    double const st[2] = {s, t};
    for (uint32_t k = 64;; k *= 4) {
        std::vector<double> const tape = peek_uniforms(mt, k);
        rt_ray out;
        int32_t used = 0;
        int const rc = rt_camera_batch(probe_device(), &c, st, tape.data(), k, 1, &out, &used);
        if (rc == RT_E_TAPE && k < (1u << 20)) continue;
        if (rc) device_failure("camera::get_ray");
        advance(mt, used);
        return from_abi(out);
    }
}

// supersampler.cpp:3-19
supersampler::supersampler() : count(0), r(0), g(0), b(0) {}

void supersampler::add_sample(colour const colour) {
    ++count;
    r += colour.red(); g += colour.green(); b += colour.blue();
}

colour supersampler::value() const {
    if (count == 0) return colour{0};
    return {r / count, g / count, b / count};
}

// ----------------------------------------------------------------------- frame ----
frame render(device_scene const& scene, camera const& cam, render_options const& opt) {
    frame f;
    f.nx = opt.nx; f.ny = opt.ny; f.ns = opt.ns;
    f.sums.assign((size_t)opt.nx * opt.ny * 3, 0.0f);
    rt_camera const c = cam.describe();
    rt_render_params p{};
    p.nx = (uint32_t)opt.nx; p.ny = (uint32_t)opt.ny;
    p.sample_begin = 0; p.sample_end = (uint32_t)opt.ns;
    p.max_depth = (uint32_t)opt.max_depth;
    p.accum_type = RT_ACCUM_F32;
    p.t_min = opt.t_min;
    p.seed = opt.seed;
    if (rt_render_host(scene.get(), &c, &p, f.sums.data(), &f.stats)) device_failure("render");
    return f;
}

frame render(object const& world, camera const& cam, render_options const& opt) {
    if (!opt.devices.empty()) return render(upload(world, opt.devices), cam, opt);
    device_scene scene(world, opt.device);
    return render(scene, cam, opt);
}

std::vector<std::unique_ptr<device_scene>> upload(object const& world, std::vector<int> const& devices) {
    std::vector<std::unique_ptr<device_scene>> scenes(devices.size());
    std::vector<std::exception_ptr> errors(devices.size());
    std::vector<char const*> thrown(devices.size(), nullptr);     // (the scene API throws string literals, as the reference does)
    auto one = [&](size_t k) {
        try {
            scenes[k] = std::make_unique<device_scene>(world, devices[k]);
        } catch (char const* e) {
            thrown[k] = e;
        } catch (...) {
            errors[k] = std::current_exception();
        }
    };
    std::vector<std::thread> threads;
    for (size_t k = 1; k < devices.size(); ++k) threads.emplace_back(one, k);
    if (!devices.empty()) one(0);
    for (std::thread& t : threads) t.join();
    for (size_t k = 0; k < devices.size(); ++k) {
        if (thrown[k]) throw thrown[k];
        if (errors[k]) std::rethrow_exception(errors[k]);
    }
    return scenes;
}

frame render(std::vector<std::unique_ptr<device_scene>> const& scenes, camera const& cam, render_options const& opt) {
    frame f;
    f.nx = opt.nx; f.ny = opt.ny; f.ns = opt.ns;
    f.sums.assign((size_t)opt.nx * opt.ny * 3, 0.0f);
    rt_camera const c = cam.describe();
    rt_render_params p{};
    p.nx = (uint32_t)opt.nx; p.ny = (uint32_t)opt.ny;
    p.sample_begin = 0; p.sample_end = (uint32_t)opt.ns;
    p.max_depth = (uint32_t)opt.max_depth;
    p.accum_type = RT_ACCUM_F32;
    p.t_min = opt.t_min;
    p.seed = opt.seed;
    std::vector<rt_scene*> handles;
    for (auto const& s : scenes) handles.push_back(s ? s->get() : nullptr);
    if (rt_render_multi(handles.data(), (uint32_t)handles.size(), &c, &p, f.sums.data(), &f.stats)) device_failure("render");
    return f;
}

colour frame::pixel(int i, int row) const {
    float const* s = &sums[((size_t)row * nx + i) * 3];
    if (ns == 0) return colour{0};
    return colour{(double)s[0] / ns, (double)s[1] / ns, (double)s[2] / ns}.gamma2();
}

// main.cpp:26-28,145-150
void frame::write_ppm(std::FILE* out) const {
    std::fprintf(out, "P3\n%d %d\n255\n", nx, ny);
    for (int row = 0; row < ny; ++row)
        for (int i = 0; i < nx; ++i) {
            colour const c = pixel(i, row);
            std::fprintf(out, "%d %d %d\n", int(255.99 * c.red()), int(255.99 * c.green()), int(255.99 * c.blue()));
        }
}

// The same bytes as raw P6: at 3840x2160 the P3 text is ~100 MB, the binary file 25 MB.
void frame::write_ppm_binary(std::FILE* out) const {
    std::fprintf(out, "P6\n%d %d\n255\n", nx, ny);
    std::vector<unsigned char> row((size_t)nx * 3);
    for (int r = 0; r < ny; ++r) {
        for (int i = 0; i < nx; ++i) {
            colour const c = pixel(i, r);
            row[(size_t)i * 3] = (unsigned char)int(255.99 * c.red());
            row[(size_t)i * 3 + 1] = (unsigned char)int(255.99 * c.green());
            row[(size_t)i * 3 + 2] = (unsigned char)int(255.99 * c.blue());
        }
        std::fwrite(row.data(), 1, row.size(), out);
    }
}

// The same bytes as an 8-bit RGB PNG.  The image data is a zlib stream of stored
// (uncompressed) deflate blocks, so nothing beyond the C library is needed and the file is as
// large as the P6 one; every viewer reads it.
void frame::write_png(std::FILE* out) const {
    static uint32_t crc_table[256];
    if (!crc_table[1])
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[n] = c;
        }
    auto be32 = [](unsigned char* p, uint32_t v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; };
    auto chunk = [&](char const* type, std::vector<unsigned char> const& data) {
        unsigned char head[8], tail[4];
        be32(head, (uint32_t)data.size());
        std::memcpy(head + 4, type, 4);
        uint32_t c = 0xFFFFFFFFu;
        for (int i = 4; i < 8; ++i) c = crc_table[(c ^ head[i]) & 0xFF] ^ (c >> 8);
        for (unsigned char b : data) c = crc_table[(c ^ b) & 0xFF] ^ (c >> 8);
        be32(tail, c ^ 0xFFFFFFFFu);
        std::fwrite(head, 1, 8, out);
        if (!data.empty()) std::fwrite(data.data(), 1, data.size(), out);
        std::fwrite(tail, 1, 4, out);
    };
    static unsigned char const signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::fwrite(signature, 1, 8, out);
    std::vector<unsigned char> ihdr(13, 0);
    be32(ihdr.data(), (uint32_t)nx);
    be32(ihdr.data() + 4, (uint32_t)ny);
    ihdr[8] = 8;   // bits per channel
    ihdr[9] = 2;   // RGB
    chunk("IHDR", ihdr);

    // scanlines, each led by filter type 0
    size_t const stride = (size_t)nx * 3 + 1;
    std::vector<unsigned char> raw(stride * (size_t)ny);
    for (int r = 0; r < ny; ++r) {
        unsigned char* row = raw.data() + stride * (size_t)r;
        row[0] = 0;
        for (int i = 0; i < nx; ++i) {
            colour const c = pixel(i, r);
            row[1 + (size_t)i * 3] = (unsigned char)int(255.99 * c.red());
            row[2 + (size_t)i * 3] = (unsigned char)int(255.99 * c.green());
            row[3 + (size_t)i * 3] = (unsigned char)int(255.99 * c.blue());
        }
    }
    uint32_t a = 1, b = 0;   // Adler-32 of the scanlines
    for (size_t i = 0; i < raw.size();) {
        size_t const n = std::min<size_t>(raw.size() - i, 5552);
        for (size_t k = 0; k < n; ++k) { a += raw[i + k]; b += a; }
        a %= 65521u; b %= 65521u;
        i += n;
    }
    std::vector<unsigned char> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    size_t at = 0;
    do {
        size_t const n = std::min<size_t>(raw.size() - at, 65535);
        z.push_back(at + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)at, raw.begin() + (long)(at + n));
        at += n;
    } while (at < raw.size());
    unsigned char adler[4];
    be32(adler, (b << 16) | a);
    z.insert(z.end(), adler, adler + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
}

// ------------------------------------------------------------- scene builders ----
// main.cpp:52-59
object_list two_perlin_spheres(std::mt19937& mt) {
    std::vector<std::unique_ptr<object const>> objects;
    objects.push_back(std::make_unique<sphere const>(vec3{0, -1000, 0}, 1000, std::make_unique<lambertian const>(mt, std::make_unique<perlin_texture const>())));
    objects.push_back(std::make_unique<sphere const>(vec3{0, 2, 0}, 2, std::make_unique<lambertian const>(mt, std::make_unique<perlin_texture const>())));
    return object_list(std::move(objects));
}

// main.cpp:61-99, with every draw bound to a named value so the order is the textual one
bvh_node build_world(std::mt19937& mt) {
    std::vector<std::unique_ptr<object const>> objects;
    objects.reserve(500);
    auto chequer = std::make_unique<chequer_texture const>(std::make_unique<constant_texture const>(colour{0.2, 0.3, 0.1}),
                                                           std::make_unique<constant_texture const>(colour{0.9, 0.9, 0.9}));
    objects.push_back(std::make_unique<sphere const>(vec3{0, -1000, -1}, 1000, std::make_unique<lambertian const>(mt, std::move(chequer))));
    objects.push_back(std::make_unique<sphere const>(vec3{0, 1, 0}, 1, std::make_unique<dielectric const>(mt, 1.5)));
    objects.push_back(std::make_unique<sphere const>(vec3{-4, 1, 0}, 1, std::make_unique<lambertian const>(mt, std::make_unique<constant_texture const>(colour{0.4, 0.2, 0.1}))));
    objects.push_back(std::make_unique<sphere const>(vec3{4, 1, 0}, 1, std::make_unique<metal const>(mt, vec3{0.7, 0.6, 0.5}, 0)));

    std::uniform_real_distribution<double> dist{0, 1};
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            double const cx = a + 0.9 * dist(mt);
            double const cz = b + 0.9 * dist(mt);
            vec3 const centre{cx, 0.2, cz};
            if ((centre - vec3{4, 0.2, 0}).length() <= 0.9) continue;
            double const choose_mat = dist(mt);
            if (choose_mat < 0.8) {
                double const lift = 0.5 * dist(mt);
                vec3 const offset{0, lift, 0};
                double const r1 = dist(mt), r2 = dist(mt), g1 = dist(mt), g2 = dist(mt), b1 = dist(mt), b2 = dist(mt);
                colour const c{r1 * r2, g1 * g2, b1 * b2};
                objects.push_back(std::make_unique<moving_sphere const>(centre, centre + offset, 0, 1, 0.2,
                                                                        std::make_unique<lambertian const>(mt, std::make_unique<constant_texture const>(c))));
            } else if (choose_mat < 0.95) {
                double const ar = 0.5 * (1 + dist(mt)), ag = 0.5 * (1 + dist(mt)), ab = 0.5 * (1 + dist(mt));
                double const fuzz = 0.5 * dist(mt);
                objects.push_back(std::make_unique<sphere const>(centre, 0.2, std::make_unique<metal const>(mt, vec3{ar, ag, ab}, fuzz)));
            } else {
                objects.push_back(std::make_unique<sphere const>(centre, 0.2, std::make_unique<dielectric const>(mt, 1.5)));
            }
        }
    }
    return bvh_node(mt, objects, 0, 1);
}

}  // namespace rtiow
