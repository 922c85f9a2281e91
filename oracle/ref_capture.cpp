// ref_capture.cpp -- drives the UNMODIFIED reference classes (compiled from
// /root/reference/c++/*.cpp, main.cpp excluded, by oracle/Makefile) to produce the
// golden vectors under tests/golden/ and to time the reference renderer.
//
// TEST INFRASTRUCTURE ONLY.  Nothing here is product code; the binary is written to
// oracle/_ref/ and is only ever executed by oracle/make_goldens.py, tests and
// bench.py's cpu_baseline leg.
//
// The reference's main() cannot be called (fixed scene, random_device seed, PPM to
// stdout), and ray_colour lives in main.cpp's anonymous namespace, so the pixel loop
// and the recursion are driven from here through the reference's public interfaces
// (object::hit, material::scatter, camera::get_ray, supersampler), in the order
// main.cpp:30-50 and main.cpp:126-141 use them.
//
// All files are raw little-endian doubles; the record layouts are given per command.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <optional>
#include <random>
#include <span>
#include <string>
#include <vector>

// The BVH's children and box are private (bvh_node.hpp:14-15).  Every standard header
// the reference headers need is already included above, so turning `class` into
// `struct` for the reference headers only changes the default member access; layout
// and symbol names are unaffected.
#define class struct
#include "aabb.hpp"
#include "bvh_node.hpp"
#include "camera.hpp"
#include "colour.hpp"
#include "material.hpp"
#include "moving_sphere.hpp"
#include "object.hpp"
#include "object_list.hpp"
#include "perlin.hpp"
#include "ray.hpp"
#include "sphere.hpp"
#include "supersampler.hpp"
#include "texture.hpp"
#include "vec3.hpp"
#undef class

extern "C" {
#include "rt_oracle.h"  // recipe parser only (orc_scene_load); no oracle arithmetic is used here
}

namespace {

std::vector<double> read_doubles(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<double> v((size_t)sz / sizeof(double));
    if (!v.empty() && fread(v.data(), sizeof(double), v.size(), f) != v.size()) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

void write_doubles(const char *path, const std::vector<double> &v) {
    FILE *f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    if (!v.empty() && fwrite(v.data(), sizeof(double), v.size(), f) != v.size()) { fprintf(stderr, "short write %s\n", path); exit(2); }
    fclose(f);
}

vec3 V(ovec3 v) { return vec3{v.x, v.y, v.z}; }

struct built_scene {
    std::unique_ptr<object const> world;             // object_list over the recipe's items
    std::map<material const *, int> prim_of_material;
    std::vector<bvh_node const *> bvh_roots;         // one per bvh group, recipe order
    oscene *recipe = nullptr;
};

std::unique_ptr<texture const> make_texture(const oscene *s, int id) {
    const otexture &t = s->tex[id];
    switch (t.kind) {
    case ORC_TEX_CONSTANT: return std::make_unique<constant_texture const>(colour{t.c[0], t.c[1], t.c[2]});
    case ORC_TEX_CHEQUER: return std::make_unique<chequer_texture const>(make_texture(s, t.even), make_texture(s, t.odd));
    default: return std::make_unique<perlin_texture const>();
    }
}

std::unique_ptr<material const> make_material(const oscene *s, int id, std::mt19937 &mt) {
    const omaterial &m = s->mat[id];
    switch (m.kind) {
    case ORC_MAT_LAMBERTIAN: return std::make_unique<lambertian const>(mt, make_texture(s, m.tex));
    case ORC_MAT_METAL: return std::make_unique<metal const>(mt, vec3{m.albedo[0], m.albedo[1], m.albedo[2]}, m.fuzz);
    default: return std::make_unique<dielectric const>(mt, m.ri);
    }
}

std::unique_ptr<object const> make_prim(built_scene &b, int id, std::mt19937 &mt) {
    const oprim &p = b.recipe->prim[id];
    auto mat = make_material(b.recipe, p.mat, mt);
    b.prim_of_material[mat.get()] = id;
    if (p.kind == ORC_PRIM_SPHERE) return std::make_unique<sphere const>(V(p.c0), p.r, std::move(mat));
    return std::make_unique<moving_sphere const>(V(p.c0), V(p.c1), p.t0, p.t1, p.r, std::move(mat));
}

// mt is the generator handed to the materials (and later to the camera and the pixel
// loop); every bvh group is built with its own std::mt19937(seed) as the recipe says.
built_scene build(const char *recipe_path, std::mt19937 &mt) {
    built_scene b;
    char err[512];
    b.recipe = orc_scene_load(recipe_path, err, sizeof err);
    if (!b.recipe) { fprintf(stderr, "%s\n", err); exit(2); }
    std::vector<std::unique_ptr<object const>> items;
    for (int i = 0; i < b.recipe->n_item; ++i) {
        const oitem &it = b.recipe->item[i];
        if (!it.is_bvh) {
            items.push_back(make_prim(b, it.first_prim, mt));
        } else {
            // the group's objects: its primitives, the members of a list group as one object_list
            std::vector<std::unique_ptr<object const>> objs;
            for (int k = 0; k < it.n_prims;) {
                const int list = b.recipe->prim[it.first_prim + k].list;
                if (list < 0) { objs.push_back(make_prim(b, it.first_prim + k, mt)); ++k; continue; }
                std::vector<std::unique_ptr<object const>> members;
                for (int m = 0; m < b.recipe->list[list].n; ++m) members.push_back(make_prim(b, b.recipe->list[list].first + m, mt));
                objs.push_back(std::make_unique<object_list const>(std::move(members)));
                k += b.recipe->list[list].n;
            }
            std::mt19937 bm(it.seed);
            auto node = std::make_unique<bvh_node const>(bm, objs, it.t0, it.t1);
            b.bvh_roots.push_back(node.get());
            items.push_back(std::move(node));
        }
    }
    b.world = std::make_unique<object_list const>(std::move(items));
    return b;
}

// ---- topology dump ---------------------------------------------------------------
// Pre-order: each node writes box (min xyz, max xyz), left, right; a child that is a
// node is given its pre-order index, a primitive child is ~prim.
struct topo { std::vector<double> rec; int count = 0; };

int prim_id(const built_scene &b, object const *o) {
    if (auto s = dynamic_cast<sphere const *>(o)) return b.prim_of_material.at(s->mat.get());
    if (auto m = dynamic_cast<moving_sphere const *>(o)) return b.prim_of_material.at(m->mat.get());
    fprintf(stderr, "unexpected leaf type\n");
    exit(2);
}

int dump_node(const built_scene &b, bvh_node const *n, topo &t) {
    int me = t.count++;
    t.rec.resize((size_t)t.count * 8);
    auto child = [&](object const *c) -> int {
        if (auto bn = dynamic_cast<bvh_node const *>(c)) return dump_node(b, bn, t);
        return ~prim_id(b, c);
    };
    int l = child(n->left.get());
    int r = (n->right.get() == n->left.get()) ? l : child(n->right.get());
    double *rec = &t.rec[(size_t)me * 8];
    rec[0] = n->box->min().x(); rec[1] = n->box->min().y(); rec[2] = n->box->min().z();
    rec[3] = n->box->max().x(); rec[4] = n->box->max().y(); rec[5] = n->box->max().z();
    rec[6] = l; rec[7] = r;
    return me;
}

// ---- integrator over the reference classes (main.cpp:30-50) ----------------------
struct harvest {
    std::vector<double> *rays = nullptr;   // 7 per ray
    std::vector<double> *hits = nullptr;   // 9 per ray: hit prim t p n
    size_t max_rays = 0;
    const built_scene *b = nullptr;
};

colour ray_colour_ref(ray const &r, object const &world, int depth, harvest *hv) {
    auto const result = world.hit(r, 0.001, std::numeric_limits<double>::max());
    if (hv && hv->rays->size() / 7 < hv->max_rays) {
        auto &R = *hv->rays;
        R.insert(R.end(), {r.origin().x(), r.origin().y(), r.origin().z(), r.direction().x(), r.direction().y(), r.direction().z(), r.time()});
        auto &H = *hv->hits;
        if (result) H.insert(H.end(), {1.0, (double)hv->b->prim_of_material.at(result->material), result->t, result->p.x(), result->p.y(), result->p.z(), result->normal.x(), result->normal.y(), result->normal.z()});
        else H.insert(H.end(), {0.0, -1.0, 0, 0, 0, 0, 0, 0, 0});
    }
    if (result.has_value()) {
        std::optional<scatter_record> sc = result->material->scatter(r, *result);
        if (depth < 50 && sc.has_value()) {
            auto const next = ray_colour_ref(sc->scattered, world, depth + 1, hv);
            return {next.red() * sc->attenuation.x(), next.green() * sc->attenuation.y(), next.blue() * sc->attenuation.z()};
        }
        return colour{0};
    }
    auto const unit_direction = r.direction().unit_vector();
    auto const t = 0.5 * (unit_direction.y() + 1.0);
    colour const white{1.0}, blue{0.5, 0.7, 1.0};
    return (1.0 - t) * white + t * blue;
}

camera make_camera(std::mt19937 &mt, const double *a) {
    return camera{mt, vec3{a[0], a[1], a[2]}, vec3{a[3], a[4], a[5]}, vec3{a[6], a[7], a[8]}, a[9], a[10], a[11], a[12], a[13], a[14]};
}

// the pixel loop of main.cpp:126-141; returns per-pixel means, top row first
double render_ref(const built_scene &b, std::mt19937 &mt, const double *cam, int nx, int ny, int ns, std::vector<double> &out, harvest *hv) {
    std::uniform_real_distribution<double> dist{0, 1};
    camera c = make_camera(mt, cam);
    out.clear();
    out.reserve((size_t)nx * ny * 3);
    auto const start = std::chrono::steady_clock::now();
    for (int j = ny - 1; j >= 0; --j) {
        for (int i = 0; i < nx; ++i) {
            supersampler ss;
            for (int s = 0; s < ns; ++s) {
                double const ii = i + dist(mt), jj = j + dist(mt), u = ii / double(nx), v = jj / double(ny);
                ray const r = c.get_ray(u, v);
                ss.add_sample(ray_colour_ref(r, *b.world, 0, hv));
            }
            colour const m = ss.value();
            out.push_back(m.red()); out.push_back(m.green()); out.push_back(m.blue());
        }
    }
    auto const end = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(end - start).count();
}

// uniforms consumed between two generator states, as libstdc++ canonical doubles
int replay_uniforms(std::mt19937 before, const std::mt19937 &after, double *out, int kmax) {
    int n = 0;
    while (!(before == after)) {
        double u = std::generate_canonical<double, 53>(before);
        if (n < kmax) out[n] = u;
        ++n;
        if (n > 100000) { fprintf(stderr, "replay runaway\n"); exit(2); }
    }
    return n;
}

void parse_cam(char **argv, double cam[15]) { for (int k = 0; k < 15; ++k) cam[k] = atof(argv[k]); }

int usage() {
    fprintf(stderr,
            "ref_capture aabb CASES OUT                      cases[n][14] -> hit[n]\n"
            "ref_capture trace RECIPE RAYS TMIN TMAX OUT     rays[n][7] -> [n][9] hit prim t p n\n"
            "ref_capture topology RECIPE OUT                 every bvh group, [nodes][8] box l r\n"
            "ref_capture texture RECIPE TEX POINTS OUT       p[n][3] -> rgb[n][3]\n"
            "ref_capture scatter RECIPE SEED K IN OUT        in[n][15] prim ray t p n -> [n][12+K]\n"
            "ref_capture camera SEED K CAM(15) ST OUT        st[n][2] -> [n][8+K]\n"
            "ref_capture render RECIPE SEED NX NY NS CAM(15) OUT [RAYS HITS MAXRAYS]\n");
    return 2;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) return usage();
    std::string cmd = argv[1];
    if (cmd == "aabb" && argc == 4) {
        auto in = read_doubles(argv[2]);
        size_t n = in.size() / 14;
        std::vector<double> out(n);
        for (size_t i = 0; i < n; ++i) {
            const double *c = &in[i * 14];
            aabb box{vec3{c[0], c[1], c[2]}, vec3{c[3], c[4], c[5]}};
            ray r{vec3{c[6], c[7], c[8]}, vec3{c[9], c[10], c[11]}, 0.0};
            out[i] = box.hit(r, c[12], c[13]) ? 1.0 : 0.0;
        }
        write_doubles(argv[3], out);
        return 0;
    }
    if (cmd == "trace" && argc == 7) {
        std::mt19937 mt(1);
        built_scene b = build(argv[2], mt);
        auto in = read_doubles(argv[3]);
        double tmin = atof(argv[4]), tmax = atof(argv[5]);
        if (std::string(argv[5]) == "max") tmax = std::numeric_limits<double>::max();
        size_t n = in.size() / 7;
        std::vector<double> out(n * 9, 0.0);
        for (size_t i = 0; i < n; ++i) {
            const double *a = &in[i * 7];
            ray r{vec3{a[0], a[1], a[2]}, vec3{a[3], a[4], a[5]}, a[6]};
            auto h = b.world->hit(r, tmin, tmax);
            double *o = &out[i * 9];
            if (h) {
                o[0] = 1; o[1] = b.prim_of_material.at(h->material); o[2] = h->t;
                o[3] = h->p.x(); o[4] = h->p.y(); o[5] = h->p.z();
                o[6] = h->normal.x(); o[7] = h->normal.y(); o[8] = h->normal.z();
            } else {
                o[1] = -1;
            }
        }
        write_doubles(argv[6], out);
        return 0;
    }
    if (cmd == "topology" && argc == 4) {
        std::mt19937 mt(1);
        built_scene b = build(argv[2], mt);
        topo t;
        for (auto root : b.bvh_roots) dump_node(b, root, t);
        write_doubles(argv[3], t.rec);
        return 0;
    }
    if (cmd == "texture" && argc == 6) {
        std::mt19937 mt(1);
        built_scene b = build(argv[2], mt);
        auto tex = make_texture(b.recipe, atoi(argv[3]));
        auto in = read_doubles(argv[4]);
        size_t n = in.size() / 3;
        std::vector<double> out(n * 3);
        for (size_t i = 0; i < n; ++i) {
            colour c = tex->value(0, 0, vec3{in[i * 3], in[i * 3 + 1], in[i * 3 + 2]});
            out[i * 3] = c.red(); out[i * 3 + 1] = c.green(); out[i * 3 + 2] = c.blue();
        }
        write_doubles(argv[5], out);
        return 0;
    }
    if (cmd == "scatter" && argc == 7) {
        std::mt19937 mt((uint32_t)strtoul(argv[3], nullptr, 10));
        built_scene b = build(argv[2], mt);
        int K = atoi(argv[4]);
        auto in = read_doubles(argv[5]);
        size_t n = in.size() / 15, rec = 12 + (size_t)K;
        std::vector<double> out(n * rec, 0.0);
        // material objects by primitive
        std::map<int, material const *> mat_of_prim;
        for (auto &kv : b.prim_of_material) mat_of_prim[kv.second] = kv.first;
        for (size_t i = 0; i < n; ++i) {
            const double *a = &in[i * 15];
            material const *m = mat_of_prim.at((int)a[0]);
            ray r{vec3{a[1], a[2], a[3]}, vec3{a[4], a[5], a[6]}, a[7]};
            hit_record h{a[8], vec3{a[9], a[10], a[11]}, vec3{a[12], a[13], a[14]}, m};
            std::mt19937 before = mt;
            auto sc = m->scatter(r, h);
            double *o = &out[i * rec];
            o[11] = replay_uniforms(before, mt, o + 12, K);
            if (sc) {
                o[0] = 1;
                o[1] = sc->attenuation.x(); o[2] = sc->attenuation.y(); o[3] = sc->attenuation.z();
                o[4] = sc->scattered.origin().x(); o[5] = sc->scattered.origin().y(); o[6] = sc->scattered.origin().z();
                o[7] = sc->scattered.direction().x(); o[8] = sc->scattered.direction().y(); o[9] = sc->scattered.direction().z();
                o[10] = sc->scattered.time();
            }
        }
        write_doubles(argv[6], out);
        return 0;
    }
    if (cmd == "camera" && argc == 21) {
        std::mt19937 mt((uint32_t)strtoul(argv[2], nullptr, 10));
        int K = atoi(argv[3]);
        double cam[15];
        parse_cam(argv + 4, cam);
        camera c = make_camera(mt, cam);
        auto in = read_doubles(argv[19]);
        size_t n = in.size() / 2, rec = 8 + (size_t)K;
        std::vector<double> out(n * rec, 0.0);
        for (size_t i = 0; i < n; ++i) {
            std::mt19937 before = mt;
            ray r = c.get_ray(in[i * 2], in[i * 2 + 1]);
            double *o = &out[i * rec];
            o[0] = r.origin().x(); o[1] = r.origin().y(); o[2] = r.origin().z();
            o[3] = r.direction().x(); o[4] = r.direction().y(); o[5] = r.direction().z();
            o[6] = r.time();
            o[7] = replay_uniforms(before, mt, o + 8, K);
        }
        write_doubles(argv[20], out);
        return 0;
    }
    if (cmd == "render" && (argc == 23 || argc == 26)) {
        std::mt19937 mt((uint32_t)strtoul(argv[3], nullptr, 10));
        built_scene b = build(argv[2], mt);
        int nx = atoi(argv[4]), ny = atoi(argv[5]), ns = atoi(argv[6]);
        double cam[15];
        parse_cam(argv + 7, cam);
        std::vector<double> out, rays, hits;
        harvest hv;
        if (argc == 26) { hv.rays = &rays; hv.hits = &hits; hv.max_rays = (size_t)atol(argv[25]); hv.b = &b; }
        double secs = render_ref(b, mt, cam, nx, ny, ns, out, argc == 26 ? &hv : nullptr);
        if (std::string(argv[22]) != "-") write_doubles(argv[22], out);
        if (argc == 26) { write_doubles(argv[23], rays); write_doubles(argv[24], hits); }
        printf("{\"render_seconds\": %.6f, \"paths\": %ld, \"mpaths_per_s\": %.6f}\n", secs, (long)nx * ny * ns,
               (double)nx * ny * ns / secs / 1e6);
        return 0;
    }
    return usage();
}
