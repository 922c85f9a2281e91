// recipe.cpp -- scene recipes: text I/O, instantiation through the scene API's public
// constructors, and the generators for the benchmark and parity scenes.
#include "recipe.hpp"

#include <chrono>

#include <system_error>

#include <algorithm>

#include <exception>

#include <thread>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace rtiow {

int recipe::add_constant(double r, double g, double b) {
    textures.push_back({RT_TEX_CONSTANT, {r, g, b}, 0, 0});
    return (int)textures.size() - 1;
}
int recipe::add_lambertian(int t) {
    materials.push_back({RT_MAT_LAMBERTIAN, t, {0, 0, 0}, 0, 0});
    return (int)materials.size() - 1;
}
int recipe::add_metal(double r, double g, double b, double fuzz) {
    materials.push_back({RT_MAT_METAL, 0, {r, g, b}, fuzz, 0});
    return (int)materials.size() - 1;
}
int recipe::add_dielectric(double ri) {
    materials.push_back({RT_MAT_DIELECTRIC, 0, {0, 0, 0}, 0, ri});
    return (int)materials.size() - 1;
}
void recipe::add_sphere(int m, double x, double y, double z, double r) {
    prims.push_back({RT_PRIM_SPHERE, m, {x, y, z}, {x, y, z}, 0, 1, r, -1});
}
void recipe::add_moving_sphere(int m, double const c0[3], double const c1[3], double t0, double t1, double r) {
    prims.push_back({RT_PRIM_MOVING_SPHERE, m, {c0[0], c0[1], c0[2]}, {c1[0], c1[1], c1[2]}, t0, t1, r, -1});
}

void recipe::save(std::string const& path) const {
    std::FILE* f = std::fopen(path.c_str(), "w");
    if (!f) throw std::runtime_error("cannot write " + path);
    std::fprintf(f, "rtiow-recipe 1\n");
    for (tex const& t : textures) {
        if (t.kind == RT_TEX_CONSTANT) std::fprintf(f, "texture constant %.17g %.17g %.17g\n", t.c[0], t.c[1], t.c[2]);
        else if (t.kind == RT_TEX_CHEQUER) std::fprintf(f, "texture chequer %d %d\n", t.even, t.odd);
        else std::fprintf(f, "texture perlin\n");
    }
    for (mat const& m : materials) {
        if (m.kind == RT_MAT_LAMBERTIAN) std::fprintf(f, "material lambertian %d\n", m.tex);
        else if (m.kind == RT_MAT_METAL) std::fprintf(f, "material metal %.17g %.17g %.17g %.17g\n", m.albedo[0], m.albedo[1], m.albedo[2], m.fuzz);
        else std::fprintf(f, "material dielectric %.17g\n", m.ri);
    }
    auto put_prim = [&](prim const& p) {
        if (p.kind == RT_PRIM_SPHERE)
            std::fprintf(f, "sphere %d %.17g %.17g %.17g %.17g\n", p.mat, p.c0[0], p.c0[1], p.c0[2], p.r);
        else
            std::fprintf(f, "moving_sphere %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", p.mat, p.c0[0], p.c0[1],
                         p.c0[2], p.c1[0], p.c1[1], p.c1[2], p.t0, p.t1, p.r);
    };
    for (item const& it : items) {
        if (it.is_bvh) std::fprintf(f, "bvh_begin %u %.17g %.17g\n", it.seed, it.t0, it.t1);
        int open_list = -1;
        for (int k = 0; k < it.count; ++k) {
            prim const& p = prims[(size_t)(it.first + k)];
            if (p.list != open_list) {
                if (open_list >= 0) std::fprintf(f, "list_end\n");
                if (p.list >= 0) std::fprintf(f, "list_begin\n");
                open_list = p.list;
            }
            put_prim(p);
        }
        if (open_list >= 0) std::fprintf(f, "list_end\n");
        if (it.is_bvh) std::fprintf(f, "bvh_end\n");
    }
    std::fclose(f);
}

namespace {
// the numbers after a line's keyword: an integer, then `n` doubles (strtod, a fraction of the
// cost of a scanf format over the whole line, which matters for scenes of 10^6 spheres)
char const* skip_word(char const* s) {
    while (*s == ' ' || *s == '\t') ++s;
    while (*s && *s != ' ' && *s != '\t' && *s != '\n' && *s != '\r') ++s;
    return s;
}

// `n` doubles after the first `words` words of the line
bool doubles_after(char const* s, int words, double* out, int n) {
    for (int k = 0; k < words; ++k) s = skip_word(s);
    for (int k = 0; k < n; ++k) {
        char* end;
        out[k] = std::strtod(s, &end);
        if (end == s) return false;
        s = end;
    }
    return true;
}

bool second_word(char const* s, char* out, size_t cap) {
    s = skip_word(s);
    while (*s == ' ' || *s == '\t') ++s;
    char const* e = skip_word(s);
    size_t const n = (size_t)(e - s);
    if (n == 0 || n >= cap) return false;
    std::memcpy(out, s, n);
    out[n] = 0;
    return true;
}

bool int_and_doubles(char const* s, int* first, double* out, int n) {
    s = skip_word(s);   // the keyword
    char* end;
    long const v = std::strtol(s, &end, 10);
    if (end == s) return false;
    *first = (int)v;
    s = end;
    for (int k = 0; k < n; ++k) {
        out[k] = std::strtod(s, &end);
        if (end == s) return false;
        s = end;
    }
    return true;
}
}  // namespace

recipe recipe::load(std::string const& path) {
    std::FILE* f = std::fopen(path.c_str(), "r");
    if (!f) throw std::runtime_error("cannot open " + path);
    recipe rc;
    char line[1024], kw[64], sub[64];
    int lineno = 0;
    bool header = false, in_bvh = false, in_list = false;
    int list_members = 0;
    item group{};
    auto bad = [&](char const* why) {
        std::fclose(f);
        throw std::runtime_error(path + ":" + std::to_string(lineno) + ": " + why);
    };
    while (std::fgets(line, sizeof line, f)) {
        ++lineno;
        {   // the first word
            char const* b = line;
            while (*b == ' ' || *b == '\t') ++b;
            char const* e = skip_word(b);
            size_t const n = (size_t)(e - b);
            if (n == 0 || n >= sizeof kw || *b == '#') continue;
            std::memcpy(kw, b, n);
            kw[n] = 0;
        }
        if (!header) {
            int ver = 0;
            if (std::strcmp(kw, "rtiow-recipe") != 0 || std::sscanf(line, "%*s %d", &ver) != 1 || ver != 1) bad("not a version-1 recipe");
            header = true;
            continue;
        }
        if (!std::strcmp(kw, "texture")) {
            tex t{};
            if (!second_word(line, sub, sizeof sub)) bad("texture kind missing");
            if (!std::strcmp(sub, "constant")) {
                t.kind = RT_TEX_CONSTANT;
                if (!doubles_after(line, 2, t.c, 3)) bad("constant texture needs R G B");
            } else if (!std::strcmp(sub, "chequer")) {
                t.kind = RT_TEX_CHEQUER;
                if (std::sscanf(line, "%*s %*s %d %d", &t.even, &t.odd) != 2) bad("chequer texture needs EVEN ODD");
                if (t.even < 0 || t.odd < 0 || t.even >= (int)rc.textures.size() || t.odd >= (int)rc.textures.size()) bad("chequer child out of range");
            } else if (!std::strcmp(sub, "perlin")) {
                t.kind = RT_TEX_PERLIN;
            } else bad("unknown texture kind");
            rc.textures.push_back(t);
        } else if (!std::strcmp(kw, "material")) {
            mat m{};
            if (!second_word(line, sub, sizeof sub)) bad("material kind missing");
            if (!std::strcmp(sub, "lambertian")) {
                m.kind = RT_MAT_LAMBERTIAN;
                double v;
                if (!doubles_after(line, 2, &v, 1) || v < 0 || v >= (double)rc.textures.size() || v != (double)(int)v) bad("lambertian needs a texture");
                m.tex = (int)v;
            } else if (!std::strcmp(sub, "metal")) {
                m.kind = RT_MAT_METAL;
                double v[4];
                if (!doubles_after(line, 2, v, 4)) bad("metal needs R G B FUZZ");
                m.albedo[0] = v[0]; m.albedo[1] = v[1]; m.albedo[2] = v[2]; m.fuzz = v[3];
            } else if (!std::strcmp(sub, "dielectric")) {
                m.kind = RT_MAT_DIELECTRIC;
                if (std::sscanf(line, "%*s %*s %lf", &m.ri) != 1) bad("dielectric needs RI");
            } else bad("unknown material kind");
            rc.materials.push_back(m);
        } else if (!std::strcmp(kw, "sphere") || !std::strcmp(kw, "moving_sphere")) {
            prim p{};
            if (kw[0] == 's') {
                p.kind = RT_PRIM_SPHERE;
                double v[4];
                if (!int_and_doubles(line, &p.mat, v, 4)) bad("sphere needs MAT CX CY CZ R");
                p.c0[0] = v[0]; p.c0[1] = v[1]; p.c0[2] = v[2]; p.r = v[3];
                std::memcpy(p.c1, p.c0, sizeof p.c0);
                p.t0 = 0; p.t1 = 1;
            } else {
                p.kind = RT_PRIM_MOVING_SPHERE;
                double v[9];
                if (!int_and_doubles(line, &p.mat, v, 9)) bad("moving_sphere needs MAT C0 C1 T0 T1 R");
                std::memcpy(p.c0, v, sizeof p.c0); std::memcpy(p.c1, v + 3, sizeof p.c1);
                p.t0 = v[6]; p.t1 = v[7]; p.r = v[8];
            }
            if (p.mat < 0 || p.mat >= (int)rc.materials.size()) bad("material out of range");
            p.list = in_list ? rc.n_lists - 1 : -1;
            if (in_list) ++list_members;
            rc.prims.push_back(p);
            if (in_bvh) group.count++;
            else rc.items.push_back({false, 0, 0, 0, (int)rc.prims.size() - 1, 1});
        } else if (!std::strcmp(kw, "bvh_begin")) {
            if (in_bvh) bad("nested bvh_begin");
            unsigned long seed;
            group = item{};
            if (std::sscanf(line, "%*s %lu %lf %lf", &seed, &group.t0, &group.t1) != 3) bad("bvh_begin needs SEED T0 T1");
            group.is_bvh = true; group.seed = (uint32_t)seed; group.first = (int)rc.prims.size();
            in_bvh = true;
        } else if (!std::strcmp(kw, "list_begin")) {
            if (!in_bvh || in_list) bad("list_begin outside a bvh group, or nested");
            in_list = true;
            list_members = 0;
            ++rc.n_lists;
        } else if (!std::strcmp(kw, "list_end")) {
            if (!in_list || list_members == 0) bad("list_end without primitives");
            in_list = false;
        } else if (!std::strcmp(kw, "bvh_end")) {
            if (in_list) bad("bvh_end inside a list group");
            if (!in_bvh || group.count == 0) bad("bvh_end without primitives");
            in_bvh = false;
            rc.items.push_back(group);
        } else bad("unknown keyword");
    }
    if (!header || in_bvh) { lineno = 0; bad("truncated recipe"); }
    std::fclose(f);
    return rc;
}

namespace {

std::unique_ptr<texture const> make_texture(recipe const& rc, int id) {
    recipe::tex const& t = rc.textures[(size_t)id];
    if (t.kind == RT_TEX_CONSTANT) return std::make_unique<constant_texture const>(colour{t.c[0], t.c[1], t.c[2]});
    if (t.kind == RT_TEX_CHEQUER) return std::make_unique<chequer_texture const>(make_texture(rc, t.even), make_texture(rc, t.odd));
    return std::make_unique<perlin_texture const>();
}

std::unique_ptr<material const> make_material(recipe const& rc, int id, std::mt19937& mt) {
    recipe::mat const& m = rc.materials[(size_t)id];
    if (m.kind == RT_MAT_LAMBERTIAN) return std::make_unique<lambertian const>(mt, make_texture(rc, m.tex));
    if (m.kind == RT_MAT_METAL) return std::make_unique<metal const>(mt, vec3{m.albedo[0], m.albedo[1], m.albedo[2]}, m.fuzz);
    return std::make_unique<dielectric const>(mt, m.ri);
}

std::unique_ptr<object const> make_prim(recipe const& rc, int id, std::mt19937& mt, std::vector<material const*>* by_prim) {
    recipe::prim const& p = rc.prims[(size_t)id];
    std::unique_ptr<material const> m = make_material(rc, p.mat, mt);
    if (by_prim) (*by_prim)[(size_t)id] = m.get();
    if (p.kind == RT_PRIM_SPHERE) return std::make_unique<sphere const>(vec3{p.c0[0], p.c0[1], p.c0[2]}, p.r, std::move(m));
    return std::make_unique<moving_sphere const>(vec3{p.c0[0], p.c0[1], p.c0[2]}, vec3{p.c1[0], p.c1[1], p.c1[2]}, p.t0, p.t1, p.r, std::move(m));
}

}  // namespace

std::unique_ptr<object const> recipe::instantiate(std::mt19937& mt, std::vector<material const*>* by_prim) const {
    if (by_prim) by_prim->assign(prims.size(), nullptr);
    std::vector<std::unique_ptr<object const>> world;
    for (item const& it : items) {
        if (!it.is_bvh) {
            world.push_back(make_prim(*this, it.first, mt, by_prim));
            continue;
        }
        // the group's objects: its primitives, the members of a list group as one object_list
        auto const t_group = std::chrono::steady_clock::now();
        std::vector<std::unique_ptr<object const>> objs;
        objs.reserve((size_t)it.count);
        // A large group of plain primitives is created by several threads, each a slice of the group (every
        // primitive is three allocations -- sphere, material, texture -- and nothing is drawn from `mt`: the
        // materials only keep a reference to it).
        bool plain = it.count >= (1 << 16);
        for (int k = 0; plain && k < it.count; ++k) plain = prims[(size_t)(it.first + k)].list < 0;
        if (plain) {
            objs.resize((size_t)it.count);
            int const slices = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
            std::vector<std::exception_ptr> failed((size_t)slices);
            auto slice = [&](int t) {
                try {
                    for (int k = (int)((long long)it.count * t / slices); k < (int)((long long)it.count * (t + 1) / slices); ++k)
                        objs[(size_t)k] = make_prim(*this, it.first + k, mt, by_prim);
                } catch (...) { failed[(size_t)t] = std::current_exception(); }
            };
            std::vector<std::thread> threads;
            try {
                for (int t = 1; t < slices; ++t) threads.emplace_back(slice, t);
            } catch (std::system_error const&) {}     // fewer threads than hoped for: the rest is done here
            int const started = (int)threads.size() + 1;
            slice(0);
            for (int t = started; t < slices; ++t) slice(t);
            for (std::thread& t : threads) t.join();
            for (auto const& f : failed) if (f) std::rethrow_exception(f);
        }
        for (int k = 0; !plain && k < it.count;) {
            int const list = prims[(size_t)(it.first + k)].list;
            if (list < 0) { objs.push_back(make_prim(*this, it.first + k, mt, by_prim)); ++k; continue; }
            std::vector<std::unique_ptr<object const>> members;
            for (; k < it.count && prims[(size_t)(it.first + k)].list == list; ++k) members.push_back(make_prim(*this, it.first + k, mt, by_prim));
            objs.push_back(std::make_unique<object_list const>(std::move(members)));
        }
        std::mt19937 bm(it.seed);
        auto const t_objects = std::chrono::steady_clock::now();
        world.push_back(std::make_unique<bvh_node const>(bm, objs, it.t0, it.t1));
        if (std::getenv("RTIOW_VERBOSE") && it.count >= (1 << 16))
            std::fprintf(stderr, "rtiow host:   %d objects created in %.3f s, bvh_node built in %.3f s\n", it.count,
                         std::chrono::duration<double>(t_objects - t_group).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_objects).count());
    }
    return std::make_unique<object_list const>(std::move(world));
}

// ------------------------------------------------------------------ generators ----
recipe recipe_two_perlin_spheres() {
    recipe rc;
    rc.textures.push_back({RT_TEX_PERLIN, {0, 0, 0}, 0, 0});
    rc.textures.push_back({RT_TEX_PERLIN, {0, 0, 0}, 0, 0});
    int const m0 = rc.add_lambertian(0), m1 = rc.add_lambertian(1);
    rc.add_sphere(m0, 0, -1000, 0, 1000);
    rc.items.push_back({false, 0, 0, 0, 0, 1});
    rc.add_sphere(m1, 0, 2, 0, 2);
    rc.items.push_back({false, 0, 0, 0, 1, 1});
    return rc;
}

recipe recipe_random_spheres(uint32_t seed, bool moving, bool chequer) {
    recipe rc;
    std::mt19937 mt(seed);
    std::uniform_real_distribution<double> dist{0, 1};
    int ground_tex;
    if (chequer) {
        int const even = rc.add_constant(0.2, 0.3, 0.1), odd = rc.add_constant(0.9, 0.9, 0.9);
        rc.textures.push_back({RT_TEX_CHEQUER, {0, 0, 0}, even, odd});
        ground_tex = (int)rc.textures.size() - 1;
    } else {
        ground_tex = rc.add_constant(0.5, 0.5, 0.5);
    }
    rc.add_sphere(rc.add_lambertian(ground_tex), 0, -1000, -1, 1000);
    rc.add_sphere(rc.add_dielectric(1.5), 0, 1, 0, 1);
    rc.add_sphere(rc.add_lambertian(rc.add_constant(0.4, 0.2, 0.1)), -4, 1, 0, 1);
    rc.add_sphere(rc.add_metal(0.7, 0.6, 0.5, 0), 4, 1, 0, 1);
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            double const cx = a + 0.9 * dist(mt);
            double const cz = b + 0.9 * dist(mt);
            double const dx = cx - 4, dy = 0.2 - 0.2, dz = cz - 0;
            if (std::sqrt(dx * dx + dy * dy + dz * dz) <= 0.9) continue;
            double const choose_mat = dist(mt);
            if (choose_mat < 0.8) {
                double const lift = 0.5 * dist(mt);
                double const r1 = dist(mt), r2 = dist(mt), g1 = dist(mt), g2 = dist(mt), b1 = dist(mt), b2 = dist(mt);
                int const m = rc.add_lambertian(rc.add_constant(r1 * r2, g1 * g2, b1 * b2));
                double const c0[3] = {cx, 0.2, cz}, c1[3] = {cx + 0, 0.2 + lift, cz + 0};
                if (moving) rc.add_moving_sphere(m, c0, c1, 0, 1, 0.2);
                else rc.add_sphere(m, cx, 0.2, cz, 0.2);
            } else if (choose_mat < 0.95) {
                double const ar = 0.5 * (1 + dist(mt)), ag = 0.5 * (1 + dist(mt)), ab = 0.5 * (1 + dist(mt));
                double const fuzz = 0.5 * dist(mt);
                rc.add_sphere(rc.add_metal(ar, ag, ab, fuzz), cx, 0.2, cz, 0.2);
            } else {
                rc.add_sphere(rc.add_dielectric(1.5), cx, 0.2, cz, 0.2);
            }
        }
    }
    rc.items.push_back({true, (uint32_t)mt(), 0, 1, 0, (int)rc.prims.size()});
    return rc;
}

recipe recipe_stress(uint32_t seed, int n) {
    recipe rc;
    std::mt19937 mt(seed);
    std::uniform_real_distribution<double> dist{0, 1};
    double const R = 100000.0;
    double const side = std::sqrt((double)(n > 0 ? n : 1));  // one sphere per unit area
    rc.prims.reserve((size_t)(n > 0 ? n : 0) + 1);
    rc.materials.reserve((size_t)(n > 0 ? n : 0) + 1);
    rc.textures.reserve((size_t)(n > 0 ? n : 0) + 1);
    rc.add_sphere(rc.add_lambertian(rc.add_constant(0.5, 0.5, 0.5)), 0, -R, 0, R);
    for (int i = 0; i < n; ++i) {
        double const x = (dist(mt) - 0.5) * side, z = (dist(mt) - 0.5) * side;
        double const r = 0.2 + 0.2 * dist(mt);
        double const y = std::sqrt(R * R - x * x - z * z) - R + r;
        double const choose_mat = dist(mt);
        int m;
        if (choose_mat < 0.8) {
            double const cr = dist(mt) * dist(mt), cg = dist(mt) * dist(mt), cb = dist(mt) * dist(mt);
            m = rc.add_lambertian(rc.add_constant(cr, cg, cb));
        } else if (choose_mat < 0.95) {
            double const ar = 0.5 * (1 + dist(mt)), ag = 0.5 * (1 + dist(mt)), ab = 0.5 * (1 + dist(mt));
            double const fuzz = 0.5 * dist(mt);
            m = rc.add_metal(ar, ag, ab, fuzz);
        } else {
            m = rc.add_dielectric(1.5);
        }
        rc.add_sphere(m, x, y, z, r);
    }
    rc.items.push_back({true, (uint32_t)mt(), 0, 1, 0, (int)rc.prims.size()});
    return rc;
}

}  // namespace rtiow
