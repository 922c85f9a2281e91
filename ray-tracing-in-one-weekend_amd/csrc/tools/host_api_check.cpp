// host_api_check.cpp -- exercises the host scene API (rtiow.hpp) the way a program written
// against the reference's classes would: object::hit, material::scatter, texture::value,
// camera::get_ray, aabb::hit, one call at a time, each of which runs on the GPU.  The record
// layouts are those of the golden vectors under tests/golden/ (raw little-endian doubles),
// so tests/test_host_api_gpu.py can compare the output files with them directly.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <limits>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "recipe.hpp"
#include "rtiow.hpp"

using namespace rtiow;

namespace {

std::vector<double> read_doubles(char const* path) {
    std::FILE* f = std::fopen(path, "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    long const bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<double> v((size_t)bytes / sizeof(double));
    if (!v.empty() && std::fread(v.data(), sizeof(double), v.size(), f) != v.size()) std::exit(2);
    std::fclose(f);
    return v;
}

void write_doubles(char const* path, std::vector<double> const& v) {
    std::FILE* f = std::fopen(path, "wb");
    if (!f || (!v.empty() && std::fwrite(v.data(), sizeof(double), v.size(), f) != v.size())) {
        std::fprintf(stderr, "cannot write %s\n", path);
        std::exit(2);
    }
    std::fclose(f);
}

// uniforms the generator produced between two of its states
int replay(std::mt19937 before, std::mt19937 const& after, double* out, int kmax) {
    int n = 0;
    while (!(before == after)) {
        double const u = std::generate_canonical<double, 53>(before);
        if (n < kmax) out[n] = u;
        if (++n > 100000) { std::fprintf(stderr, "generator states never meet\n"); std::exit(2); }
    }
    return n;
}

class boxless final : public object {
public:
    std::optional<aabb> bounding_box(double, double) const override { return std::nullopt; }
    int32_t flatten_child(scene_flattener&, aabb const&) const override { return 0; }
};

}  // namespace

int main(int argc, char** argv) try {
    std::string const cmd = argc > 1 ? argv[1] : "";
    if (cmd == "hit" && argc == 7) {   // RECIPE RAYS TMIN TMAX OUT -> [n][9] hit prim t p n
        std::mt19937 mt(1);
        std::vector<material const*> by_prim;
        auto world = recipe::load(argv[2]).instantiate(mt, &by_prim);
        std::map<material const*, int> prim_of;
        for (size_t i = 0; i < by_prim.size(); ++i) prim_of[by_prim[i]] = (int)i;
        auto const in = read_doubles(argv[3]);
        double const tmin = std::atof(argv[4]);
        double const tmax = std::string(argv[5]) == "max" ? std::numeric_limits<double>::max() : std::atof(argv[5]);
        std::vector<double> out(in.size() / 7 * 9, 0.0);
        for (size_t i = 0; i < in.size() / 7; ++i) {
            double const* a = &in[i * 7];
            auto const h = world->hit(ray{vec3{a[0], a[1], a[2]}, vec3{a[3], a[4], a[5]}, a[6]}, tmin, tmax);
            double* o = &out[i * 9];
            o[1] = -1;
            if (h) {
                o[0] = 1; o[1] = prim_of.at(h->material); o[2] = h->t;
                o[3] = h->p.x(); o[4] = h->p.y(); o[5] = h->p.z();
                o[6] = h->normal.x(); o[7] = h->normal.y(); o[8] = h->normal.z();
            }
        }
        write_doubles(argv[6], out);
        return 0;
    }
    if (cmd == "scatter" && argc == 7) {   // RECIPE SEED K IN OUT: in[n][15] prim ray t p n -> [n][12+K]
        std::mt19937 mt((uint32_t)std::strtoul(argv[3], nullptr, 10));
        std::vector<material const*> by_prim;
        auto world = recipe::load(argv[2]).instantiate(mt, &by_prim);
        int const K = std::atoi(argv[4]);
        auto const in = read_doubles(argv[5]);
        size_t const n = in.size() / 15, rec = 12 + (size_t)K;
        std::vector<double> out(n * rec, 0.0);
        for (size_t i = 0; i < n; ++i) {
            double const* a = &in[i * 15];
            material const* m = by_prim.at((size_t)a[0]);
            ray const r{vec3{a[1], a[2], a[3]}, vec3{a[4], a[5], a[6]}, a[7]};
            hit_record const h{a[8], vec3{a[9], a[10], a[11]}, vec3{a[12], a[13], a[14]}, m};
            std::mt19937 const before = mt;
            auto const sc = m->scatter(r, h);
            double* o = &out[i * rec];
            o[11] = replay(before, mt, o + 12, K);
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
    if (cmd == "camera" && argc == 21) {   // SEED K CAM(15) ST OUT: st[n][2] -> [n][8+K]
        std::mt19937 mt((uint32_t)std::strtoul(argv[2], nullptr, 10));
        int const K = std::atoi(argv[3]);
        double c[15];
        for (int k = 0; k < 15; ++k) c[k] = std::atof(argv[4 + k]);
        camera const cam{mt, vec3{c[0], c[1], c[2]}, vec3{c[3], c[4], c[5]}, vec3{c[6], c[7], c[8]}, c[9], c[10], c[11], c[12], c[13], c[14]};
        auto const in = read_doubles(argv[19]);
        size_t const n = in.size() / 2, rec = 8 + (size_t)K;
        std::vector<double> out(n * rec, 0.0);
        for (size_t i = 0; i < n; ++i) {
            std::mt19937 const before = mt;
            ray const r = cam.get_ray(in[i * 2], in[i * 2 + 1]);
            double* o = &out[i * rec];
            o[0] = r.origin().x(); o[1] = r.origin().y(); o[2] = r.origin().z();
            o[3] = r.direction().x(); o[4] = r.direction().y(); o[5] = r.direction().z();
            o[6] = r.time();
            o[7] = replay(before, mt, o + 8, K);
        }
        write_doubles(argv[20], out);
        return 0;
    }
    if (cmd == "texture" && argc == 5) {   // KIND POINTS OUT (KIND: chequer | perlin | noise)
        std::unique_ptr<texture const> tex;
        std::string const kind = argv[2];
        if (kind == "chequer")
            tex = std::make_unique<chequer_texture const>(std::make_unique<constant_texture const>(colour{0.2, 0.3, 0.1}),
                                                          std::make_unique<constant_texture const>(colour{0.9, 0.9, 0.9}));
        else tex = std::make_unique<perlin_texture const>();
        auto const in = read_doubles(argv[3]);
        std::vector<double> out(in.size());
        for (size_t i = 0; i < in.size() / 3; ++i) {
            vec3 const p{in[i * 3], in[i * 3 + 1], in[i * 3 + 2]};
            if (kind == "noise") {
                double const v = perlin{}.noise(p);
                out[i * 3] = out[i * 3 + 1] = out[i * 3 + 2] = v;
            } else {
                colour const c = tex->value(0, 0, p);
                out[i * 3] = c.red(); out[i * 3 + 1] = c.green(); out[i * 3 + 2] = c.blue();
            }
        }
        write_doubles(argv[4], out);
        return 0;
    }
    if (cmd == "aabb" && argc == 4) {   // CASES OUT: cases[n][14] -> hit[n]
        auto const in = read_doubles(argv[2]);
        std::vector<double> out(in.size() / 14);
        for (size_t i = 0; i < out.size(); ++i) {
            double const* c = &in[i * 14];
            aabb const box{vec3{c[0], c[1], c[2]}, vec3{c[3], c[4], c[5]}};
            out[i] = box.hit(ray{vec3{c[6], c[7], c[8]}, vec3{c[9], c[10], c[11]}, 0.0}, c[12], c[13]) ? 1.0 : 0.0;
        }
        write_doubles(argv[3], out);
        return 0;
    }
    if (cmd == "errors") {
        // bvh_node throws a string literal when an object has no bounding box (bvh_node.cpp:16,47)
        std::mt19937 mt(1);
        std::vector<std::unique_ptr<object const>> objs;
        objs.push_back(std::make_unique<boxless const>());
        objs.push_back(std::make_unique<boxless const>());
        try {
            bvh_node const node(mt, objs, 0, 1);
            std::puts("no throw");
            return 1;
        } catch (char const* what) {
            std::printf("caught: %s\n", what);
        }
        // supersampler (supersampler.cpp:3-19) and colour::gamma2
        supersampler ss;
        std::printf("empty: %g\n", ss.value().red());
        ss.add_sample(colour{0.25, 0.5, 1.0});
        ss.add_sample(colour{0.75, 0.5, 0.0});
        colour const m = ss.value().gamma2();
        std::printf("mean gamma: %.17g %.17g %.17g\n", m.red(), m.green(), m.blue());
        // the reference's own scene builders render
        std::mt19937 mt2(7);
        object_list const world = two_perlin_spheres(mt2);
        camera const cam{mt2, vec3{13, 2, 3}, vec3{0, 0, 0}, vec3{0, 1, 0}, 20, 2.0, 0, 10, 0, 1};
        render_options opt;
        opt.nx = 40; opt.ny = 20; opt.ns = 4; opt.seed = 3;
        frame const f = render(world, cam, opt);
        std::printf("two_perlin: %llu paths %llu rays\n", (unsigned long long)f.stats.paths, (unsigned long long)f.stats.rays);
        std::mt19937 mt3(12345);
        bvh_node const book = build_world(mt3);
        frame const g = render(book, cam, opt);
        std::printf("build_world: %llu paths %llu rays\n", (unsigned long long)g.stats.paths, (unsigned long long)g.stats.rays);
        colour const c = g.pixel(20, 10);
        std::printf("pixel: %g %g %g\n", c.red(), c.green(), c.blue());
        // render_options::devices: one GPU in the list is the single-GPU frame; the same GPU twice splits the
        // samples between two uploads of the world, which render one after the other into that GPU's sums
        opt.ns = 6;
        int dev = 0;
        frame const one = render(book, cam, opt);
        opt.devices = {dev};
        frame const listed = render(book, cam, opt);
        opt.devices = {dev, dev};
        frame const twice = render(book, cam, opt);
        double worst = 0;
        for (size_t i = 0; i < one.sums.size(); ++i) worst = std::fmax(worst, std::fabs((double)twice.sums[i] - (double)one.sums[i]));
        std::printf("devices: one listed %s, split in two: %llu rays of %llu, largest difference of the sums %.3g\n",
                    listed.sums == one.sums && listed.stats.rays == one.stats.rays ? "equal" : "DIFFERENT",
                    (unsigned long long)twice.stats.rays, (unsigned long long)one.stats.rays, worst);
        return 0;
    }
    std::fprintf(stderr, "usage: host_api_check hit|scatter|camera|texture|aabb|errors ...\n");
    return 2;
} catch (std::exception const& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
} catch (char const* e) {
    std::fprintf(stderr, "error: %s\n", e);
    return 1;
}
