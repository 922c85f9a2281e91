// raytracer.cpp -- the frame driver: builds a world with the scene API, renders it on
// one MI355X (or, with --devices, on several: the samples are split between them) and writes the image as the reference's main() does (main.cpp:107-151):
// "Render duration" on stderr, P3 text on stdout.  The reference fixes scene, size and
// sample count at compile time (main.cpp:108,113); here they are flags whose defaults
// are those constants.  --camera takes the view the reference hard-codes at main.cpp:114-120
// (look-from, look-at, vertical field of view); --format p6 and png hold the same bytes as
// the reference's P3 text.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "recipe.hpp"
#include "rtiow.hpp"

using namespace rtiow;

static int usage() {
    std::fprintf(stderr,
                 "usage: raytracer [--scene two_perlin|book1|book2|stress|FILE.recipe] [--nx N] [--ny N] [--ns N]\n"
                 "                 [--seed S] [--scene-seed S] [--spheres N] [--device D | --devices D0,D1,...] [--aperture A] [--focus F]\n"
                 "                 [--camera FX,FY,FZ,AX,AY,AZ,VFOV] [--out FILE] [--format p3|p6|png] [--binary]\n"
                 "  without --format the suffix of FILE decides (.png: png, else p3); --binary is --format p6\n");
    return 2;
}

int main(int argc, char** argv) {
    std::string scene = "two_perlin", out_path;
    render_options opt;
    uint32_t scene_seed = 1;
    int spheres = 1000000;
    double aperture = 0;
    std::string format;    // p3 (the reference's text), p6 or png
    double view[7] = {13, 2, 3, 0, 0, 0, 20}, focus = 10;   // main.cpp:114-120
    for (int i = 1; i < argc; ++i) {
        std::string const a = argv[i];
        auto need = [&](char const* name) -> char const* {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(2); }
            return argv[++i];
        };
        if (a == "--scene") scene = need("--scene");
        else if (a == "--nx") opt.nx = std::atoi(need("--nx"));
        else if (a == "--ny") opt.ny = std::atoi(need("--ny"));
        else if (a == "--ns") opt.ns = std::atoi(need("--ns"));
        else if (a == "--seed") opt.seed = std::strtoull(need("--seed"), nullptr, 10);
        else if (a == "--scene-seed") scene_seed = (uint32_t)std::strtoul(need("--scene-seed"), nullptr, 10);
        else if (a == "--spheres") spheres = std::atoi(need("--spheres"));
        else if (a == "--device") opt.device = std::atoi(need("--device"));
        else if (a == "--devices") {
            for (char const* p = need("--devices"); *p;) {
                char* end = nullptr;
                opt.devices.push_back((int)std::strtol(p, &end, 10));
                if (end == p) return usage();
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (a == "--aperture") aperture = std::atof(need("--aperture"));
        else if (a == "--out") out_path = need("--out");
        else if (a == "--binary") format = "p6";
        else if (a == "--format") format = need("--format");
        else if (a == "--focus") focus = std::atof(need("--focus"));
        else if (a == "--camera") {
            if (std::sscanf(need("--camera"), "%lf,%lf,%lf,%lf,%lf,%lf,%lf", view, view + 1, view + 2, view + 3, view + 4, view + 5, view + 6) != 7) return usage();
        }
        else return usage();
    }
    if (format.empty()) format = out_path.size() > 4 && out_path.compare(out_path.size() - 4, 4, ".png") == 0 ? "png" : "p3";
    if (format != "p3" && format != "p6" && format != "png") return usage();
    try {
        std::mt19937 mt(scene_seed);
        recipe rc;
        if (scene == "two_perlin") rc = recipe_two_perlin_spheres();
        else if (scene == "book1") rc = recipe_random_spheres(scene_seed, false, false);
        else if (scene == "book2") rc = recipe_random_spheres(scene_seed, true, true);
        else if (scene == "stress") rc = recipe_stress(scene_seed, spheres);
        else rc = recipe::load(scene);
        std::unique_ptr<object const> world = rc.instantiate(mt);
        // the camera of main.cpp:114-120
        camera const cam{mt, vec3{view[0], view[1], view[2]}, vec3{view[3], view[4], view[5]}, vec3{0, 1, 0}, view[6], double(opt.nx) / double(opt.ny), aperture, focus, 0, 1};
        std::unique_ptr<device_scene> dev;
        std::vector<std::unique_ptr<device_scene>> devs;
        if (opt.devices.empty()) dev = std::make_unique<device_scene>(*world, opt.device);
        else devs = upload(*world, opt.devices);
        auto const start = std::chrono::steady_clock::now();
        frame const f = dev ? render(*dev, cam, opt) : render(devs, cam, opt);
        auto const end = std::chrono::steady_clock::now();
        double const secs = std::chrono::duration<double>(end - start).count();
        std::fprintf(stderr, "Render duration: %gs\n", secs);
        std::fprintf(stderr, "kernel %.3f ms, %.3f Mpaths/s, %.3f rays/path\n", f.stats.kernel_ms,
                     (double)f.stats.paths / (f.stats.kernel_ms * 1e3), (double)f.stats.rays / (double)f.stats.paths);
        std::FILE* out = out_path.empty() ? stdout : std::fopen(out_path.c_str(), "wb");
        if (!out) { std::fprintf(stderr, "cannot write %s\n", out_path.c_str()); return 1; }
        if (format == "png") f.write_png(out);
        else if (format == "p6") f.write_ppm_binary(out);
        else f.write_ppm(out);
        if (out != stdout) std::fclose(out);
    } catch (std::exception const& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    } catch (char const* e) {
        std::fprintf(stderr, "error: %s\n", e);
        return 1;
    }
    return 0;
}
