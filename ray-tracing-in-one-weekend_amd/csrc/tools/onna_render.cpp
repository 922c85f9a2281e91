// onna_render -- the stand-in for the Lisp tracer's `example` (onna/onna.lisp:5-23): builds one of the
// scenes of onna/scenes.lisp through the host API, renders it on the GPU and writes the PNG that
// write-png (onna/image.lisp:30-40) would.  The scene, image width, samples, depth limit, seed, GPU
// and output file are flags instead of edits to the source.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "onna.hpp"

int main(int argc, char** argv) {
    std::string name = "cornell-box", out = "onna.png", image_path;
    uint32_t width = 0, samples = 0, depth = 0;
    uint64_t seed = 1;
    int device = -1;
    std::vector<int> devices;
    for (int i = 1; i < argc; ++i) {
        auto value = [&](char const* flag) -> char const* {
            if (std::strcmp(argv[i], flag) != 0) return nullptr;
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", flag); std::exit(2); }
            return argv[++i];
        };
        if (char const* v = value("--scene")) name = v;
        else if (char const* v = value("--out")) out = v;
        else if (char const* v = value("--image")) image_path = v;
        else if (char const* v = value("--width")) width = (uint32_t)std::atoi(v);
        else if (char const* v = value("--samples")) samples = (uint32_t)std::atoi(v);
        else if (char const* v = value("--depth")) depth = (uint32_t)std::atoi(v);
        else if (char const* v = value("--seed")) seed = std::strtoull(v, nullptr, 10);
        else if (char const* v = value("--device")) device = std::atoi(v);
        else if (char const* v = value("--devices")) {   // 0,1,2,...: the frame's samples split over these GPUs
            for (char const* q = v; *q;) { devices.push_back((int)std::strtol(q, const_cast<char**>(&q), 10)); if (*q == ',') ++q; }
        }
        else {
            std::fprintf(stderr, "usage: onna_render [--scene NAME] [--width W] [--samples N] [--depth D] [--seed S] [--device G | --devices 0,1,...] [--image earthmap.png] [--out FILE.png]\nscenes:");
            for (auto const& s : onna::scene_names()) std::fprintf(stderr, " %s", s.c_str());
            std::fprintf(stderr, "\n");
            return 2;
        }
    }
    try {
        onna::random_state rs(seed);
        onna::image8 image;
        if (!image_path.empty()) image = onna::read_png(image_path);
        onna::scene s = onna::scene_by_name(name, rs, image);
        if (width) s.camera.image_width = width;
        if (samples) s.camera.samples_per_pixel = samples;
        if (depth) s.camera.max_depth = depth;
        on_camera const cam = onna::make_camera(s.camera);
        onna::frame const f = devices.empty() ? onna::render(cam, s.world, s.lights, seed, device) : onna::render(cam, s.world, s.lights, seed, devices);
        onna::write_png(out, f.resolve());
        std::fprintf(stderr, "Render duration: %.3f s (%ux%u, %u samples per pixel, %.1f Mpaths/s, %.2f rays per path)\n", f.stats.kernel_ms * 1e-3, f.width,
                     f.height, f.samples, (double)f.stats.paths / (f.stats.kernel_ms * 1e3), (double)f.stats.rays / (double)f.stats.paths);
    } catch (std::exception const& e) {
        std::fprintf(stderr, "onna_render: %s\n", e.what());
        return 1;
    }
    return 0;
}
