// onna_api_check -- exercises the host scene API of onna.hpp the way a program written against the
// reference's Lisp constructors would: builds objects, asks for hits, renders, writes the image.
// Every ray is traced on the GPU; run by the GPU test suite.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include <vector>

#include "onna.hpp"

#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) { std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main() {
    using namespace onna;
    try {
        // hit-test of single objects (scene.lisp:22)
        object_ptr const q = make_quad({-1, -1, -2}, {2, 0, 0}, {0, 2, 0});
        rt_ray const ray{{0.5, -0.25, 0}, {0, 0, -1}, 0};
        on_hit h = hit_test(ray, q, {0.001, HUGE_VAL});
        CHECK(h.hit == 1 && h.t == 2.0 && h.u == 0.75 && h.v == 0.375 && h.normal[2] == 1.0);
        h = hit_test(ray, make_translate(q, {5, 0, 0}), {0.001, HUGE_VAL});
        CHECK(h.hit == 0);
        object_ptr const box = make_translate(make_rotate_y(make_box({-1, -1, -1}, {1, 1, 1}, nullptr), 45), {0, 0, -5});
        h = hit_test(rt_ray{{0, 0, 0}, {0, 0, -1}, 0}, box, {0.001, HUGE_VAL});
        CHECK(h.hit == 1 && std::fabs(h.t - (5 - std::sqrt(2.0))) < 1e-12);   // the edge of the turned box faces the ray
        aabb const bb = box->bounding_box();
        CHECK(std::fabs(bb.x.min + std::sqrt(2.0)) < 1e-3 && std::fabs(bb.z.max + 5 - std::sqrt(2.0)) < 1e-3);

        // a world and its lights, rendered twice: the frame repeats
        scene const s = cornell_box();
        camera_options opt = s.camera;
        opt.image_width = 32; opt.samples_per_pixel = 4; opt.max_depth = 8;
        on_camera const cam = make_camera(opt);
        device_scene const on_gpu(s.world, s.lights);
        frame const a = render(cam, on_gpu, 7), b = render(cam, on_gpu, 7);
        CHECK(a.width == 32 && a.height == 32 && a.samples == 4 && a.sums == b.sums && a.stats.paths == 32u * 32u * 4u && a.stats.rays >= a.stats.paths);
        double total = 0;
        for (double v : a.sums) { CHECK(std::isfinite(v) && v >= 0); total += v; }
        CHECK(total > 0);
        image8 const img = a.resolve();
        std::string const path = "/tmp/onna_api_check.png";
        write_png(path, img);
        image8 const back = read_png(path);
        CHECK(back.width == img.width && back.height == img.height && back.rgb == img.rgb);

        // the frame's samples split over a list of GPUs (here the same one twice): the sums of the two ranges
        frame const split = render(cam, s.world, s.lights, 7, std::vector<int>{0, 0});
        CHECK(split.sums.size() == a.sums.size() && split.stats.paths == a.stats.paths && split.stats.rays == a.stats.rays);
        for (size_t i = 0; i < a.sums.size(); ++i) CHECK(std::fabs(split.sums[i] - a.sums[i]) <= 1e-13 * std::fabs(a.sums[i]) + 1e-15);

        // errors are exceptions
        bool threw = false;
        try { make_bvh_node({}); } catch (std::invalid_argument const&) { threw = true; }
        CHECK(threw);
        threw = false;
        try { render(cam, make_sphere({0, 0, 0}, 1.0), nullptr); } catch (std::runtime_error const&) { threw = true; }   // no material
        CHECK(threw);
        std::printf("onna api check ok: %llu paths, %llu rays, %.3f ms\n", (unsigned long long)a.stats.paths, (unsigned long long)a.stats.rays, a.stats.kernel_ms);
    } catch (std::exception const& e) {
        std::fprintf(stderr, "onna_api_check: %s\n", e.what());
        return 1;
    }
    return 0;
}
