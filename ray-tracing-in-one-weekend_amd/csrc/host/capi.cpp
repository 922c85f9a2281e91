// capi.cpp -- C entry points over the host scene API, for callers that cannot use the
// C++ classes directly (the Python harness in tests/ and bench.py).  They only build
// worlds, cameras and recipes; every per-ray call goes to rtiow_hip.h with the
// rt_scene handle obtained here.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <vector>
#include <memory>
#include <stdexcept>
#include <string>

#include "recipe.hpp"
#include "rtiow.hpp"

using namespace rtiow;

namespace {
thread_local std::string g_error;

template <class F> int guarded(F&& f) {
    try {
        f();
        return 0;
    } catch (std::exception const& e) {
        g_error = e.what();
    } catch (char const* e) {
        g_error = e;
    } catch (...) {
        g_error = "unknown error";
    }
    return 1;
}
}  // namespace

struct rth_world {
    std::mt19937 mt{0};                       // the generator the materials are bound to
    std::unique_ptr<object const> world;
    std::unique_ptr<device_scene> scene;
    std::unique_ptr<scene_flattener> flat;   // tables of a world that was not uploaded
    rt_scene_desc desc{};
    std::vector<int32_t> recipe_id;           // recipe primitive number of each table primitive (filled on demand)
    std::vector<material const*> by_prim;     // the recipe's primitives' materials, in recipe order
    double seconds[4] = {0, 0, 0, 0};         // generate or load | objects + bvh_node build | flatten | rt_scene_create

    scene_flattener const& tables() const { return scene ? scene->tables() : *flat; }

    // recipe -> objects through the scene API -> tables (-> GPU when `upload`)
    template <class F> void build(F&& make_recipe, int device, bool upload) {
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](int k) {
            auto const t1 = std::chrono::steady_clock::now();
            seconds[k] = std::chrono::duration<double>(t1 - t0).count();
            t0 = t1;
        };
        recipe const rc = make_recipe();
        lap(0);
        world = rc.instantiate(mt, &by_prim);
        lap(1);
        if (upload) {
            scene = std::make_unique<device_scene>(*world, device);
            seconds[2] = scene->flatten_seconds; seconds[3] = scene->create_seconds;
        } else {
            flat = std::make_unique<scene_flattener>();
            world->flatten_items(*flat);
            lap(2);
        }
        desc = tables().desc();
        if (std::getenv("RTIOW_VERBOSE"))
            std::fprintf(stderr, "rtiow host: recipe %.3f s, objects and bvh_node build %.3f s, flatten %.3f s, rt_scene_create %.3f s\n", seconds[0], seconds[1], seconds[2], seconds[3]);
    }

    void number_prims() {
        if (!recipe_id.empty() || tables().material_of_prim.empty()) return;
        scene_flattener const& tables = this->tables();
        // every primitive owns its material: the material's address identifies the primitive
        std::vector<std::pair<material const*, int32_t>> id(by_prim.size());
        for (size_t i = 0; i < by_prim.size(); ++i) id[i] = {by_prim[i], (int32_t)i};
        std::sort(id.begin(), id.end());
        recipe_id.resize(tables.material_of_prim.size());
        for (size_t i = 0; i < recipe_id.size(); ++i) {
            material const* const m = tables.material_of_prim[i];
            auto const it = std::lower_bound(id.begin(), id.end(), std::make_pair(m, (int32_t)0));
            if (it == id.end() || it->first != m) throw std::runtime_error("a primitive's material is not one of the recipe's");
            recipe_id[i] = it->second;
        }
    }
};

namespace {
recipe generate(const char* kind, uint32_t seed, int n) {
    std::string const k = kind ? kind : "";
    if (k == "two_perlin") return recipe_two_perlin_spheres();
    if (k == "book1") return recipe_random_spheres(seed, false, false);
    if (k == "book2") return recipe_random_spheres(seed, true, true);
    if (k == "stress") return recipe_stress(seed, n);
    throw std::runtime_error("unknown scene kind '" + k + "'");
}
}  // namespace
extern "C" {

const char* rth_last_error(void) { return g_error.c_str(); }

// kind: "two_perlin", "book1" (static spheres, grey ground), "book2" (moving spheres,
// chequer ground: main.cpp:61-99 as written), "stress" (n spheres)
int rth_generate(const char* kind, uint32_t seed, int n, const char* path) {
    return guarded([&] { generate(kind, seed, n).save(path); });
}

// Reads a recipe and writes it out again (the writer's own formatting): what a saved scene must survive.
int rth_recipe_copy(const char* from, const char* to) {
    return guarded([&] { recipe::load(from).save(to); });
}

// The same as rth_generate + rth_world_load without the detour through a recipe file: the
// generator's world is built through the scene API, flattened and uploaded to `device`.
int rth_world_generate(const char* kind, uint32_t seed, int n, int device, rth_world** out) {
    *out = nullptr;
    return guarded([&] {
        auto w = std::make_unique<rth_world>();
        w->build([&] { return generate(kind, seed, n); }, device, true);
        *out = w.release();
    });
}

// Builds the recipe's world through the scene API, flattens it and uploads it to `device`.
int rth_world_load(const char* recipe_path, int device, rth_world** out) {
    *out = nullptr;
    return guarded([&] {
        auto w = std::make_unique<rth_world>();
        w->build([&] { return recipe::load(recipe_path); }, device, true);
        *out = w.release();
    });
}

void rth_world_free(rth_world* w) { delete w; }
rt_scene* rth_world_scene(rth_world* w) { return w->scene->get(); }
const rt_scene_desc* rth_world_desc(rth_world* w) { return &w->desc; }

// The tables number primitives in the order the world was flattened (left-to-right leaf
// order); this gives, for each of them, its number in the recipe (construction order).
const int32_t* rth_world_recipe_ids(rth_world* w) {
    if (guarded([&] { w->number_prims(); })) return nullptr;
    return w->recipe_id.data();
}

// seconds spent in: making the recipe (generator or file), objects + bvh_node build, flatten, rt_scene_create
void rth_world_seconds(rth_world* w, double* out4) { std::memcpy(out4, w->seconds, sizeof w->seconds); }

// Flattens a recipe's world without touching a GPU: tables only (for host-side checks).
int rth_world_tables(const char* recipe_path, rth_world** out) {
    *out = nullptr;
    return guarded([&] {
        auto w = std::make_unique<rth_world>();
        w->build([&] { return recipe::load(recipe_path); }, -1, false);
        *out = w.release();
    });
}

// cam: from(3) at(3) up(3) vfov aspect aperture focus_dist t0 t1  (camera.cpp:25-36)
int rth_camera(const double* cam, rt_camera* out) {
    return guarded([&] {
        std::mt19937 mt;
        camera const c{mt, vec3{cam[0], cam[1], cam[2]}, vec3{cam[3], cam[4], cam[5]}, vec3{cam[6], cam[7], cam[8]},
                       cam[9], cam[10], cam[11], cam[12], cam[13], cam[14]};
        *out = c.describe();
    });
}

// Writes per-pixel sums (nx*ny*3 floats, top row first, ns samples each) the way the frame
// class does: format "p3" (main.cpp:145-150), "p6" or "png".
int rth_write_image(const char* path, const char* format, int nx, int ny, int ns, const float* sums) {
    return guarded([&] {
        std::string const fmt = format ? format : "";
        if (nx <= 0 || ny <= 0 || !sums || !path) throw std::runtime_error("rth_write_image: bad argument");
        if (fmt != "p3" && fmt != "p6" && fmt != "png") throw std::runtime_error("unknown image format '" + fmt + "'");
        frame f;
        f.nx = nx; f.ny = ny; f.ns = ns;
        f.sums.assign(sums, sums + (size_t)nx * ny * 3);
        std::FILE* out = std::fopen(path, "wb");
        if (!out) throw std::runtime_error(std::string("cannot write ") + path);
        if (fmt == "png") f.write_png(out);
        else if (fmt == "p6") f.write_ppm_binary(out);
        else f.write_ppm(out);
        if (std::fclose(out) != 0) throw std::runtime_error(std::string("error writing ") + path);
    });
}

void rth_perlin_tables(double* ranfloat, uint8_t* px, uint8_t* py, uint8_t* pz) {
    std::memcpy(ranfloat, perlin::ranfloat.data(), 256 * sizeof(double));
    std::memcpy(px, perlin::perm_x.data(), 256);
    std::memcpy(py, perlin::perm_y.data(), 256);
    std::memcpy(pz, perlin::perm_z.data(), 256);
}

}  // extern "C"
