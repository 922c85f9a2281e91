// capi.cpp -- C entry points over the host scene API, for callers that cannot use the
// C++ classes directly (the Python harness in tests/ and bench.py).  They only build
// worlds, cameras and recipes; every per-ray call goes to rtiow_hip.h with the
// rt_scene handle obtained here.
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
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
    std::vector<int32_t> recipe_id;           // recipe primitive number of each table primitive

    void number_prims(scene_flattener const& tables, std::vector<material const*> const& by_prim) {
        std::map<material const*, int32_t> id;
        for (size_t i = 0; i < by_prim.size(); ++i) id[by_prim[i]] = (int32_t)i;
        recipe_id.clear();
        for (material const* m : tables.material_of_prim) recipe_id.push_back(id.at(m));
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

// The same as rth_generate + rth_world_load without the detour through a recipe file: the
// generator's world is built through the scene API, flattened and uploaded to `device`.
int rth_world_generate(const char* kind, uint32_t seed, int n, int device, rth_world** out) {
    *out = nullptr;
    return guarded([&] {
        auto w = std::make_unique<rth_world>();
        recipe const rc = generate(kind, seed, n);
        std::vector<material const*> by_prim;
        w->world = rc.instantiate(w->mt, &by_prim);
        w->scene = std::make_unique<device_scene>(*w->world, device);
        w->desc = w->scene->tables().desc();
        w->number_prims(w->scene->tables(), by_prim);
        *out = w.release();
    });
}

// Builds the recipe's world through the scene API, flattens it and uploads it to `device`.
int rth_world_load(const char* recipe_path, int device, rth_world** out) {
    *out = nullptr;
    return guarded([&] {
        auto w = std::make_unique<rth_world>();
        recipe const rc = recipe::load(recipe_path);
        std::vector<material const*> by_prim;
        w->world = rc.instantiate(w->mt, &by_prim);
        w->scene = std::make_unique<device_scene>(*w->world, device);
        w->desc = w->scene->tables().desc();
        w->number_prims(w->scene->tables(), by_prim);
        *out = w.release();
    });
}

void rth_world_free(rth_world* w) { delete w; }
rt_scene* rth_world_scene(rth_world* w) { return w->scene->get(); }
const rt_scene_desc* rth_world_desc(rth_world* w) { return &w->desc; }

// The tables number primitives in the order the world was flattened (left-to-right leaf
// order); this gives, for each of them, its number in the recipe (construction order).
const int32_t* rth_world_recipe_ids(rth_world* w) { return w->recipe_id.data(); }

// Flattens a recipe's world without touching a GPU: tables only (for host-side checks).
int rth_world_tables(const char* recipe_path, rth_world** out) {
    *out = nullptr;
    return guarded([&] {
        auto w = std::make_unique<rth_world>();
        recipe const rc = recipe::load(recipe_path);
        std::vector<material const*> by_prim;
        w->world = rc.instantiate(w->mt, &by_prim);
        w->flat = std::make_unique<scene_flattener>();
        w->world->flatten_items(*w->flat);
        w->desc = w->flat->desc();
        w->number_prims(*w->flat, by_prim);
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
