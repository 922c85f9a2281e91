// recipe.hpp -- plain-data scene recipes: what a scene builder passes to the scene
// API's constructors, in construction order.  A recipe can be written to and read from
// a small text file, and instantiated into the object graph of rtiow.hpp.
//
//   rtiow-recipe 1
//   texture constant R G B | texture chequer EVEN ODD | texture perlin
//   material lambertian TEX | material metal R G B FUZZ | material dielectric RI
//   sphere MAT CX CY CZ R
//   moving_sphere MAT C0X C0Y C0Z C1X C1Y C1Z T0 T1 R
//   bvh_begin SEED T0 T1 ... primitives ... bvh_end
//   list_begin ... primitives ... list_end               (inside a bvh group only)
//
// Textures, materials and primitives are numbered from 0 in order of appearance.  The
// top level is an object_list; a bvh group is one bvh_node built over its objects with a
// fresh std::mt19937(SEED); a list group inside it is one object_list handed to that
// bvh_node as a single object.
#pragma once

#include <cstdint>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "rtiow.hpp"

namespace rtiow {

struct recipe {
    struct tex { int kind; double c[3]; int even, odd; };
    struct mat { int kind; int tex; double albedo[3]; double fuzz, ri; };
    struct prim { int kind; int mat; double c0[3], c1[3], t0, t1, r; int list = -1; };   // list: the list group it belongs to, -1 = none
    struct item { bool is_bvh; uint32_t seed; double t0, t1; int first, count; };

    std::vector<tex> textures;
    std::vector<mat> materials;
    std::vector<prim> prims;
    std::vector<item> items;
    int n_lists = 0;

    static recipe load(std::string const& path);   // throws std::runtime_error
    void save(std::string const& path) const;

    // Builds the world through the public constructors.  Materials are bound to `mt`
    // (every primitive gets its own material object, as the reference's ownership
    // requires); each bvh group uses its own generator.  When `material_by_prim` is given
    // it receives, per recipe primitive, the material object created for it.
    std::unique_ptr<object const> instantiate(std::mt19937& mt, std::vector<material const*>* material_by_prim = nullptr) const;

    int add_constant(double r, double g, double b);
    int add_lambertian(int tex);
    int add_metal(double r, double g, double b, double fuzz);
    int add_dielectric(double ri);
    void add_sphere(int mat, double x, double y, double z, double r);
    void add_moving_sphere(int mat, double const c0[3], double const c1[3], double t0, double t1, double r);
};

// main.cpp:52-59
recipe recipe_two_perlin_spheres();

// main.cpp:61-99 driven by std::mt19937(seed) with the draws in textual order.
// moving = false keeps every draw but builds static spheres; chequer = false replaces
// the ground's chequer texture by constant grey.  The bvh group's seed is the
// generator's next output after the last sphere.
recipe recipe_random_spheres(uint32_t seed, bool moving, bool chequer);

// n small spheres resting on a large ground sphere, centres uniform over a square patch
// whose side grows with sqrt(n) (1000 x 1000 at n = 10^6), radii in [0.2, 0.4), material
// mix 80/15/5 as main.cpp:81-93; one bvh group over everything.
recipe recipe_stress(uint32_t seed, int n);

}  // namespace rtiow
