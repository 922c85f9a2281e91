// rt_device.hpp -- gfx950 device code of the path-tracing hot path: ray/box and
// ray/sphere tests, BVH traversal, textures, scatter, camera rays and the bounce loop.
//
// Everything is IEEE double precision and the translation unit is built with
// -ffp-contract=off, so each product and sum rounds on its own in the order the
// reference's expressions evaluate (c++/vec3.cpp:24-74).  Reference citations are
// file:line under c++/ of hgiddens/ray-tracing-in-one-weekend.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtiow_hip.h"

namespace rtd {

// ---------------------------------------------------------------- scene layout ----
// Which leaves of the reference's tree a ray reaches depends only on the leaves' own boxes
// (see the flattener in rtiow_hip.hip), so the inner boxes only have to CONTAIN the leaf
// boxes below them.  A node of the device tree therefore holds the boxes of both children
// in single precision, rounded outward, and its slab test is a conservative one (node_test):
// it may let a ray into a subtree the reference would have skipped, never the reverse.  What
// the reference computes -- the double-precision test of aabb.cpp:8-57 -- is done on the box of
// each leaf (DLeaf) before its spheres are tried: the walk reaches a superset of the
// reference's leaves and that test removes the excess.  (In scenes without moving spheres the tree
// goes down to single spheres, each behind a box of its own, and the render kernel applies the
// leaf's test to the winning hit only: see world_closest_voted.)
//
// A leaf is a reference node whose children are primitives (bvh_node.cpp:31-37: one object
// twice, or two objects); its one or two spheres are entries 2*leaf and 2*leaf+1 of the
// primitive table.
struct __attribute__((aligned(16))) DNode {
    float bx[2][6];    // per child: min x, max x, min y, max y, min z, max z (plane_down / plane_up)
    int32_t child[2];  // >= 0 node index, < 0 leaf reference (see leaf_ref)
    uint32_t flags;    // NF_SAFE0 | NF_SAFE1 (a description of the subtrees; the walks decide by the leaf references' own bits)
    uint32_t pad;
};
static_assert(sizeof(DNode) == 64, "DNode is one 64-byte line");

struct DLeaf { double box[6]; };

// The four-wide tree of scenes without moving spheres: the binary tree with every other level folded
// away (flattener::collapse), so that a walk takes half as many dependent steps.  A node holds the boxes
// of up to four children, plane by plane -- the four children's min x, then their max x, min y, ... --
// so that the four planes a ray meets first on an axis are ONE 16-byte read at an address chosen by the
// direction's sign (the swap of aabb.cpp:16-18).  The boxes are the binary tree's (single precision,
// moved and rounded outward), the test is node_test's.  An empty slot has child = REF_NONE and an
// inverted box.  128 bytes: one cache line per step for trees that live in device memory.
struct __attribute__((aligned(16))) DNode4 {
    float pl[6][4];     // [min x, max x, min y, max y, min z, max z][child]
    int32_t child[4];   // >= 0 DNode4 index, < 0 leaf reference, REF_NONE = no child
    uint32_t pad[4];
};
static_assert(sizeof(DNode4) == 128, "DNode4 is one 128-byte line");

// A plane of an inner box in single precision: moved outward by 2^-22 of its magnitude (the part of
// the conservative test's error bound that is proportional to plane / d, see make_ctx), then
// rounded outward.
constexpr double kPlaneWiden = 0x1p-22;
__host__ __device__ inline float plane_down(double v) {
    v -= kPlaneWiden * (v < 0 ? -v : v);
    float f = (float)v;
    if ((double)f > v) f = nextafterf(f, -HUGE_VALF);
    return f;
}
__host__ __device__ inline float plane_up(double v) {
    v += kPlaneWiden * (v < 0 ? -v : v);
    float f = (float)v;
    if ((double)f < v) f = nextafterf(f, HUGE_VALF);
    return f;
}

// The box a single static sphere gets in trees that go down to single spheres (scenes without moving
// spheres): the radius, 2^-20 of it, and 2^-30 of the largest coordinate (a radius of zero, or one far below
// the centre's magnitude, still gets a box wider than the sphere test's rounding), and, when the scene allows
// it (far_allowance > 0), what the rounding of the reference's discriminant can add for a ray that starts
// up to far_allowance from the centre (see FarSlack).  b = min x, max x, min y, ...
__host__ __device__ inline double far_slack_pad(double radius, double far_allowance) {
    if (!(far_allowance > 0)) return 0;
    const double a = 0x1p-49 * far_allowance * far_allowance / radius, b = 0x1p-23 * far_allowance;
    return (a < b ? a : b) * (1 + 0x1p-10);   // (a NaN -- a zero radius -- takes b)
}
__host__ __device__ inline void sphere_pad_box(const double *c0, double radius, double far_allowance, double *b) {
    const double ax = c0[0] < 0 ? -c0[0] : c0[0], ay = c0[1] < 0 ? -c0[1] : c0[1], az = c0[2] < 0 ? -c0[2] : c0[2];
    const double far = ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az);
    const double ar = radius < 0 ? -radius : radius;
    const double r = ar * (1.0 + 0x1p-20) + 0x1p-30 * (far + ar) + far_slack_pad(ar, far_allowance);
    for (int a = 0; a < 3; ++a) { b[2 * a] = c0[a] - r; b[2 * a + 1] = c0[a] + r; }
}

// Every primitive under a SAFE child stays inside the child's box (static spheres), so
// the descent may be skipped once the box starts beyond the best hit.  A moving sphere
// leaves the box the reference gives it (moving_sphere.cpp:58), but not by more than the scene's
// reach (DScene::reach_a, reach_b): the node test skips a box when, grown by the reach on its entry
// side, it starts beyond the best hit.  (The flags are kept for the leaf steps, which test the box
// of a leaf with a moving sphere up front.)
enum : uint32_t { NF_SAFE0 = 1u, NF_SAFE1 = 2u };

// leaf reference: bit 31 set; bits 0-25 leaf number; bit 26: the leaf's second sphere alone (trees over
// single spheres, see flattener::split_pairs); bit 27: all its spheres are static; bit 28: both
// spheres of the leaf; bits 29, 30: first / second sphere is a moving sphere
constexpr int32_t LEAF_BIT = (int32_t)0x80000000u;
constexpr int32_t LEAF_SAFE = 1 << 27, LEAF_TWO = 1 << 28, LEAF_MOVING0 = 1 << 29, LEAF_MOVING1 = 1 << 30;
constexpr int32_t LEAF_SECOND = 1 << 26;             // without LEAF_TWO: the reference is to the leaf's second sphere alone
constexpr int32_t LEAF_INDEX_MASK = 0x03FFFFFF;
constexpr int32_t REF_NONE = (int32_t)0x83FFFFFFu;   // a leaf number no scene can reach

enum : uint32_t { PK_STATIC = 0, PK_MOVING = 1, PK_MOVING_UNIT = 2 };

struct __attribute__((aligned(16))) DPrim {
    double c0[3]; double r;    // centre at time0, radius
    double dc[3]; double t0;   // centre1 - centre0, time0
    double t1; uint32_t mat; uint32_t kind;
    int32_t orig; uint32_t pad[3];
};
static_assert(sizeof(DPrim) == 96, "DPrim layout");

// One entry of the world's object_list.  ref is a node index or a leaf reference;
// has_box = 0 for a bare primitive in the list (object_list.cpp:18 tries it directly).
// ref4: the same item in the four-wide tree (a DNode4 index, or the same leaf reference), where the scene has one.
struct DItem { int32_t ref; uint32_t has_box; int32_t ref4; uint32_t pad; };
// inv_ri = 1 / ri and r0sq = ((1 - ri) / (1 + ri))^2, as material.cpp:81 and :27-28 form them on every
// call, are formed once at upload
struct DMat { uint32_t kind; uint32_t tex; double albedo[3]; double fuzz; double ri; double inv_ri; double r0sq; };
struct DTex { uint32_t kind; uint32_t even, odd; uint32_t pad; double c[3]; };

// How far the line of a ray may pass from a sphere that the reference's test (sphere.cpp:9-22) still
// hits.  The test asks for a positive discriminant b*b - a*c, and the rounding of b, a and c leaves
// that quantity uncertain by at most 16 ulp of a*L*L, L the distance from the ray's origin to the
// centre: a sphere of radius r is "hit" by lines passing within sqrt(r*r + 2^-49 L*L) of its centre,
// i.e. up to min(2^-50 L*L/r, 2^-24.5 L) outside it.  Trees whose leaves are single spheres behind
// their own boxes (scenes without moving spheres) must let such rays through: every segment widens
// the boxes it tests by that amount, with L bounded by the distance from its origin to the centre of
// the box around the static spheres' centres plus that box's half-diagonal, and r by their smallest
// radius.  The same amount keeps the skip behind the best hit honest in every scene: the point where
// such a line comes closest to the centre lies outside the sphere's box by at most this slack.  For
// origins inside a scene like the book's it is far below the boxes' fixed padding; it matters for
// origins 10^5 radii away and more.  k = 0 (no static sphere in the scene): no widening.
struct FarSlack {
    float cx, cy, cz;   // centre of the world's box
    float h;            // its half-diagonal, rounded up
    float k;            // 2^-49 / smallest radius, rounded up; 0 = the scene needs none
};

struct DNode4;
struct DScene {
    const DNode *nodes;
    const DNode4 *nodes4;     // the four-wide tree of scenes without moving spheres (null: none)
    const DLeaf *leaves;
    const DPrim *prims;       // 2 per leaf
    const DItem *items;
    const DMat *mats;
    const DTex *texs;
    const double *ranfloat;   // 256
    const uint8_t *perm;      // 3 x 256: x, y, z
    uint32_t n_items;
    uint32_t stack_depth;     // entries per lane
    uint32_t recheck_all;     // test switch: treat every postponed box test as failed (see world_closest_voted)
    uint32_t wide_radiance;   // some albedo or texture colour lies outside [0, 1]: a sample's radiance may leave [0, 1]
    FarSlack far;
    // How far a moving sphere can be from the box the reference gives it (moving_sphere.cpp:51-60
    // covers the start position only, and shutter times reach outside [time0, time1]: camera.cpp:41).
    // Its centre at ray time t is c0 + s dc with s = (t - time0) / (time1 - time0); along axis k it is
    // at most reach_a[k] |t| + reach_b[k] away from c0, which the box contains together with the
    // radius around it: every point of the moved sphere lies within the box grown by that much per
    // axis.  Rounded up; 0 in scenes without moving spheres.
    float reach_a[3], reach_b[3];
};

struct DCamera {
    double origin[3], u[3], v[3], w[3], llc[3], horizontal[3], vertical[3];
    double lens_radius, t0, t1;
};

// ------------------------------------------------------------------------ vec3 ----
struct v3 { double x, y, z; };
__device__ __forceinline__ v3 mk(double x, double y, double z) { v3 r = {x, y, z}; return r; }
__device__ __forceinline__ v3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ v3 vadd(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 vmul(v3 a, double d) { return mk(a.x * d, a.y * d, a.z * d); }
__device__ __forceinline__ v3 vdiv(v3 a, double d) { return mk(a.x / d, a.y / d, a.z / d); }
__device__ __forceinline__ v3 vneg(v3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double vlen(v3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ v3 vunit(v3 a) { return vdiv(a, vlen(a)); }

struct Ray { v3 o, d; double time; };

// Division by a denominator that many divisions share.  The IEEE division sequence of the hardware
// (v_div_scale, v_rcp, two Newton steps on the reciprocal, quotient, remainder, v_div_fmas, v_div_fixup)
// spends most of its instructions on the denominator alone.  refined_rcp is that part; with it, and
// when nothing needs scaling or fixing up (finite operands, quotient well inside the normal range, or
// a zero numerator), the rest of the sequence is q = n y corrected once by its own remainder -- the
// same operations on the same values, hence the same correctly rounded quotient
// (tools/division_check.hip compares the two over 10^9 numerators for image sizes up to 2^16).
__device__ __forceinline__ double refined_rcp(double d) {
    double y = __builtin_amdgcn_rcp(d);
    y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
    y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
    return y;
}
__device__ __forceinline__ double div_by_shared(double n, double d, double rcp_d) {
    const double q = n * rcp_d;
    return __builtin_fma(__builtin_fma(-d, q, n), rcp_d, q);
}

// per-segment constants of the slab and sphere tests
struct RayCtx {
    v3 o, d;
    double time;
    double ix, iy, iz;   // 1/d per axis (aabb.cpp:12,27,42)
    // 1/d < 0 -> the slab's two planes swap (aabb.cpp:16-18).  Kept as the byte offset of
    // the plane the ray meets first inside a (min, max) pair: 0 or 8.
    uint32_t nx8, ny8, nz8;
    double a;            // d.d (sphere.cpp:14)
    // single-precision constants of the conservative node test: t = plane * i + c with i = 1/d and
    // c = -o/d per axis, c lowered for the planes the ray meets first and raised for the others
    // (see node_test)
    float ixf, iyf, izf, cnx, cny, cnz, cfx, cfy, cfz;
    // by how much, in t, a moving sphere's hit may come before the ray's entry into its box, per axis
    // (the reach of the scene's moving spheres at this ray's time over the speed along the axis)
    float rnx, rny, rnz;
};

// The conservative test computes t = fma(plane, i, c) in single precision.  Against the exact
// (plane - o) / d = plane / d + C: i carries a relative rounding error of 2^-24, c two of them (its
// own and that of the shift below), the fma another on the sum, so
// |t - exact| <= 2^-23 |plane / d| + 1.5 * 2^-23 |C| (1 + 2^-21).  Both parts are paid in advance:
// the stored planes lie outward by 2^-22 |plane| (plane_down / plane_up), and the constant is
// lowered (entry side) or raised (exit side) by K * sum|c| with K = 2^-22 -- a sum, so that one
// amount serves all three axes.  tools/conservative_check.hip hunts for counter-examples.
constexpr float kWidenK = 0x1p-22f;
// a bound rounded away from the interval: x lowered (raised) by at least its conversion error
__device__ __forceinline__ float below(double x) { const float f = (float)x; return f - __builtin_fabsf(f) * 0x1p-22f; }
__device__ __forceinline__ float above(double x) { const float f = (float)x; return f + __builtin_fabsf(f) * 0x1p-22f; }


// `reach`: reach_a[3], reach_b[3] of the scene (null: no moving spheres).
__device__ __forceinline__ RayCtx make_ctx(const Ray &r, const FarSlack &far, const float *reach = nullptr) {
    RayCtx c;
    c.o = r.o; c.d = r.d; c.time = r.time;
    c.ix = 1.0 / r.d.x; c.iy = 1.0 / r.d.y; c.iz = 1.0 / r.d.z;
    c.nx8 = c.ix < 0 ? 8u : 0u; c.ny8 = c.iy < 0 ? 8u : 0u; c.nz8 = c.iz < 0 ? 8u : 0u;
    c.a = vdot(r.d, r.d);
    c.ixf = (float)c.ix; c.iyf = (float)c.iy; c.izf = (float)c.iz;
    const float cx = (float)-(r.o.x * c.ix), cy = (float)-(r.o.y * c.iy), cz = (float)-(r.o.z * c.iz);
    // A zero, huge or NaN direction component makes an infinite or NaN constant; the test then
    // yields infinite or NaN bounds for that axis, and those drop out or pass: such rays simply
    // visit more of the tree.  The error bound is relative, which single precision only honours
    // away from its subnormal range: the floor 2^-100 covers products and constants that underflow,
    // and a reciprocal below 2^-100 (a direction component beyond 2^100) or one that overflows single
    // precision widens without limit.
    // (the sums bound every |c| and |i| and, unlike a maximum, hand a NaN on)
    const float smallest = __builtin_fminf(__builtin_fminf(__builtin_fabsf(c.ixf), __builtin_fabsf(c.iyf)), __builtin_fabsf(c.izf));
    const float all = __builtin_fabsf(c.ixf) + __builtin_fabsf(c.iyf) + __builtin_fabsf(c.izf);
    const float widen = kWidenK * (__builtin_fabsf(cx) + __builtin_fabsf(cy) + __builtin_fabsf(cz)) +
                        (smallest >= 0x1p-100f && all < HUGE_VALF ? 0x1p-100f : HUGE_VALF);
    float rx = 0, ry = 0, rz = 0, reach_all = 0;
    if (reach) {
        const float at = __builtin_fabsf((float)r.time) * (1.0f + 0x1p-20f);
        rx = __builtin_fmaf(reach[0], at, reach[3]); ry = __builtin_fmaf(reach[1], at, reach[4]); rz = __builtin_fmaf(reach[2], at, reach[5]);
        reach_all = rx + ry + rz;
    }
    // the spatial slack of the sphere test for this origin (see FarSlack), per axis in units of t
    float wx = widen, wy = widen, wz = widen;
#ifndef RT_NO_FAR_SLACK
    if (far.k > 0) {
        const float dx = (float)r.o.x - far.cx, dy = (float)r.o.y - far.cy, dz = (float)r.o.z - far.cz;
        const float L = __builtin_sqrtf(dx * dx + dy * dy + dz * dz) * (1.0f + 0x1p-20f) + far.h + reach_all;
        const float slack = __builtin_fminf(far.k * L * L, 0x1p-23f * L) * (1.0f + 0x1p-20f);   // (NaN or infinite L: infinite)
        wx += slack * __builtin_fabsf(c.ixf); wy += slack * __builtin_fabsf(c.iyf); wz += slack * __builtin_fabsf(c.izf);
    }
#endif
    c.cfx = cx + wx; c.cfy = cy + wy; c.cfz = cz + wz;
    c.cnx = cx - wx; c.cny = cy - wy; c.cnz = cz - wz;
    // (0 * inf gives NaN, which drops out of the skip test's maximum: no constraint from that axis)
    c.rnx = rx * __builtin_fabsf(c.ixf) * (1.0f + 0x1p-20f); c.rny = ry * __builtin_fabsf(c.iyf) * (1.0f + 0x1p-20f); c.rnz = rz * __builtin_fabsf(c.izf) * (1.0f + 0x1p-20f);
    return c;
}

// ------------------------------------------------------------------- random ----
// Per-path stream: Philox4x32-10 keyed by the render seed over the counter
// (pixel, sample) gives the start state of a xoshiro128++ generator; a double takes
// the top 53 bits of two successive 32-bit outputs.
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c[0]), l0 = 0xD2511F53u * c[0];
        uint32_t h1 = __umulhi(0xCD9E8D57u, c[2]), l1 = 0xCD9E8D57u * c[2];
        uint32_t n0 = h1 ^ c[1] ^ k0, n2 = h0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = l1; c[2] = n2; c[3] = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

struct StreamRng {
    uint32_t s0, s1, s2, s3;
    __device__ __forceinline__ void init(uint64_t seed, uint32_t pixel, uint32_t sample) {
        uint32_t c[4] = {pixel, sample, 0x52544957u, 0u};
        philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c);
        s0 = c[0]; s1 = c[1]; s2 = c[2]; s3 = c[3];
        if ((s0 | s1 | s2 | s3) == 0) s0 = 1u;
    }
    __device__ __forceinline__ uint32_t next() {
        uint32_t t0 = s0 + s3;
        uint32_t result = ((t0 << 7) | (t0 >> 25)) + s0;
        uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 11) | (s3 >> 21);
        return result;
    }
    // The draw is k * 2^-53 with k = (hi:lo) >> 11, a 53-bit integer.  It is formed without an
    // integer-to-double conversion: the low 52 bits of k are the fraction of a double x in
    // [1, 2), and with top = bit 52 of k, k * 2^-53 = (x - 1 + top) / 2.  Every step is exact.
    __device__ __forceinline__ double twice_canonical() {
        const uint32_t hi = next();
        const uint32_t lo = next();
        const uint32_t xlo = __builtin_amdgcn_alignbit(hi, lo, 11);
        const uint32_t xhi = 0x3FF00000u | ((hi >> 11) & 0xFFFFFu);
        const double x = __hiloint2double((int)xhi, (int)xlo);
        return x + ((hi >> 31) ? 0.0 : -1.0);
    }
    __device__ __forceinline__ double canonical() { return twice_canonical() * 0.5; }
    // uniform_real_distribution{-1, 1}: canonical * 2 + -1, where canonical * 2 is exact
    __device__ __forceinline__ double symmetric() { return twice_canonical() + -1.0; }
    __device__ __forceinline__ bool exhausted() const { return false; }
    // a stream owes nobody a particular draw count: draws whose values cannot reach the
    // result may be skipped (see camera_ray)
    static constexpr bool kMaySkipDeadDraws = true;

    // The correctly rounded single-precision square root of x in [1, 4): the hardware's v_sqrt_f32 is good to
    // one unit in the last place; its neighbours are tried against x by the sign of an exactly computed
    // residual (s_down s < x <= s s_up brackets the rounding boundaries: between s_down s and the square of
    // the midpoint lies no single-precision number).
    __device__ __forceinline__ static float sqrt_rounded(float x) {
        float s = __builtin_amdgcn_sqrtf(x);
        const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
        const float below_x = __builtin_fmaf(-down, s, x), above_x = __builtin_fmaf(-up, s, x);
        if (below_x <= 0.0f) s = down;
        if (above_x > 0.0f) s = up;
        return s;
    }

    // material.cpp:35-40 asks for a point uniform in the unit sphere and gets it by rejection from the cube:
    // 1.9 attempts on average, and a wave waits for its unluckiest lane (6.5 attempts).  A path's own stream
    // owes nobody a particular way of spending its outputs, only the distribution:
    __device__ __forceinline__ void in_unit_sphere(double &x, double &y, double &z) {
        // no rejection: the direction by the equal-area octahedral map of the square onto the sphere (Clarberg,
        // "Fast equal-area mapping of the (hemi)sphere using SIMD", 2008) in single precision -- 24 bits per
        // coordinate of the square, a unit vector to 2^-22 -- and the radius as the largest of three 32-bit
        // uniforms (density 3 r^2), shortened by 2^-21 so that the point stays inside the unit sphere.  Five
        // generator outputs, no loop, every operation correctly rounded (the oracle repeats them one for one).
        const uint32_t ha = next(), hb = next();
        const float a = __builtin_fmaf((float)(ha >> 8), 0x1p-23f, 0x1p-24f - 1.0f), b = __builtin_fmaf((float)(hb >> 8), 0x1p-23f, 0x1p-24f - 1.0f);   // exact
        const float aa = __builtin_fabsf(a), ab = __builtin_fabsf(b);
        const float d = 1.0f - (aa + ab), r = 1.0f - __builtin_fabsf(d);
        const float th = (r > 0 ? __fdiv_rn(ab - aa, r) : 0.0f) * 0.78539816339744830962f, q = th * th;
        const float f = r * sqrt_rounded(__builtin_fmaf(-r, r, 2.0f)) * 0.70710678118654752440f;   // (2 - r r lies in [1, 2])
        // cos and sin of th, |th| <= pi/4 (Taylor polynomials: errors below 1e-7)
        float c = 2.4801587e-05f;
        c = __builtin_fmaf(c, q, -0.0013888889f); c = __builtin_fmaf(c, q, 0.041666668f); c = __builtin_fmaf(c, q, -0.5f); c = __builtin_fmaf(c, q, 1.0f);
        float sn = 2.7557319e-06f;
        sn = __builtin_fmaf(sn, q, -0.00019841270f); sn = __builtin_fmaf(sn, q, 0.0083333333f); sn = __builtin_fmaf(sn, q, -0.16666667f); sn = __builtin_fmaf(sn * q, th, th);
        const float fx = __builtin_copysignf(f * (c - sn), a), fy = __builtin_copysignf(f * (c + sn), b), fz = __builtin_copysignf(__builtin_fmaf(-r, r, 1.0f), d);
        const uint32_t w0 = next(), w1 = next(), w2 = next();
        const uint32_t w = w0 > w1 ? (w0 > w2 ? w0 : w2) : (w1 > w2 ? w1 : w2);
        const double rho = __builtin_fma((double)w, 0x1p-32 * (1.0 - 0x1p-21), 0x1p-33);
        x = (double)fx * rho; y = (double)fy * rho; z = (double)fz * rho;
    }
};

// uniforms handed in by the caller, consumed in order
struct TapeRng {
    const double *tape;
    int len, pos;
    bool over;
    __device__ __forceinline__ double canonical() {
        double u = 0.0;
        if (pos < len) u = tape[pos]; else over = true;
        ++pos;
        return u;
    }
    __device__ __forceinline__ double symmetric() { return (canonical() * (1.0 - -1.0)) + -1.0; }
    __device__ __forceinline__ bool exhausted() const { return over; }
    // a tape replays the reference's draws one for one
    static constexpr bool kMaySkipDeadDraws = false;
};

// ------------------------------------------------------------------- box test ----
// aabb::hit (aabb.cpp:8-57).  b = min x, max x, min y, max y, min z, max z.  The bounds
// only ever tighten and a NaN slab value (0 * inf) leaves them unchanged, so testing
// once after the third axis decides exactly as the reference's three early exits do.
__device__ __forceinline__ bool box_test(const double *b, const RayCtx &r, double tmin, double tmax, double &entry) {
    double t0 = (b[0] - r.o.x) * r.ix, t1 = (b[1] - r.o.x) * r.ix;
    tmin = __builtin_fmax(r.nx8 ? t1 : t0, tmin);
    tmax = __builtin_fmin(r.nx8 ? t0 : t1, tmax);
    t0 = (b[2] - r.o.y) * r.iy; t1 = (b[3] - r.o.y) * r.iy;
    tmin = __builtin_fmax(r.ny8 ? t1 : t0, tmin);
    tmax = __builtin_fmin(r.ny8 ? t0 : t1, tmax);
    t0 = (b[4] - r.o.z) * r.iz; t1 = (b[5] - r.o.z) * r.iz;
    tmin = __builtin_fmax(r.nz8 ? t1 : t0, tmin);
    tmax = __builtin_fmin(r.nz8 ? t0 : t1, tmax);
    entry = tmin;
    return !(tmax <= tmin);
}

// The same test reading each slab's two planes in the order the ray meets them: the
// swap of aabb.cpp:16-18 becomes a choice of address, which costs nothing when every
// lane reads its own node anyway (LDS-resident tree).
__device__ __forceinline__ bool box_test_ordered(const char *b, const RayCtx &r, double tmin, double tmax, double &entry) {
    const double x0 = *reinterpret_cast<const double *>(b + r.nx8), x1 = *reinterpret_cast<const double *>(b + (r.nx8 ^ 8u));
    const double y0 = *reinterpret_cast<const double *>(b + 16 + r.ny8), y1 = *reinterpret_cast<const double *>(b + 16 + (r.ny8 ^ 8u));
    const double z0 = *reinterpret_cast<const double *>(b + 32 + r.nz8), z1 = *reinterpret_cast<const double *>(b + 32 + (r.nz8 ^ 8u));
    tmin = __builtin_fmax((x0 - r.o.x) * r.ix, tmin);
    tmax = __builtin_fmin((x1 - r.o.x) * r.ix, tmax);
    tmin = __builtin_fmax((y0 - r.o.y) * r.iy, tmin);
    tmax = __builtin_fmin((y1 - r.o.y) * r.iy, tmax);
    tmin = __builtin_fmax((z0 - r.o.z) * r.iz, tmin);
    tmax = __builtin_fmin((z1 - r.o.z) * r.iz, tmax);
    entry = tmin;
    return !(tmax <= tmin);
}

// ---------------------------------------------------------------- sphere test ----
// moving_sphere.cpp:3-5
__device__ __forceinline__ v3 prim_centre(const DPrim *p, uint32_t moving, double time) {
    v3 c0 = ld3(p->c0);
    if (!moving) return c0;
    double s = (p->kind == PK_MOVING_UNIT) ? time : (time - p->t0) / (p->t1 - p->t0);
    return vadd(c0, vmul(ld3(p->dc), s));
}

// sphere.cpp:9-45, moving_sphere.cpp:13-49: the root the reference accepts in
// (tmin, tmax), nearer root first
// (radius2 = radius * radius, as sphere.cpp:16 forms it)
// (in two halves: the discriminant, and the roots of a positive one)
__device__ __forceinline__ void sphere_disc(v3 c, double radius2, const RayCtx &r, double &b, double &disc) {
    v3 oc = vsub(r.o, c);
    b = vdot(oc, r.d);
    double cc = vdot(oc, oc) - radius2;
    disc = b * b - r.a * cc;
}
__device__ __forceinline__ bool sphere_roots(double b, double disc, const RayCtx &r, double tmin, double tmax, double &t_out) {
    double root = sqrt(disc);
    double t = (-b - root) / r.a;
    bool hit = t < tmax && t > tmin;
    if (!hit) {   // the farther root only when the nearer one is not accepted (sphere.cpp:31)
        t = (-b + root) / r.a;
        hit = t < tmax && t > tmin;
    }
    t_out = t;
    return hit;
}
__device__ __forceinline__ bool sphere_root(v3 c, double radius2, const RayCtx &r, double tmin, double tmax, double &t_out) {
    double b, disc;
    sphere_disc(c, radius2, r, b, disc);
    return disc > 0 && sphere_roots(b, disc, r, tmin, tmax, t_out);
}

// Lane-utilisation counters of the render kernel, compiled in with -DRT_LANE_STATS (the
// "lanestats" make target) and empty otherwise.  Phase k counts how often a wave executed it
// (c[2k]) and with how many lanes (c[2k+1]): 0 node step, 1 leaf step, 2 round (lanes with a live
// path), 3 hit shading, 4 sky, 5 claiming new paths.
struct LaneStats {
#ifdef RT_LANE_STATS
    unsigned long long *c = nullptr;   // 12 slots of the wave in device memory; null = not counting
    // whichever lanes are active here, exactly one of them books the execution for the wave
    __device__ __forceinline__ void add(int phase, uint32_t lanes) {
        const unsigned long long active = __ballot(true);
        if (c && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)active) - 1u) { c[2 * phase] += 1; c[2 * phase + 1] += lanes; }
    }
#else
    __device__ __forceinline__ void add(int, uint32_t) {}
#endif
};

// ------------------------------------------------------------------ traversal ----
// bvh_node::hit (bvh_node.cpp:52-69) as a walk with an explicit stack.  The reference
// visits both children of every node whose box is hit, with the interval it was given,
// and on equal t keeps the right-hand result: "the smallest t, and among equal t the
// last leaf in left-to-right order".  Primitives are numbered in that order, so any
// visiting order gives the same answer.
struct Trav {
    int32_t cur;     // node index, leaf reference, or REF_NONE when the walk is over
    int32_t sp;      // stack entries in use
    double bt;       // best hit of the item being walked
    int32_t bp;      // its primitive (device order), -1 = none
    float bt_above;  // bt rounded up, for the conservative node test; +inf while there is no hit
};

// A lane's traversal stack.  Entry k lives in LDS at lds[k * stride] (the block's lanes are
// interleaved) while k < cap; deeper entries, which only trees too large for LDS produce,
// live in device memory at spill[(k - cap) * spill_stride].
struct LaneStack {
    int *lds;
    int stride;
    int cap;
    int *spill;
    int spill_stride;
};

template <bool SPILL> __device__ __forceinline__ void stack_put(const LaneStack &st, int k, int32_t v) {
    if (!SPILL || k < st.cap) st.lds[k * st.stride] = v;
    else st.spill[(size_t)(k - st.cap) * st.spill_stride] = v;
}
template <bool SPILL> __device__ __forceinline__ int32_t stack_get(const LaneStack &st, int k) {
    if (!SPILL || k < st.cap) return st.lds[k * st.stride];
    return st.spill[(size_t)(k - st.cap) * st.spill_stride];
}

template <bool SPILL> __device__ __forceinline__ int32_t trav_pop(Trav &tv, const LaneStack &st) {
    int32_t r = REF_NONE;
    if (tv.sp > 0) { --tv.sp; r = stack_get<SPILL>(st, tv.sp); }
    return r;
}

template <bool SPILL>
__device__ __forceinline__ void node_decide(int32_t c0, int32_t c1, bool h0, bool h1, float e0, float e1, const LaneStack &st, Trav &tv) {
    const bool near0 = e0 <= e1;
    if (h0 && h1) {
        stack_put<SPILL>(st, tv.sp, near0 ? c1 : c0);
        ++tv.sp;
        tv.cur = near0 ? c0 : c1;
    } else if (h0 || h1) {
        tv.cur = h0 ? c0 : c1;
    } else {
        tv.cur = trav_pop<SPILL>(tv, st);
    }
}

// The conservative slab test of one child box.  n* are the planes the ray meets first on each
// axis, f* the ones it meets last (single precision, moved and rounded outward).  Returns false
// only when the exact test of aabb.cpp:8-57 on the box inside would fail: the planes and the
// constants carry the rounding error bound in advance (see make_ctx), a NaN slab value (0 * inf)
// drops out of max3/min3 as it leaves the reference's bounds unchanged, and NaN bounds pass.
// `entry` is a lower bound of the exact entry distance.
// `limit` is the best hit so far rounded up (+inf for none).  A box that starts beyond it is skipped.
// STILL (every sphere stays in its box): the limit rides along in the three-operand minimum for free.
// Otherwise the spheres of the box may be up to the scene's reach outside it, and the box is skipped
// when the box grown by the reach on its entry side starts beyond the limit (the hit point of a
// moved sphere lies inside the grown box, hence behind that entry).
template <bool STILL>
__device__ __forceinline__ bool node_test(float nx, float fx, float ny, float fy, float nz, float fz, const RayCtx &r,
                                          float tmin_below, float tmax_above, float limit, float &entry) {
    const float tnx = __builtin_fmaf(nx, r.ixf, r.cnx), tfx = __builtin_fmaf(fx, r.ixf, r.cfx);
    const float tny = __builtin_fmaf(ny, r.iyf, r.cny), tfy = __builtin_fmaf(fy, r.iyf, r.cfy);
    const float tnz = __builtin_fmaf(nz, r.izf, r.cnz), tfz = __builtin_fmaf(fz, r.izf, r.cfz);
    const float e_lo = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
    const float x_hi = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
    entry = e_lo;
    // v_min3_f32 / v_max_f32 as they are (NaN operands drop out): the builtins would
    // first re-canonicalise the two interval bounds in every step, against signalling NaNs that
    // nothing here produces
    float x_in, e_in;
    if (STILL) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(x_in) : "v"(x_hi), "v"(tmax_above), "v"(limit));
    else asm("v_min_f32 %0, %1, %2" : "=v"(x_in) : "v"(x_hi), "v"(tmax_above));
    asm("v_max_f32 %0, %1, %2" : "=v"(e_in) : "v"(e_lo), "v"(tmin_below));
    bool hit = !(x_in < e_in);
    if (!STILL) {
        float e_grown;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(e_grown) : "v"(tnx - r.rnx), "v"(tny - r.rny), "v"(tnz - r.rnz));
        hit = hit && !(limit < e_grown);
    }
    return hit;
}

// One node read from device memory: test both child boxes, continue with the nearer, push the
// farther.  A child is also skipped when it starts beyond the best hit so far (tv.bt_above, +inf
// while there is none); see node_test.  STILL: the scene has no moving sphere.
template <bool SPILL, bool STILL = false>
__device__ __forceinline__ void node_step(const DNode *nodes, const RayCtx &r, float tmin_below, float tmax_above, const LaneStack &st, Trav &tv) {
    const DNode *nd = nodes + tv.cur;
    const int32_t c0 = nd->child[0], c1 = nd->child[1];
    float b[2][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { b[0][k] = nd->bx[0][k]; b[1][k] = nd->bx[1][k]; }
    float e[2];
    bool h[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        h[c] = node_test<STILL>(r.nx8 ? b[c][1] : b[c][0], r.nx8 ? b[c][0] : b[c][1], r.ny8 ? b[c][3] : b[c][2], r.ny8 ? b[c][2] : b[c][3],
                         r.nz8 ? b[c][5] : b[c][4], r.nz8 ? b[c][4] : b[c][5], r, tmin_below, tmax_above, tv.bt_above, e[c]);
    }
    node_decide<SPILL>(c0, c1, h[0], h[1], e[0], e[1], st, tv);
}

// Records of the LDS copy of the tree.  The strides are chosen so that lanes reading the
// same field of different records spread over the 64 banks: 17 dwords for the 32-bit reads
// of a node, 58 dwords (a factor 2, which a 64-bit read covers) for a leaf.
constexpr uint32_t LNODE_STRIDE = 68;    // bx[2][6] floats | child[2] | flags
constexpr uint32_t LPRIM_BYTES = 88;     // c0[3] r*r dc[3] t0 t1 | mat kind orig pad
constexpr uint32_t LLEAF_STRIDE = 232;   // box[6] doubles | sphere | sphere | pad
// scenes without moving spheres: box[6] | c0[3] r*r | c0[3] r*r | pad (an odd number of 8-byte words)
constexpr uint32_t SPRIM_BYTES = 32, SLEAF_STRIDE = 120;

// Where a walk finds the tree.  The block's LDS holds the first n_lnodes nodes (all of them
// when the tree fits, else its top levels: nodes are numbered breadth-first there) and,
// when lleaves is set, every leaf; the rest is read from device memory.
struct TreeView {
    const DNode *nodes;     // device memory
    const DLeaf *leaves;
    const DPrim *prims;
    const char *lnodes;     // LDS copies
    const char *lleaves;
    int32_t n_lnodes;
};

// Where a ray finds the planes of an LDS node record, in the order it meets them (the swap of
// aabb.cpp:16-18 becomes a choice of address, free when every lane reads its own node anyway):
// the addresses of child 0's near and far plane per axis in record 0.  Child 1's planes lie 24
// bytes further, so each pair is one two-address LDS read off "plane address + record offset".
//
// They are kept as 32-bit LDS addresses that the compiler cannot see into: it would otherwise split
// the constant part of the tree's position off each of them, to fold it into the read's offset
// field, find that a two-address read has no room for it, and add it back in every step.
struct PlaneAddr {
    uint32_t xn, xf, yn, yf, zn, zf;
};
typedef __attribute__((address_space(3))) const char lds_cchar;
typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ uint32_t lds_address(const char *p) {
    uint32_t a = (uint32_t)(uintptr_t)(lds_cchar *)p;
    asm("" : "+v"(a));
    return a;
}
__device__ __forceinline__ float lds_float(uint32_t address) { return *(lds_cfloat *)(uintptr_t)address; }
__device__ __forceinline__ PlaneAddr plane_addresses(const char *lnodes, const RayCtx &r) {
    const uint32_t ox = r.nx8 >> 1, oy = r.ny8 >> 1, oz = r.nz8 >> 1;   // 0 or 4: byte offset of the nearer plane in a pair
    PlaneAddr a = {lds_address(lnodes + ox), lds_address(lnodes + (ox ^ 4u)), lds_address(lnodes + 8 + oy),
                   lds_address(lnodes + 8 + (oy ^ 4u)), lds_address(lnodes + 16 + oz), lds_address(lnodes + 16 + (oz ^ 4u))};
    return a;
}

// node_step on the LDS copy.  PREMUL: node references are byte offsets of the records already
// (trees that live in LDS whole), not indices.
template <bool SPILL, bool PREMUL, bool STILL>
__device__ __forceinline__ void node_step_lds(const char *lnodes, const PlaneAddr &pa, const RayCtx &r, float tmin_below, float tmax_above, const LaneStack &st, Trav &tv) {
    const uint32_t off = PREMUL ? (uint32_t)tv.cur : (uint32_t)tv.cur * LNODE_STRIDE;
    const char *nd = lnodes + off;
    const int32_t c0 = *reinterpret_cast<const int32_t *>(nd + 48), c1 = *reinterpret_cast<const int32_t *>(nd + 52);
    float e[2];
    bool h[2];
    float pl[2][6];   // all twelve planes first
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        pl[c][0] = lds_float(pa.xn + off + 24 * c); pl[c][1] = lds_float(pa.xf + off + 24 * c);
        pl[c][2] = lds_float(pa.yn + off + 24 * c); pl[c][3] = lds_float(pa.yf + off + 24 * c);
        pl[c][4] = lds_float(pa.zn + off + 24 * c); pl[c][5] = lds_float(pa.zf + off + 24 * c);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        h[c] = node_test<STILL>(pl[c][0], pl[c][1], pl[c][2], pl[c][3], pl[c][4], pl[c][5], r, tmin_below, tmax_above, tv.bt_above, e[c]);
    }
    node_decide<SPILL>(c0, c1, h[0], h[1], e[0], e[1], st, tv);
}

__device__ __forceinline__ void take_if_closer(double t, int32_t prim, Trav &tv) {
    if (tv.bp < 0 || t < tv.bt || (t == tv.bt && prim > tv.bp)) { tv.bt = t; tv.bp = prim; tv.bt_above = above(t); }
}

__device__ __forceinline__ void leaf_candidate(const DPrim *prims, int32_t prim, bool moving, const RayCtx &r,
                                               double tmin, double tmax, Trav &tv) {
    const DPrim *p = prims + prim;
    double t;
    if (sphere_root(prim_centre(p, moving, r.time), p->r * p->r, r, tmin, tmax, t)) take_if_closer(t, prim, tv);
}

// the discriminant of the sphere in an LDS leaf record (centre at the ray's time: moving_sphere.cpp:3-5)
__device__ __forceinline__ void leaf_disc_lds(const char *rec, bool moving, const RayCtx &r, double &b, double &disc) {
    const double *p = reinterpret_cast<const double *>(rec);
    v3 c = mk(p[0], p[1], p[2]);
    const double radius2 = p[3];   // the LDS copy holds the square
    if (moving) {
        const uint32_t kind = *reinterpret_cast<const uint32_t *>(rec + 76);
        const double s = (kind == PK_MOVING_UNIT) ? r.time : (r.time - p[7]) / (p[8] - p[7]);
        c = vadd(c, vmul(mk(p[4], p[5], p[6]), s));
    }
    sphere_disc(c, radius2, r, b, disc);
}

// One leaf.  First the reference's own test of the leaf's box (aabb.cpp:8-57, in double, with the
// interval the reference would pass down: bvh_node.cpp:53) -- the walk above it was only
// conservative.  Then its one or two spheres (sphere.cpp:9-45, moving_sphere.cpp:13-49), then the
// next stack entry.  (Skipping behind the best hit is left to the node test, whose entry bound
// carries the margins for the rounding of the reference's roots: a computed root can undercut the
// exactly computed entry of its own sphere's box by 4e-8 of its value when the ray grazes the sphere,
// which at an origin 10^6 radii away is a sizeable part of the radius.)
//
// `box_now` = false postpones the box test of a leaf of static spheres: its spheres are tried at once
// and the caller tests the box of the winning leaf only (see world_closest_voted).
// ALL_STATIC: the scene has no moving sphere.  Every leaf is SAFE, every reference is to one sphere
// (leaves of two were split on upload, see flattener::split_pairs), and the LDS records are the
// compact ones.
template <bool SPILL, bool ALL_STATIC = false>
__device__ __forceinline__ void leaf_step_lds(const char *lleaves, const RayCtx &r, double tmin, double tmax, const LaneStack &st, Trav &tv, bool box_now) {
    const int32_t ref = tv.cur;
    const int32_t leaf = ref & LEAF_INDEX_MASK;
    if (ALL_STATIC) {
        const char *rec = lleaves + (uint32_t)leaf * SLEAF_STRIDE;
        bool reached = true;
        if (box_now) {
            double entry;
            reached = box_test_ordered(rec, r, tmin, tmax, entry);
        }
        if (reached) {
            const uint32_t k = (ref & LEAF_SECOND) ? 1u : 0u;
            const double *p = reinterpret_cast<const double *>(rec + 48 + k * SPRIM_BYTES);
            double t;
            if (sphere_root(mk(p[0], p[1], p[2]), p[3], r, tmin, tmax, t)) take_if_closer(t, 2 * leaf + (int32_t)k, tv);
        }
    } else {
        const char *rec = lleaves + (uint32_t)leaf * LLEAF_STRIDE;
        const bool safe = ref & LEAF_SAFE;
        bool reached = true;
        if (box_now || !safe) {
            double entry;
            reached = box_test_ordered(rec, r, tmin, tmax, entry);
        }
        if (reached) {
            // Both discriminants first, then the roots -- square root and division, the dear part --
            // in ONE pass for whichever sphere of the lane's leaf has a positive one: run sphere by
            // sphere, each pass would execute for the few lanes that need it.  A lane whose ray meets
            // the lines of both spheres takes a second pass.  (The order of the two candidates does not
            // matter: the nearer wins, the later primitive among equals.)
            double b0, d0, b1 = 0, d1 = -1;
            leaf_disc_lds(rec + 48, ref & LEAF_MOVING0, r, b0, d0);
            if (ref & LEAF_TWO) leaf_disc_lds(rec + 48 + LPRIM_BYTES, ref & LEAF_MOVING1, r, b1, d1);
            bool first = d0 > 0, second = d1 > 0;
            while (first || second) {
                double t;
                if (sphere_roots(first ? b0 : b1, first ? d0 : d1, r, tmin, tmax, t)) take_if_closer(t, 2 * leaf + (first ? 0 : 1), tv);
                if (first) first = false; else second = false;
            }
        }
    }
    tv.cur = trav_pop<SPILL>(tv, st);
}

template <bool SPILL, bool ALL_STATIC = false>
__device__ __forceinline__ void leaf_step(const DLeaf *leaves, const DPrim *prims, const RayCtx &r, double tmin, double tmax, const LaneStack &st, Trav &tv, bool box_now) {
    const int32_t ref = tv.cur;
    const int32_t leaf = ref & LEAF_INDEX_MASK;
    const bool safe = ALL_STATIC || (ref & LEAF_SAFE);
    bool reached = true;
    if (box_now || !safe) {
        double bx[6], entry;
#pragma unroll
        for (int k = 0; k < 6; ++k) bx[k] = leaves[leaf].box[k];
        reached = box_test(bx, r, tmin, tmax, entry);
    }
    if (reached) {
        // the first sphere of the reference: the leaf's first, or its second on its own (a split leaf)
        const int32_t first = 2 * leaf + ((ref & LEAF_SECOND) ? 1 : 0);
        if (ALL_STATIC) {
            leaf_candidate(prims, first, false, r, tmin, tmax, tv);
        } else {
            leaf_candidate(prims, first, ref & ((ref & LEAF_SECOND) ? LEAF_MOVING1 : LEAF_MOVING0), r, tmin, tmax, tv);
            if (ref & LEAF_TWO) leaf_candidate(prims, 2 * leaf + 1, ref & LEAF_MOVING1, r, tmin, tmax, tv);
        }
    }
    tv.cur = trav_pop<SPILL>(tv, st);
}

// the reference's test of the box of leaf `leaf` (as in the leaf steps), for a hit found with that test postponed
template <bool LDS_TREE>
__device__ __forceinline__ bool leaf_box_reached(const TreeView &tree, int32_t leaf, const RayCtx &r, double tmin, double tmax) {
    double entry;
    if (LDS_TREE) return box_test_ordered(tree.lleaves + (uint32_t)leaf * SLEAF_STRIDE, r, tmin, tmax, entry);   // (static scenes only)
    double bx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) bx[k] = tree.leaves[leaf].box[k];
    return box_test(bx, r, tmin, tmax, entry);
}

// Starts the walk of one world item.  The reference tests the root's box first (bvh_node.cpp:53);
// that test is implied by the exact test of every leaf's box (a leaf's box lies inside the root's),
// so the walk starts at the root's children right away.  A bare sphere in the world's list
// (has_box = 0) is tried on the spot, as object_list.cpp:18 does, and the walk stays empty.
__device__ __forceinline__ void item_begin(const DScene &sc, uint32_t item, const RayCtx &r, double tmin, double tmax, Trav &tv) {
    const DItem *it = sc.items + item;
    tv.sp = 0; tv.bp = -1; tv.bt = 0; tv.bt_above = HUGE_VALF; tv.cur = REF_NONE;
    const int32_t ref = it->ref;
    if (!it->has_box) leaf_candidate(sc.prims, 2 * (ref & LEAF_INDEX_MASK), ref & LEAF_MOVING0, r, tmin, tmax, tv);
    else tv.cur = ref;
}

// object_list::hit (object_list.cpp:12-26) over the world's items, one ray per lane:
// each item sees the interval narrowed to the closest hit so far and replaces it only
// with a strictly nearer one (sphere_root's t < tmax).
__device__ __forceinline__ bool world_closest(const DScene &sc, const RayCtx &r, double tmin, double tmax,
                                              int *stack, int stride, double &t_out, int &prim_out) {
    const LaneStack st = {stack, stride, 0, nullptr, 0};   // whole stack in LDS
    const float tmin_below = below(tmin);
    double closest = tmax;
    int found = -1;
    for (uint32_t i = 0; i < sc.n_items; ++i) {
        Trav tv;
        item_begin(sc, i, r, tmin, closest, tv);
        const float tmax_above = above(closest);
        while (tv.cur != REF_NONE) {
            if (tv.cur >= 0) node_step<false>(sc.nodes, r, tmin_below, tmax_above, st, tv);
            else leaf_step<false>(sc.leaves, sc.prims, r, tmin, closest, st, tv, true);
        }
        if (tv.bp >= 0) { closest = tv.bt; found = tv.bp; }
    }
    t_out = closest;
    prim_out = found;
    return found >= 0;
}

// The same walk for a whole wave that traces together.  Node steps outnumber leaf steps
// about five to one, so letting both bodies run in every iteration would execute the
// sphere code for a handful of lanes each time.  Instead the wave counts, per iteration,
// how many of its walking lanes stand at a leaf: the leaf body runs only when at least
// LEAF_QUORUM lanes wait for it (or nobody wants a node); otherwise the node body runs.
//
// LDS_TREE: nodes and leaves are all in LDS and the stack never spills.  Otherwise the
// first tree.n_lnodes nodes are in LDS, deeper nodes and all leaves in device memory, and
// stack entries beyond st.cap spill to device memory.
//
// `work` grows by 1 per node iteration and by 3 per leaf iteration this lane took part in: the
// lane that walked longest has counted what the wave executed, in units of roughly a node step.
//
// DEFER: for scenes without moving spheres.  The box tests of the leaves are postponed to the winning
// leaf (below), and every subtree may be skipped behind the best hit without looking at flags.
template <int LEAF_QUORUM, bool LDS_TREE, bool COUNT_WORK, bool DEFER>
__device__ __forceinline__ bool world_closest_voted(const DScene &sc, const TreeView &tree, const RayCtx &r, double tmin,
                                                    double tmax, const LaneStack &st, double &t_out, int &prim_out, uint32_t &work,
                                                    LaneStats &stats) {
    constexpr bool SPILL = !LDS_TREE;
    // node steps taken on top of the one the wave voted for (measured: 485 static spheres 300.4 ms with 0,
    // 288.6 with 1, 285.9 with 2, 285.7 with 3, 283.5 with 4, 286.7 with 5; book-2 405 / 390 / 386 / 384 /
    // 387 with 4; 10^6 spheres 50.5 / 48.2 / 47.8 / 47.95 with 3)
#ifndef RT_EXTRA_STATIC
#define RT_EXTRA_STATIC 4
#endif
#ifndef RT_EXTRA_PAIRS
#define RT_EXTRA_PAIRS 3
#endif
#ifndef RT_EXTRA_LARGE
#define RT_EXTRA_LARGE 2
#endif
    constexpr int kExtraNodeSteps = LDS_TREE ? (DEFER ? RT_EXTRA_STATIC : RT_EXTRA_PAIRS) : RT_EXTRA_LARGE;
    const PlaneAddr pa = plane_addresses(tree.lnodes, r);
    const float tmin_below = below(tmin);
    double closest = tmax;
    int found = -1;
    for (uint32_t i = 0; i < sc.n_items; ++i) {
        Trav tv;
        item_begin(sc, i, r, tmin, closest, tv);
        if (LDS_TREE && tv.cur >= 0) tv.cur *= (int32_t)LNODE_STRIDE;   // node references are record offsets there
        const float tmax_above = above(closest);
        // A static sphere lies inside its leaf's box, so a ray that hits the sphere passes the box
        // test too, up to rounding in that test.  The first pass therefore tries the spheres of such
        // leaves without the box test (exact = false): its candidates are the reference's and,
        // at most, a few that the reference's box test would have kept out.  If the winner's own box
        // test passes, it is the best of a superset that contains it: the reference's answer.  If it
        // fails -- the ray grazes sphere and box within rounding -- the lane walks the item again
        // with every box tested first (exact = true, wave-uniform).  Without DEFER the first pass
        // is exact.
        bool exact = !DEFER;
        while (tv.cur != REF_NONE) {   // (a bare sphere has been tried by item_begin)
            while (tv.cur != REF_NONE) {
                const bool at_node = tv.cur >= 0;
                const int n_node = __popcll(__ballot(at_node));
                const int n_leaf = __popcll(__ballot(!at_node));
                if (n_leaf >= LEAF_QUORUM || n_node == 0) {
                    stats.add(1, (uint32_t)n_leaf);
                    if (COUNT_WORK) work += 3;
                    if (!at_node) {
                        if (LDS_TREE) leaf_step_lds<SPILL, DEFER>(tree.lleaves, r, tmin, closest, st, tv, exact);
                        else leaf_step<SPILL, DEFER>(tree.leaves, tree.prims, r, tmin, closest, st, tv, exact);
                    }
                } else {
                    stats.add(0, (uint32_t)n_node);
                    if (COUNT_WORK) work += 1;
                    if (at_node) {
                        if (LDS_TREE) node_step_lds<SPILL, true, DEFER>(tree.lnodes, pa, r, tmin_below, tmax_above, st, tv);
                        else if (tv.cur < tree.n_lnodes) node_step_lds<SPILL, false, DEFER>(tree.lnodes, pa, r, tmin_below, tmax_above, st, tv);
                        else node_step<SPILL, DEFER>(tree.nodes, r, tmin_below, tmax_above, st, tv);
                        // A few more steps right away for the lanes that are at a node again, without asking
                        // the wave: most lanes that were at a node still are (a walk visits five nodes for every
                        // leaf), the vote and the loop around it cost as much as a third of a node step, and
                        // in a straight line the compiler overlaps one step's reads with the last one's
                        // arithmetic.  (A loop that goes on "while enough lanes are at nodes" measured 3-5 %
                        // SLOWER than no extra steps at all: it is the straight line that pays.)
#pragma unroll
                        for (int extra = 0; extra < kExtraNodeSteps; ++extra) {
                            if (tv.cur >= 0) {   // (REF_NONE is negative)
                                stats.add(0, (uint32_t)__popcll(__ballot(true)));
                                if (COUNT_WORK) work += 1;
                                if (LDS_TREE) node_step_lds<SPILL, true, DEFER>(tree.lnodes, pa, r, tmin_below, tmax_above, st, tv);
                                else if (tv.cur < tree.n_lnodes) node_step_lds<SPILL, false, DEFER>(tree.lnodes, pa, r, tmin_below, tmax_above, st, tv);
                                else node_step<SPILL, DEFER>(tree.nodes, r, tmin_below, tmax_above, st, tv);
                            }
                        }
                    }
                }
            }
            if (!DEFER || exact) break;
            bool again = false;
            if (tv.bp >= 0) again = sc.recheck_all || !leaf_box_reached<LDS_TREE>(tree, tv.bp >> 1, r, tmin, closest);
            if (!__any(again)) break;
            exact = true;
            if (again) {
                item_begin(sc, i, r, tmin, closest, tv);
                if (LDS_TREE && tv.cur >= 0) tv.cur *= (int32_t)LNODE_STRIDE;
            }
        }
        if (tv.bp >= 0) { closest = tv.bt; found = tv.bp; }
    }
    t_out = closest;
    prim_out = found;
    return found >= 0;
}

// ------------------------------------------------------------ four-wide walk ----
// The walk of world_closest_voted over DNode4 nodes (scenes without moving spheres: DEFER's rules,
// every sphere behind a box of its own, every subtree skipped behind the best hit).  One step tests
// the boxes of up to four children, goes on with the nearest one hit and pushes the others, farthest
// first.  Which candidates a walk finds does not depend on the order, and the nearest candidate wins
// with the later primitive among equals (take_if_closer): the answer is the binary walk's.
//
// Trees that live in LDS whole keep 16-bit child words, so that a child and the distance at which its
// box is entered travel in ONE 32-bit key -- the top 16 bits of the entry distance (a positive float:
// its bit pattern orders like its value; 8 bits of mantissa are plenty for an ordering that is only a
// heuristic) above the child word -- and sorting the four keys with five min/max pairs sorts the
// children; a miss is the key 0xFFFFFFFF, which is also what an empty slot's word makes of any key.
//   child word: node -> byte offset of its record / 16; leaf -> 0x8000 | (2 leaf + second sphere);
//   a walk's `cur` is the word sign-extended from 16 bits: >= 0 node, < 0 leaf, -1 = nothing left
#ifndef RT_WIDE_PACKED
#define RT_WIDE_PACKED 0
#endif
constexpr uint32_t LNODE4_STRIDE = 112;    // pl[6][4] | child words[4]: 7 x 16 bytes, 16-byte reads at this stride meet no bank twice
constexpr uint32_t kKeyMiss = 0xFFFFFFFFu;
constexpr int32_t kWideNone = -1;
constexpr uint32_t kWideLdsMaxLeaves = (1u << 14) - 1u, kWideLdsMaxNodes = (1u << 15) / (LNODE4_STRIDE / 16u);

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const floatx4 lds_cfloatx4;
typedef __attribute__((address_space(3))) const uintx4 lds_cuintx4;
__device__ __forceinline__ floatx4 lds_float4(uint32_t address) { return *(lds_cfloatx4 *)(uintptr_t)address; }
__device__ __forceinline__ uintx4 lds_uint4(uint32_t address) { return *(lds_cuintx4 *)(uintptr_t)address; }

__host__ __device__ inline uint32_t wide_word(int32_t ref) {
    if (ref == REF_NONE) return kKeyMiss;
    if (ref >= 0) return (uint32_t)ref * (LNODE4_STRIDE / 16u);
    return 0x8000u | ((uint32_t)(ref & LEAF_INDEX_MASK) << 1) | ((ref & LEAF_SECOND) ? 1u : 0u);
}
__device__ __forceinline__ int32_t wide_cur(uint32_t key) { return (int32_t)(int16_t)(uint16_t)key; }

// where a ray finds, in record 0 of the LDS copy, the four planes it meets first and last per axis
__device__ __forceinline__ PlaneAddr plane_addresses4(const char *lnodes, const RayCtx &r) {
    const uint32_t ox = r.nx8 << 1, oy = r.ny8 << 1, oz = r.nz8 << 1;   // 0 or 16
    PlaneAddr a = {lds_address(lnodes + ox), lds_address(lnodes + (ox ^ 16u)), lds_address(lnodes + 32 + oy),
                   lds_address(lnodes + 32 + (oy ^ 16u)), lds_address(lnodes + 64 + oz), lds_address(lnodes + 64 + (oz ^ 16u))};
    return a;
}

// node_test<true> of one child, as a sort key: the bits of the entry distance (clamped to the interval's
// start) where the box is hit, kKeyMiss where it is not
__device__ __forceinline__ uint32_t child_entry_bits(float nx, float fx, float ny, float fy, float nz, float fz, const RayCtx &r,
                                                     float tmin_below, float tmax_above, float limit, bool &hit) {
    const float tnx = __builtin_fmaf(nx, r.ixf, r.cnx), tfx = __builtin_fmaf(fx, r.ixf, r.cfx);
    const float tny = __builtin_fmaf(ny, r.iyf, r.cny), tfy = __builtin_fmaf(fy, r.iyf, r.cfy);
    const float tnz = __builtin_fmaf(nz, r.izf, r.cnz), tfz = __builtin_fmaf(fz, r.izf, r.cfz);
    const float e_lo = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), tnz);
    const float x_hi = __builtin_fminf(__builtin_fminf(tfx, tfy), tfz);
    float x_in, e_in;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(x_in) : "v"(x_hi), "v"(tmax_above), "v"(limit));
    asm("v_max_f32 %0, %1, %2" : "=v"(e_in) : "v"(e_lo), "v"(tmin_below));
    hit = !(x_in < e_in);
    return __float_as_uint(e_in);
}

__device__ __forceinline__ void key_order(uint32_t &a, uint32_t &b) {
    const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}

// The LDS part of a lane's stack by 32-bit LDS address (entry k at base + k * pitch), so that its reads and
// writes stay LDS instructions where the spilling path's device-memory accesses are selected beside them.
typedef __attribute__((address_space(3))) int lds_int;
struct WideStack {
    uint32_t base, pitch;
    int cap;
    int *spill;
    int spill_stride;
};
__device__ __forceinline__ WideStack wide_stack(const LaneStack &st) {
    WideStack w = {lds_address(reinterpret_cast<const char *>(st.lds)), (uint32_t)st.stride * 4u, st.cap, st.spill, st.spill_stride};
    return w;
}
__device__ __forceinline__ void wide_put(const WideStack &ws, int k, int32_t v) {
    if (k < ws.cap) *(lds_int *)(uintptr_t)(ws.base + (uint32_t)k * ws.pitch) = v;
    else ws.spill[(size_t)(k - ws.cap) * ws.spill_stride] = v;
}
__device__ __forceinline__ int32_t wide_get(const WideStack &ws, int k) {
    int32_t v;
    if (k < ws.cap) v = *(lds_int *)(uintptr_t)(ws.base + (uint32_t)k * ws.pitch);
    else v = ws.spill[(size_t)(k - ws.cap) * ws.spill_stride];
    return v;
}
// Pushes the entries that are not kKeyMiss, e3 first (sorted: a miss among them is followed by misses
// only).  While three more entries fit the LDS part of the stack every entry is written where the next
// one would go and the count moves on only for a real one; otherwise entry by entry, spilling.
// (`roomy`, the same for the whole wave: no lane is within three entries of the LDS part's end)
__device__ __forceinline__ void wide_push(const WideStack &ws, Trav &tv, uint32_t e3, uint32_t e2, uint32_t e1, bool v3, bool v2, bool v1, bool roomy) {
    if (__builtin_expect(roomy, 1)) {
        *(lds_int *)(uintptr_t)(ws.base + (uint32_t)tv.sp * ws.pitch) = (int32_t)e3; tv.sp += v3 ? 1 : 0;
        *(lds_int *)(uintptr_t)(ws.base + (uint32_t)tv.sp * ws.pitch) = (int32_t)e2; tv.sp += v2 ? 1 : 0;
        *(lds_int *)(uintptr_t)(ws.base + (uint32_t)tv.sp * ws.pitch) = (int32_t)e1; tv.sp += v1 ? 1 : 0;
    } else {
        if (v3) { wide_put(ws, tv.sp, (int32_t)e3); ++tv.sp; }
        if (v2) { wide_put(ws, tv.sp, (int32_t)e2); ++tv.sp; }
        if (v1) { wide_put(ws, tv.sp, (int32_t)e1); ++tv.sp; }
    }
}
// The next entry of a walk over 16-bit child words.  An entry carries the (truncated, hence lower) entry
// distance of its box: one whose box starts beyond the best hit found since it was pushed is dropped here,
// by the comparison that would have kept it off the stack had the hit been known then (node_test).
#ifndef RT_WIDE_CULL
#define RT_WIDE_CULL 0
#endif
__device__ __forceinline__ int32_t wide_pop16(Trav &tv, const WideStack &ws) {
    int32_t r = kWideNone;
#if RT_WIDE_CULL
    const uint32_t limit = __float_as_uint(tv.bt_above);   // (+inf while there is no hit; never negative)
    while (tv.sp > 0) {
        --tv.sp;
        const uint32_t e = (uint32_t)wide_get(ws, tv.sp);
        if ((e & 0x7FFF0000u) <= limit) { r = wide_cur(e); break; }
    }
#else
    if (tv.sp > 0) { --tv.sp; r = wide_cur((uint32_t)wide_get(ws, tv.sp)); }
#endif
    return r;
}
__device__ __forceinline__ int32_t wide_pop32(Trav &tv, const WideStack &ws) {
    int32_t r = REF_NONE;
    if (tv.sp > 0) { --tv.sp; r = wide_get(ws, tv.sp); }
    return r;
}

// one node of a tree that lives in LDS whole
__device__ __forceinline__ void node_step4_lds(const PlaneAddr &pa, uint32_t words_address, const RayCtx &r, float tmin_below,
                                               float tmax_above, const WideStack &st, Trav &tv, bool roomy) {
    const uint32_t off = (uint32_t)tv.cur << 4;
    const uintx4 w = lds_uint4(words_address + off);
    const floatx4 xn = lds_float4(pa.xn + off), xf = lds_float4(pa.xf + off), yn = lds_float4(pa.yn + off), yf = lds_float4(pa.yf + off),
                  zn = lds_float4(pa.zn + off), zf = lds_float4(pa.zf + off);
    uint32_t k[4];
#if RT_WIDE_PACKED
    // the 24 slab values by packed single-precision fma, two children at a time (the four planes of a
    // kind arrive in neighbouring registers)
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    floatx2 tnx[2], tfx[2], tny[2], tfy[2], tnz[2], tfz[2];
    const floatx2 ix = {r.ixf, r.ixf}, iy = {r.iyf, r.iyf}, iz = {r.izf, r.izf};
    const floatx2 cnx = {r.cnx, r.cnx}, cny = {r.cny, r.cny}, cnz = {r.cnz, r.cnz}, cfx = {r.cfx, r.cfx}, cfy = {r.cfy, r.cfy}, cfz = {r.cfz, r.cfz};
    tnx[0] = __builtin_elementwise_fma(xn.xy, ix, cnx); tnx[1] = __builtin_elementwise_fma(xn.zw, ix, cnx);
    tfx[0] = __builtin_elementwise_fma(xf.xy, ix, cfx); tfx[1] = __builtin_elementwise_fma(xf.zw, ix, cfx);
    tny[0] = __builtin_elementwise_fma(yn.xy, iy, cny); tny[1] = __builtin_elementwise_fma(yn.zw, iy, cny);
    tfy[0] = __builtin_elementwise_fma(yf.xy, iy, cfy); tfy[1] = __builtin_elementwise_fma(yf.zw, iy, cfy);
    tnz[0] = __builtin_elementwise_fma(zn.xy, iz, cnz); tnz[1] = __builtin_elementwise_fma(zn.zw, iz, cnz);
    tfz[0] = __builtin_elementwise_fma(zf.xy, iz, cfz); tfz[1] = __builtin_elementwise_fma(zf.zw, iz, cfz);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float e_lo = __builtin_fmaxf(__builtin_fmaxf(tnx[c >> 1][c & 1], tny[c >> 1][c & 1]), tnz[c >> 1][c & 1]);
        const float x_hi = __builtin_fminf(__builtin_fminf(tfx[c >> 1][c & 1], tfy[c >> 1][c & 1]), tfz[c >> 1][c & 1]);
        float x_in, e_in;
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(x_in) : "v"(x_hi), "v"(tmax_above), "v"(tv.bt_above));
        asm("v_max_f32 %0, %1, %2" : "=v"(e_in) : "v"(e_lo), "v"(tmin_below));
        const uint32_t key = (__float_as_uint(e_in) & 0x7FFF0000u) | w[c];   // (an empty slot's word is all ones)
        k[c] = !(x_in < e_in) ? key : kKeyMiss;
    }
#else
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bool hit;
        const uint32_t bits = child_entry_bits(xn[c], xf[c], yn[c], yf[c], zn[c], zf[c], r, tmin_below, tmax_above, tv.bt_above, hit);
        const uint32_t key = (bits & 0x7FFF0000u) | w[c];   // (an empty slot's word is all ones)
        k[c] = hit ? key : kKeyMiss;
    }
#endif
    key_order(k[0], k[1]); key_order(k[2], k[3]); key_order(k[0], k[2]); key_order(k[1], k[3]); key_order(k[1], k[2]);
    wide_push(st, tv, k[3], k[2], k[1], k[3] != kKeyMiss, k[2] != kKeyMiss, k[1] != kKeyMiss, roomy);
    int32_t next = wide_cur(k[0]);
    if (next == kWideNone) next = wide_pop16(tv, st);
    tv.cur = next;
}

// one sphere of a tree that lives in LDS whole (leaf_step_lds<.., true> on a 16-bit reference)
__device__ __forceinline__ void leaf_step4_lds(const char *lleaves, const RayCtx &r, double tmin, double tmax, const WideStack &st, Trav &tv, bool box_now) {
    const uint32_t idx = (uint32_t)tv.cur & 0x7FFFu;   // 2 leaf + (second sphere): the primitive's number
    const char *rec = lleaves + (idx >> 1) * SLEAF_STRIDE;
    bool reached = true;
    if (box_now) {
        double entry;
        reached = box_test_ordered(rec, r, tmin, tmax, entry);
    }
    if (reached) {
        const double *p = reinterpret_cast<const double *>(rec + 48 + (idx & 1u) * SPRIM_BYTES);
        double t;
        if (sphere_root(mk(p[0], p[1], p[2]), p[3], r, tmin, tmax, t)) take_if_closer(t, (int32_t)idx, tv);
    }
    tv.cur = wide_pop16(tv, st);
}

// one node read from device memory: child references are the 32-bit ones of the binary tree, sorted
// along with the bits of their entry distances
__device__ __forceinline__ void pair_order(uint32_t &ka, int32_t &ra, uint32_t &kb, int32_t &rb) {
    const bool swap = kb < ka;
    const uint32_t k = swap ? kb : ka; const int32_t q = swap ? rb : ra;
    kb = swap ? ka : kb; rb = swap ? ra : rb;
    ka = k; ra = q;
}
struct PlaneOffsets { uint32_t xn, xf, yn, yf, zn, zf; };   // byte offsets in a DNode4
__device__ __forceinline__ PlaneOffsets plane_offsets4(const RayCtx &r) {
    const uint32_t ox = r.nx8 << 1, oy = r.ny8 << 1, oz = r.nz8 << 1;
    PlaneOffsets o = {ox, ox ^ 16u, 32u + oy, 32u + (oy ^ 16u), 64u + oz, 64u + (oz ^ 16u)};
    return o;
}
__device__ __forceinline__ void node_step4(const DNode4 *nodes4, const PlaneOffsets &po, const RayCtx &r, float tmin_below, float tmax_above,
                                           const WideStack &st, Trav &tv, bool roomy) {
    const char *nd = reinterpret_cast<const char *>(nodes4) + (size_t)((uint32_t)tv.cur * (uint32_t)sizeof(DNode4));
    const uintx4 w = *reinterpret_cast<const uintx4 *>(nd + 96);
    const floatx4 xn = *reinterpret_cast<const floatx4 *>(nd + po.xn), xf = *reinterpret_cast<const floatx4 *>(nd + po.xf),
                  yn = *reinterpret_cast<const floatx4 *>(nd + po.yn), yf = *reinterpret_cast<const floatx4 *>(nd + po.yf),
                  zn = *reinterpret_cast<const floatx4 *>(nd + po.zn), zf = *reinterpret_cast<const floatx4 *>(nd + po.zf);
    uint32_t k[4];
    int32_t c4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bool hit;
        const uint32_t bits = child_entry_bits(xn[c], xf[c], yn[c], yf[c], zn[c], zf[c], r, tmin_below, tmax_above, tv.bt_above, hit);
        c4[c] = (int32_t)w[c];
        k[c] = (hit && c4[c] != REF_NONE) ? (bits & 0x7FFFFFFFu) : kKeyMiss;
    }
    pair_order(k[0], c4[0], k[1], c4[1]); pair_order(k[2], c4[2], k[3], c4[3]); pair_order(k[0], c4[0], k[2], c4[2]);
    pair_order(k[1], c4[1], k[3], c4[3]); pair_order(k[1], c4[1], k[2], c4[2]);
    wide_push(st, tv, (uint32_t)c4[3], (uint32_t)c4[2], (uint32_t)c4[1], k[3] != kKeyMiss, k[2] != kKeyMiss, k[1] != kKeyMiss, roomy);
    tv.cur = k[0] != kKeyMiss ? c4[0] : wide_pop32(tv, st);
}

// one sphere read from device memory (leaf_step<.., true> with this walk's stack)
__device__ __forceinline__ void leaf_step4(const DLeaf *leaves, const DPrim *prims, const RayCtx &r, double tmin, double tmax, const WideStack &st, Trav &tv, bool box_now) {
    const int32_t ref = tv.cur;
    const int32_t leaf = ref & LEAF_INDEX_MASK;
    bool reached = true;
    if (box_now) {
        double bx[6], entry;
#pragma unroll
        for (int k = 0; k < 6; ++k) bx[k] = leaves[leaf].box[k];
        reached = box_test(bx, r, tmin, tmax, entry);
    }
    if (reached) leaf_candidate(prims, 2 * leaf + ((ref & LEAF_SECOND) ? 1 : 0), false, r, tmin, tmax, tv);
    tv.cur = wide_pop32(tv, st);
}

// world_closest_voted<.., DEFER = true> over the four-wide tree.  LDS_TREE: nodes and spheres are all
// in LDS (16-bit child words); otherwise everything is read from device memory.  The stack spills to
// device memory beyond st.cap entries in either case: a four-wide walk can hold three entries per level.
template <int LEAF_QUORUM, bool LDS_TREE, bool COUNT_WORK>
__device__ __forceinline__ bool world_closest_wide(const DScene &sc, const TreeView &tree, const RayCtx &r, double tmin, double tmax,
                                                   const LaneStack &lane_stack, double &t_out, int &prim_out, uint32_t &work, LaneStats &stats) {
    const WideStack st = wide_stack(lane_stack);
#ifndef RT_WIDE_EXTRA
#define RT_WIDE_EXTRA 2
#endif
    constexpr int kExtraNodeSteps = LDS_TREE ? RT_WIDE_EXTRA : 1;
    constexpr int32_t kNone = LDS_TREE ? kWideNone : REF_NONE;
    const PlaneAddr pa = plane_addresses4(tree.lnodes, r);
    const uint32_t words_address = lds_address(tree.lnodes + 96);
    const PlaneOffsets po = plane_offsets4(r);
    const float tmin_below = below(tmin);
    float tmax_above = HUGE_VALF;
    double closest = tmax;
    int found = -1;
    auto begin = [&](uint32_t i, Trav &tv) {
        const DItem *it = sc.items + i;
        tv.sp = 0; tv.bp = -1; tv.bt = 0; tv.bt_above = HUGE_VALF; tv.cur = kNone;
        const int32_t ref = it->ref4;
        if (!it->has_box) leaf_candidate(sc.prims, 2 * (ref & LEAF_INDEX_MASK), false, r, tmin, closest, tv);
        else tv.cur = LDS_TREE ? wide_cur(wide_word(ref)) : ref;
    };
    auto node = [&](Trav &tv, bool roomy) {
        if (LDS_TREE) node_step4_lds(pa, words_address, r, tmin_below, tmax_above, st, tv, roomy);
        else node_step4(sc.nodes4, po, r, tmin_below, tmax_above, st, tv, roomy);
    };
    for (uint32_t i = 0; i < sc.n_items; ++i) {
        Trav tv;
        begin(i, tv);
        tmax_above = above(closest);
        bool exact = false;   // (see world_closest_voted: the first pass tries the spheres without their leaves' box tests)
        while (tv.cur != kNone) {
            while (tv.cur != kNone) {
                const bool at_node = tv.cur >= 0;
                const int n_node = __popcll(__ballot(at_node));
                const int n_leaf = __popcll(__ballot(!at_node));
                if (n_leaf >= LEAF_QUORUM || n_node == 0) {
                    stats.add(1, (uint32_t)n_leaf);
                    if (COUNT_WORK) work += 3;
                    if (!at_node) {
                        if (LDS_TREE) leaf_step4_lds(tree.lleaves, r, tmin, closest, st, tv, exact);
                        else leaf_step4(tree.leaves, tree.prims, r, tmin, closest, st, tv, exact);
                    }
                } else {
                    stats.add(0, (uint32_t)n_node);
                    if (COUNT_WORK) work += 2;
                    if (at_node) {
                        // (asked once for the steps that follow: each pushes three entries at most)
                        const bool roomy = !__any(tv.sp + 3 * (1 + kExtraNodeSteps) > st.cap);
                        node(tv, roomy);
#pragma unroll
                        for (int extra = 0; extra < kExtraNodeSteps; ++extra) {
                            if (tv.cur >= 0) {
                                stats.add(0, (uint32_t)__popcll(__ballot(true)));
                                if (COUNT_WORK) work += 2;
                                node(tv, roomy);
                            }
                        }
                    }
                }
            }
            if (exact) break;
            bool again = false;
            if (tv.bp >= 0) again = sc.recheck_all || !leaf_box_reached<LDS_TREE>(tree, tv.bp >> 1, r, tmin, closest);
            if (!__any(again)) break;
            exact = true;
            if (again) begin(i, tv);
        }
        if (tv.bp >= 0) { closest = tv.bt; found = tv.bp; }
    }
    t_out = closest;
    prim_out = found;
    return found >= 0;
}

// hit_record of the winning primitive (sphere.cpp:23-29): p = o + t d, n = (p - c) / r
__device__ __forceinline__ void hit_point(const DScene &sc, int prim, const RayCtx &r, double t, v3 &p, v3 &n) {
    const DPrim *pr = sc.prims + prim;
    v3 c = prim_centre(pr, pr->kind != PK_STATIC, r.time);
    p = vadd(r.o, vmul(r.d, t));
    n = vdiv(vsub(p, c), pr->r);
}

// -------------------------------------------------------------------- textures ----
// perlin.cpp:33-47,55-73
__device__ __forceinline__ double perlin_noise(const DScene &sc, v3 p) {
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    double accum = 0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                double lattice = sc.ranfloat[sc.perm[(i + a) & 255] ^ sc.perm[256 + ((j + b) & 255)] ^
                                             sc.perm[512 + ((k + c) & 255)]];
                accum += (a * u + (1 - a) * (1 - u)) * (b * v + (1 - b) * (1 - v)) * (c * w + (1 - c) * (1 - w)) * lattice;
            }
    return accum;
}

// The chequer (texture.cpp:10-13) only asks whether sin(ax) sin(ay) sin(az) < 0.  No factor of
// a finite non-zero argument is zero and the product cannot underflow, so that is "an odd
// number of negative factors", and sin(a) < 0 exactly when floor(a / pi) is odd.  With
// q = a * RN(1/pi), |q - a/pi| <= |a/pi| 2^-51.9 < 2^-32 for |a| < 2^21; when q is farther
// than 2^-29 from an integer, floor(q) = floor(a/pi) and the sign is known from 4 operations
// instead of a full sin().  `sure` is cleared when that cannot be concluded (a near a multiple
// of pi, huge, zero or not finite).
__device__ __forceinline__ bool sine_is_negative(double a, bool &sure) {
    const double q = a * 0.31830988618379067154;
    const double n = __builtin_rint(q);
    const double d = q - n;
    if (!(__builtin_fabs(a) < 2097152.0 && __builtin_fabs(d) > 0x1p-29)) sure = false;
    return (((int)n & 1) != 0) != (d < 0);
}

// texture.hpp:19, texture.cpp:10-18.  Chequer children index earlier table entries, so
// the walk ends.
__device__ __forceinline__ v3 texture_value(const DScene &sc, uint32_t tex, v3 p) {
    const DTex *t = sc.texs + tex;
    if (t->kind == RT_TEX_CHEQUER) {
        const double ax = 10 * p.x, ay = 10 * p.y, az = 10 * p.z;
        bool sure = true;
        bool odd = sine_is_negative(ax, sure) != sine_is_negative(ay, sure);
        odd = odd != sine_is_negative(az, sure);
        if (!sure) odd = sin(ax) * sin(ay) * sin(az) < 0;
        do {
            t = sc.texs + (odd ? t->odd : t->even);
        } while (t->kind == RT_TEX_CHEQUER);
    }
    if (t->kind == RT_TEX_PERLIN) {
        double grey = perlin_noise(sc, p);
        return mk(grey, grey, grey);
    }
    return ld3(t->c);
}

// ------------------------------------------------------------------- materials ----
// material.cpp:35-40
template <class R> __device__ __forceinline__ v3 random_in_unit_sphere(R &g) {
    v3 c;
    do {
        c.x = g.symmetric();   // dist{-1, 1} (material.cpp:42)
        c.y = g.symmetric();
        c.z = g.symmetric();
    } while (!(c.x * c.x + c.y * c.y + c.z * c.z < 1) && !g.exhausted());
    return c;
}

__device__ __forceinline__ v3 random_in_unit_sphere(StreamRng &g) {
    v3 c;
    g.in_unit_sphere(c.x, c.y, c.z);
    return c;
}

// material.cpp:10-12
__device__ __forceinline__ v3 reflect(v3 v, v3 n) { return vsub(v, vmul(n, 2 * vdot(v, n))); }

// x^5 rounded once: the product is carried as an unevaluated sum of two doubles (error below 2^-100
// of the value) and rounded at the end, which is the correctly rounded result unless the exact value
// lies within 2^-100 of a rounding boundary -- what a libm pow(x, 5) of 0.5x ulp accuracy returns
// in all but such cases.  Used where the reference's draws are replayed (material.cpp:29).
__device__ __forceinline__ double pow5_rounded(double x) {
    const double h2 = x * x, l2 = __builtin_fma(x, x, -h2);
    const double p4 = h2 * h2, e4 = __builtin_fma(h2, h2, -p4) + 2 * (h2 * l2);
    const double h4 = p4 + e4, l4 = e4 - (h4 - p4);
    const double p5 = h4 * x, e5 = __builtin_fma(h4, x, -p5) + l4 * x;
    return p5 + e5;
}

// material.cpp:47-98.  Returns whether the ray scatters; the draws happen either way.
template <class R>
__device__ __forceinline__ bool scatter(const DScene &sc, uint32_t mat, const Ray &r, v3 p, v3 n, R &g, v3 &att, Ray &out) {
    const DMat *m = sc.mats + mat;
    const uint32_t kind = m->kind;
    out.o = p;
    out.time = r.time;
    // Lambertian and metal both start by drawing a point in the unit sphere
    // (material.cpp:48,61) and draw nothing else: one rejection loop serves the lanes of
    // both kinds instead of one loop per kind.
    v3 rnd = mk(0, 0, 0);
    if (kind != RT_MAT_DIELECTRIC) rnd = random_in_unit_sphere(g);
    if (kind == RT_MAT_LAMBERTIAN) {
        v3 target = vadd(vadd(p, n), rnd);
        att = texture_value(sc, m->tex, p);
        out.d = vsub(target, p);
        return true;
    }
    if (kind == RT_MAT_METAL) {
        v3 reflected = reflect(vunit(r.d), n);
        out.d = vadd(reflected, vmul(rnd, m->fuzz));
        att = ld3(m->albedo);
        return vdot(out.d, n) > 0;
    }
    // dielectric (material.cpp:77-98, refract :14-24, schlick :26-30)
    const double ri = m->ri;
    const double dn = vdot(r.d, n);
    const bool inside = dn > 0;
    const v3 outward = inside ? vneg(n) : n;
    const v3 reflected = reflect(r.d, n);
    const double ni_over_nt = inside ? ri : m->inv_ri;
    const double len = vlen(r.d);
    const double cosine = inside ? ri * dn / len : -dn / len;
    const v3 uv = vdiv(r.d, len);
    const double dt = vdot(uv, outward);
    const double disc = 1 - ni_over_nt * ni_over_nt * (1 - dt * dt);
    double reflect_prob = 1.0;
    v3 refracted = reflected;
    if (disc > 0) {
        refracted = vsub(vmul(vsub(uv, vmul(outward, dt)), ni_over_nt), vmul(outward, sqrt(disc)));
        const double r1 = m->r0sq;
        // schlick (material.cpp:26-30).  On a path's own stream pow(x, 5) is taken by repeated
        // multiplication: at most 2 ulp from a correctly rounded pow, and the value is only compared
        // with a uniform of that stream.  Where the caller's draws are replayed the comparison must
        // fall as the reference's does, and the power is rounded once, as libm rounds it.
        const double x = 1 - cosine, x2 = x * x;
        reflect_prob = r1 + (1 - r1) * (R::kMaySkipDeadDraws ? x2 * x2 * x : pow5_rounded(x));
    }
    att = mk(1, 1, 1);
    out.d = g.canonical() < reflect_prob ? reflected : refracted;   // dist{0, 1}: u * 1 + 0 = u
    return true;
}

// ---------------------------------------------------------------------- camera ----
// camera.cpp:16-21,38-48
template <class R> __device__ __forceinline__ Ray camera_ray(const DCamera &c, double s, double t, R &g) {
    // The reference samples the lens disk even when the lens radius is zero (camera.cpp:39),
    // which only matters for the position of later draws in its one shared generator.  The
    // sample is multiplied by the radius, so with a zero radius a per-path stream skips
    // the rejection loop; the tape variant, which replays the reference's draws, keeps it.
    double px = 0, py = 0;
    if (!(R::kMaySkipDeadDraws && c.lens_radius == 0)) {
        do {
            px = g.symmetric();   // camera.cpp:19 draws from dist{-1, 1}
            py = g.symmetric();
        } while (!(px * px + py * py + 0.0 * 0.0 < 1) && !g.exhausted());
    }
    v3 rd = vmul(mk(px, py, 0), c.lens_radius);
    v3 offset = vadd(vmul(ld3(c.u), rd.x), vmul(ld3(c.v), rd.y));
    double time = c.t0 + g.symmetric() * (c.t1 - c.t0);
    Ray r;
    r.o = vadd(ld3(c.origin), offset);
    r.d = vsub(vsub(vadd(vadd(ld3(c.llc), vmul(ld3(c.horizontal), s)), vmul(ld3(c.vertical), t)), ld3(c.origin)), offset);
    r.time = time;
    return r;
}

// main.cpp:45-48
__device__ __forceinline__ v3 sky(v3 d) {
    const double unit_y = d.y / vlen(d);   // only the y component of the unit vector is used
    double t = 0.5 * (unit_y + 1.0);
    return mk((1.0 - t) * 1.0 + t * 0.5, (1.0 - t) * 1.0 + t * 0.7, (1.0 - t) * 1.0 + t * 1.0);
}

}  // namespace rtd
