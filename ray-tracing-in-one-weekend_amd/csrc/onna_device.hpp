// onna_device.hpp -- gfx950 device code for the features of the reference's Common Lisp tracer
// (hgiddens/ray-tracing-in-one-weekend, onna/): hit-test over the object graph, textures, PDFs,
// materials, the camera and ray-colour.  Citations are file:line under onna/.
//
// Everything is IEEE double precision; the translation unit is built with -ffp-contract=off and every
// expression keeps the operand order of the Lisp expression it stands for (vec3+ sums its arguments
// into a zero vector, n-ary + and * associate to the left, unit-vec3 multiplies by the reciprocal of
// the length), so that a hit's t, point and normal are the values the Lisp arithmetic produces.
//
// The Lisp hit-test is a recursion over generic functions.  Here it is ONE loop over an explicit
// stack per lane (walk): a sequence leaves a continuation on the stack, a bvh node its right child, an
// instance (translate, rotate-y) a marker that restores the ray when its subtree is done.  Every
// container of the Lisp code hands the interval (min, closest hit so far) to what it visits next, in
// depth-first order, so the loop keeps one `closest` and visits in that same order.  The instances'
// effect on the hit record (scene.lisp:332, 403-404) is postponed to the winner: the loop records the
// chain of instances a candidate was found under, and the winner's point, normal and texture
// coordinates are computed once, under that chain, and carried back out through it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "onna_hip.h"
#include "rt_device.hpp"   // the per-sample random streams (rtd::StreamRng)

namespace ond {

struct V { double x, y, z; };
__device__ __forceinline__ V v3(double x, double y, double z) { V r = {x, y, z}; return r; }
__device__ __forceinline__ V ld(const double *p) { return v3(p[0], p[1], p[2]); }
// vec.lisp:27-33
__device__ __forceinline__ V add2(V a, V b) { return v3((0.0 + a.x) + b.x, (0.0 + a.y) + b.y, (0.0 + a.z) + b.z); }
__device__ __forceinline__ V add3(V a, V b, V c) { return v3(((0.0 + a.x) + b.x) + c.x, ((0.0 + a.y) + b.y) + c.y, ((0.0 + a.z) + b.z) + c.z); }
// vec.lisp:35-60
__device__ __forceinline__ V neg(V a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V sub(V a, V b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V scaled(V a, double d) { return v3(a.x * d, a.y * d, a.z * d); }
__device__ __forceinline__ double dot(V u, V v) { return (u.x * v.x + u.y * v.y) + u.z * v.z; }
__device__ __forceinline__ V cross(V u, V v) { return v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
__device__ __forceinline__ double lensq(V v) { return (v.x * v.x + v.y * v.y) + v.z * v.z; }
__device__ __forceinline__ double len(V v) { return sqrt(lensq(v)); }
__device__ __forceinline__ V unit(V v) { return scaled(v, 1.0 / len(v)); }   // vec.lisp:62-64
// vec.lisp:94-99
__device__ __forceinline__ V padd1(V p, V a) { return v3(p.x + (0.0 + a.x), p.y + (0.0 + a.y), p.z + (0.0 + a.z)); }
__device__ __forceinline__ V padd2(V p, V a, V b) { const V s = add2(a, b); return v3(p.x + s.x, p.y + s.y, p.z + s.z); }
// vec.lisp:74-83
__device__ __forceinline__ V reflect(V v, V n) { return sub(v, scaled(n, 2 * dot(v, n))); }
__device__ __forceinline__ double lmin(double a, double b) { return b < a ? b : a; }
__device__ __forceinline__ double lmax(double a, double b) { return b > a ? b : a; }
__device__ __forceinline__ V refract(V uv, V n, double eta) {
    const double cos_theta = lmin(dot(neg(uv), n), 1.0);
    const V perp = scaled(add2(uv, scaled(n, cos_theta)), eta);
    const V par = scaled(n, -sqrt(fabs(1.0 - lensq(perp))));
    return add2(perp, par);
}

struct Ray { V o, d; double time; };
__device__ __forceinline__ V at(const Ray &r, double t) { return padd1(r.o, scaled(r.d, t)); }   // ray.lisp:9-11

constexpr double kPi = 3.14159265358979323846;
constexpr double kShadowAcne = (double)0.001f;    // camera.lisp:92, a single-float literal
constexpr double kMediumStep = (double)0.0001f;   // scene.lisp:432

// An object as the kernels read it: the description's fields with what a bvh visit needs -- kind, links, box -- in the
// first 64 bytes, and a 256-byte stride, so that a node visit touches one cache line and a sphere two.
struct __attribute__((aligned(64))) DObj {
    uint32_t kind;
    int32_t material, a, b;
    double box[6];
    double v[18];
    uint32_t flags;      // kObjUsesUV
    uint32_t pad0;
    double pad[5];
};
constexpr uint32_t kObjUsesUV = 1u;   // a sphere whose material's texture reads (u, v): an image somewhere in it
static_assert(sizeof(DObj) == 256, "DObj is four 64-byte lines");

struct DScene {
    const DObj *objects;
    const int32_t *children;      // list members, each list closed by -1 (on_scene_create adds the terminators)
    const on_material_desc *materials;
    const on_texture_desc *textures;
    const on_image_desc *images;
    const uint8_t *pixels;
    const on_perlin_desc *perlins;
    int32_t world, lights;
};

// ---- draws ------------------------------------------------------------------------------------------
struct StreamDraw {
    rtd::StreamRng g;
    __device__ __forceinline__ double uniform() { return g.canonical(); }   // (random 1d0)
    __device__ __forceinline__ bool over() const { return false; }
};
struct TapeDraw {
    const double *tape; int n, pos; bool ran_out;
    __device__ __forceinline__ double uniform() {
        double u = 0.0;
        if (pos < n) u = tape[pos]; else ran_out = true;
        ++pos;
        return u;
    }
    __device__ __forceinline__ bool over() const { return ran_out; }
};
// vec.lisp:66-72
template <class G> __device__ __forceinline__ V random_unit(G &g) {
    for (;;) {
        V p;
        p.x = -1.0 + 2.0 * g.uniform();
        p.y = -1.0 + 2.0 * g.uniform();
        p.z = -1.0 + 2.0 * g.uniform();
        const double l2 = lensq(p);
        if ((1e-160 < l2 && l2 <= 1.0) || g.over()) return scaled(p, 1.0 / sqrt(l2));
    }
}

// ---- instances --------------------------------------------------------------------------------------
constexpr int kMaxChain = 4;       // instances around one object (on_scene_create checks)
constexpr int kStackCap = 32;      // stack entries per lane (checked likewise)
// (four named fields rather than an array: indexed by selects, they stay in registers)
struct Chain {
    int32_t op0, op1, op2, op3, n;
    __device__ __forceinline__ int32_t get(int k) const { return k == 0 ? op0 : (k == 1 ? op1 : (k == 2 ? op2 : op3)); }
    __device__ __forceinline__ void append(int32_t v) {
        if (n == 0) op0 = v; else if (n == 1) op1 = v; else if (n == 2) op2 = v; else op3 = v;
        ++n;
    }
};
static_assert(kMaxChain == 4, "Chain holds four instances");

// scene.lisp:329-330, 381-401
__device__ __forceinline__ Ray enter_instance(const DObj *o, const Ray &r) {
    Ray q = r;
    if (o->kind == ON_OBJ_TRANSLATE) {
        q.o = padd1(r.o, neg(ld(o->v)));
    } else {
        const double sn = o->v[0], cs = o->v[1];
        q.o = v3(cs * r.o.x - sn * r.o.z, r.o.y, sn * r.o.x + cs * r.o.z);
        q.d = v3(cs * r.d.x - sn * r.d.z, r.d.y, sn * r.d.x + cs * r.d.z);
    }
    return q;
}
__device__ __forceinline__ Ray under_chain(const DScene &sc, const Ray &world, const Chain &c) {
    Ray r = world;
    for (int k = 0; k < c.n; ++k) r = enter_instance(sc.objects + c.get(k), r);
    return r;
}

// ---- primitive tests: only what decides the hit (its time) -------------------------------------------
// aabb.lisp:38-59.  (Forming 1 / direction once per ray instead of once per box, as the Lisp does, measured no
// faster on the scenes with a bvh and 20 % slower on the ones without: six more registers live through the walk.)
__device__ __forceinline__ bool hit_aabb(const double *b, const Ray &r, double lo, double hi) {
    const double o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double inv = 1.0 / d[a];
        const double t0 = (b[2 * a] - o[a]) * inv, t1 = (b[2 * a + 1] - o[a]) * inv;
        if (t0 < t1) { lo = lmax(lo, t0); hi = lmin(hi, t1); }
        else { lo = lmax(lo, t1); hi = lmin(hi, t0); }
        if (hi <= lo) return false;
    }
    return true;
}
// scene.lisp:165-179
__device__ __forceinline__ bool sphere_time(const DObj *o, const Ray &r, double lo, double hi, double &t_out) {
    const Ray centre = {ld(o->v), ld(o->v + 3), 0};
    const double radius = o->v[6];
    const V cur = at(centre, r.time);
    const V oc = sub(cur, r.o);
    const double a = lensq(r.d), h = dot(r.d, oc), c = lensq(oc) - radius * radius;
    const double disc = h * h - a * c;
    if (disc < 0) return false;
    const double sd = sqrt(disc);
    double root = (h - sd) / a;
    if (!(lo < root && root < hi)) {
        root = (h + sd) / a;
        if (!(lo < root && root < hi)) return false;
    }
    t_out = root;
    return true;
}
// scene.lisp:240-262
__device__ __forceinline__ bool quad_time(const DObj *o, const Ray &r, double lo, double hi, double &t_out, double &alpha, double &beta) {
    const V normal = ld(o->v + 12);
    const double denom = dot(normal, r.d);
    if (fabs(denom) < 1e-8) return false;
    const double t = (o->v[15] - dot(normal, r.o)) / denom;
    if (!(lo <= t && t <= hi)) return false;
    const V planar = sub(at(r, t), ld(o->v));
    const V w = ld(o->v + 9);
    alpha = dot(w, cross(planar, ld(o->v + 6)));
    beta = dot(w, cross(ld(o->v + 3), planar));
    if (!(0.0 <= alpha && alpha <= 1.0 && 0.0 <= beta && beta <= 1.0)) return false;
    t_out = t;
    return true;
}

// ---- the walk ---------------------------------------------------------------------------------------
struct Found { double t; int32_t object; Chain chain; };

struct LaneStack { int32_t *base; int stride; };
__device__ __forceinline__ void push(const LaneStack &st, int &sp, int32_t v) { st.base[sp * st.stride] = v; ++sp; }
__device__ __forceinline__ int32_t pop(const LaneStack &st, int &sp) { --sp; return st.base[sp * st.stride]; }

constexpr int32_t kListTag = (int32_t)0x80000000u, kLeaveInstance = (int32_t)0xC0000000u, kTagMask = (int32_t)0xC0000000u;

// hit-test of `root` for the ray `world_ray` seen under `chain` (scene.lisp:22 and its methods), with the
// interval (lo, hi).  MEDIA: constant media may lie under root (their boundaries are walked with MEDIA
// off: a medium inside a boundary is refused at upload).  Uses stack entries from sp0 up.
template <bool MEDIA, class G>
__device__ __forceinline__ bool walk(const DScene &sc, int32_t root, const Ray &world_ray, Chain chain, double lo, double hi, const LaneStack &st,
                                     int sp0, G &g, Found &out) {
    bool found = false;
    Ray cur = under_chain(sc, world_ray, chain);
    int sp = sp0;
    push(st, sp, root);
    while (sp > sp0) {
        int32_t code = pop(st, sp);
        if (code == kLeaveInstance) {
            --chain.n;
            cur = under_chain(sc, world_ray, chain);
            continue;
        }
        if ((code & kTagMask) == kListTag) {   // scene.lisp:92-99: the next member of a sequence
            const int32_t pos = code & ~kTagMask;
            const int32_t member = sc.children[pos];
            if (member < 0) continue;
            push(st, sp, kListTag | (pos + 1));
            code = member;
        }
        const DObj *o = sc.objects + code;
        const uint32_t kind = o->kind;
        if (kind == ON_OBJ_SPHERE) {
            double t;
            if (sphere_time(o, cur, lo, hi, t)) { found = true; hi = t; out.t = t; out.object = code; out.chain = chain; }
        } else if (kind == ON_OBJ_QUAD) {
            double t, alpha, beta;
            if (quad_time(o, cur, lo, hi, t, alpha, beta)) { found = true; hi = t; out.t = t; out.object = code; out.chain = chain; }
        } else if (kind == ON_OBJ_LIST) {
            push(st, sp, kListTag | o->a);
        } else if (kind == ON_OBJ_BVH) {   // scene.lisp:77-85: the left child, then the right one with what the left one left of the interval
            if (hit_aabb(o->box, cur, lo, hi)) { push(st, sp, o->b); push(st, sp, o->a); }
        } else if (kind == ON_OBJ_TRANSLATE || kind == ON_OBJ_ROTATE_Y) {
            push(st, sp, kLeaveInstance);
            chain.append(code);
            cur = enter_instance(o, cur);
            push(st, sp, o->a);
        } else if (MEDIA && kind == ON_OBJ_MEDIUM) {   // scene.lisp:423-452
            Found r1, r2;
            if (!walk<false>(sc, o->a, world_ray, chain, -HUGE_VAL, HUGE_VAL, st, sp, g, r1)) continue;
            if (!walk<false>(sc, o->a, world_ray, chain, r1.t + kMediumStep, HUGE_VAL, st, sp, g, r2)) continue;
            double t1 = lmax(r1.t, lo);
            const double t2 = lmin(r2.t, hi);
            if (t1 >= t2) continue;
            t1 = lmax(t1, 0.0);
            const double ray_length = len(cur.d);
            const double inside = (t2 - t1) * ray_length;
            const double hit_distance = o->v[0] * log(g.uniform());
            const double hit_time = t1 + hit_distance / ray_length;
            if (hit_distance > inside) continue;
            found = true; hi = hit_time; out.t = hit_time; out.object = code; out.chain = chain;
        }
    }
    return found;
}

// the hit-record of the winner (scene.lisp:5-20, 151-164, 256-262, 446-452), carried out through its instances
struct Hit { double t; V p, normal; double u, v; int32_t object, material; bool front; };
// `every_uv` = false: a sphere's (u, v) -- an arc cosine and an arc tangent -- are left at zero where no texture reads them
__device__ __forceinline__ Hit hit_record(const DScene &sc, const Ray &world_ray, const Found &f, bool every_uv = true) {
    const Ray r = under_chain(sc, world_ray, f.chain);
    const DObj *o = sc.objects + f.object;
    Hit h;
    h.t = f.t; h.object = f.object; h.material = o->material;
    h.p = at(r, f.t);
    V outward;
    if (o->kind == ON_OBJ_SPHERE) {
        const Ray centre = {ld(o->v), ld(o->v + 3), 0};
        outward = scaled(sub(h.p, at(centre, r.time)), 1.0 / o->v[6]);
        h.u = 0; h.v = 0;
        if (every_uv || (o->flags & kObjUsesUV)) {   // scene.lisp:118-127
            const double theta = acos(-outward.y), phi = atan2(-outward.z, outward.x) + kPi;
            h.u = phi / (2 * kPi);
            h.v = theta / kPi;
        }
    } else if (o->kind == ON_OBJ_QUAD) {
        outward = ld(o->v + 12);
        const V planar = sub(h.p, ld(o->v));
        const V w = ld(o->v + 9);
        h.u = dot(w, cross(planar, ld(o->v + 6)));
        h.v = dot(w, cross(ld(o->v + 3), planar));
    } else {   // a medium: the normal is arbitrary (scene.lisp:448-450)
        outward = v3(1, 0, 0);
        h.u = 0; h.v = 0;
    }
    h.front = dot(r.d, outward) < 0;
    h.normal = h.front ? outward : neg(outward);
    for (int k = f.chain.n - 1; k >= 0; --k) {   // scene.lisp:332, 391-404, innermost first
        const DObj *op = sc.objects + f.chain.get(k);
        if (op->kind == ON_OBJ_TRANSLATE) {
            h.p = padd1(h.p, ld(op->v));
        } else {
            const double sn = op->v[0], cs = op->v[1];
            const V p = h.p, n = h.normal;
            h.p = v3(cs * p.x + sn * p.z, p.y, (-sn) * p.x + cs * p.z);
            h.normal = v3(cs * n.x + sn * n.z, n.y, (-sn) * n.x + cs * n.z);
        }
    }
    return h;
}

// ---- textures ---------------------------------------------------------------------------------------
// texture.lisp:71-97.  The Lisp masks its lattice indices with 256, the table size, which picks entry
// 256 of 256-entry tables whenever bit 8 is set; 255, the mask of the same author's C++ tracer
// (c++/perlin.cpp:61-68), is used instead.
__device__ __forceinline__ double perlin_noise(const on_perlin_desc *p, V pt) {
    const double fi = floor(pt.x), fj = floor(pt.y), fk = floor(pt.z);
    const double u = pt.x - fi, v = pt.y - fj, w = pt.z - fk;
    const long long i = (long long)fi, j = (long long)fj, k = (long long)fk;
    const double uu = (u * u) * (3 - 2 * u), vv = (v * v) * (3 - 2 * v), ww = (w * w) * (3 - 2 * w);
    double total = 0;
    for (int di = 0; di < 2; ++di) {
        double sum_j = 0;
        for (int dj = 0; dj < 2; ++dj) {
            double sum_k = 0;
            for (int dk = 0; dk < 2; ++dk) {
                const int idx = p->perm[0][(i + di) & 255] ^ p->perm[1][(j + dj) & 255] ^ p->perm[2][(k + dk) & 255];
                const V c = ld(p->vec[idx]);
                const V weight = v3(u - di, v - dj, w - dk);
                const double li = (1 - uu) + di * (uu - (1 - uu)), lj = (1 - vv) + dj * (vv - (1 - vv)), lk = (1 - ww) + dk * (ww - (1 - ww));
                sum_k = sum_k + ((li * lj) * lk) * dot(c, weight);
            }
            sum_j = sum_j + sum_k;
        }
        total = total + sum_j;
    }
    return total;
}
// texture.lisp:99-104
__device__ __forceinline__ double perlin_turbulence(const on_perlin_desc *p, V pt, int depth) {
    double weight = 1, sum = 0;
    for (int i = 0; i < depth; ++i) {
        sum = sum + weight * perlin_noise(p, pt);
        weight = weight * 0.5;
        pt = scaled(pt, 2.0);
    }
    return fabs(sum);
}
// texture.lisp:8-10, 22-30, 39-44, 106-111
__device__ __forceinline__ V texture_value(const DScene &sc, int32_t tex, double u, double v, V p) {
    const on_texture_desc *t = sc.textures + tex;
    while (t->kind == ON_TEX_CHEQUER) {
        const double x = floor(t->scale * p.x), y = floor(t->scale * p.y), z = floor(t->scale * p.z);
        const double sum = x + y + z;
        const bool even = fmod(sum, 2.0) == 0.0;
        t = sc.textures + (even ? t->even : t->odd);
    }
    if (t->kind == ON_TEX_IMAGE) {
        const on_image_desc *im = sc.images + t->table;
        v = 1.0 - v;
        double fi = floor(u * im->width), fj = floor(v * im->height);
        // (u = 1 or v = 0 index one past the end in the Lisp; the last column / row is taken)
        if (!(fi >= 0)) fi = 0;
        if (!(fj >= 0)) fj = 0;
        if (fi > im->width - 1) fi = im->width - 1;
        if (fj > im->height - 1) fj = im->height - 1;
        const uint8_t *px = sc.pixels + im->offset + 3 * ((size_t)fj * im->width + (size_t)fi);
        // image.lisp:12-13: (x / 256)^2, exact
        return v3((double)(px[0] * px[0]) / 65536.0, (double)(px[1] * px[1]) / 65536.0, (double)(px[2] * px[2]) / 65536.0);
    }
    if (t->kind == ON_TEX_PERLIN) {
        const double n = sin(t->scale * p.z + perlin_turbulence(sc.perlins + t->table, p, 7) * 10);
        const double c = 0.5 * (1 + n);
        return v3(c, c, c);
    }
    return ld(t->colour);
}

// ---- PDFs -------------------------------------------------------------------------------------------
// onb.lisp:5-21
struct ONB { V u, v, w; };
__device__ __forceinline__ ONB make_onb(V n) {
    ONB o;
    o.w = unit(n);
    const V a = fabs(o.w.x) > 0.9 ? v3(0, 1, 0) : v3(1, 0, 0);
    o.v = unit(cross(o.w, a));
    o.u = unit(cross(o.w, o.v));
    return o;
}
__device__ __forceinline__ V onb_transform(const ONB &o, V v) { return add3(scaled(o.u, v.x), scaled(o.v, v.y), scaled(o.w, v.z)); }

// object-pdf-value of one sphere or quad (scene.lisp:184-196, 267-283)
__device__ __forceinline__ double one_pdf_value(const DObj *o, V origin, V direction) {
    const Ray r = {origin, direction, 0};
    if (o->kind == ON_OBJ_SPHERE) {
        double t;
        if (!sphere_time(o, r, kShadowAcne, HUGE_VAL, t)) return 0;
        const V dir = sub(ld(o->v), origin);   // towards the centre at time 0
        const double d2 = lensq(dir), radius = o->v[6];
        const double cos_max = sqrt(1 - (radius * radius) / d2);
        const double solid = (2 * kPi) * (1 - cos_max);
        return 1.0 / solid;
    }
    if (o->kind == ON_OBJ_QUAD) {
        double t, alpha, beta;
        if (!quad_time(o, r, kShadowAcne, HUGE_VAL, t, alpha, beta)) return 0;
        const V outward = ld(o->v + 12);
        const V normal = dot(direction, outward) < 0 ? outward : neg(outward);
        const double d2 = (t * t) * lensq(direction);
        const double cosine = fabs(dot(direction, normal) / len(direction));
        return d2 / (cosine * o->v[16]);
    }
    return 0;
}
// ... of the lights: one of them, or a sequence (scene.lisp:106-111)
__device__ __forceinline__ double object_pdf_value(const DScene &sc, int32_t obj, V origin, V direction) {
    const DObj *o = sc.objects + obj;
    if (o->kind != ON_OBJ_LIST) return one_pdf_value(o, origin, direction);
    const double weight = 1.0 / (double)o->b;
    double sum = 0;
    for (int i = 0; i < o->b; ++i) sum = sum + weight * one_pdf_value(sc.objects + sc.children[o->a + i], origin, direction);
    return sum;
}
// random-direction (scene.lisp:113-115, 198-211, 285-290)
template <class G> __device__ __forceinline__ V object_random_direction(const DScene &sc, int32_t obj, V origin, G &g) {
    const DObj *o = sc.objects + obj;
    if (o->kind == ON_OBJ_LIST) {
        int i = (int)floor(g.uniform() * o->b);
        if (i >= o->b) i = o->b - 1;
        o = sc.objects + sc.children[o->a + i];
    }
    if (o->kind == ON_OBJ_SPHERE) {
        const V dir = sub(ld(o->v), origin);
        const double d2 = lensq(dir), radius = o->v[6];
        const ONB uvw = make_onb(dir);
        const double r1 = g.uniform(), r2 = g.uniform();
        const double z = 1 + r2 * (sqrt(1 - (radius * radius) / d2) - 1);
        const double phi = (2 * kPi) * r1;
        const double x = cos(phi) * sqrt(1 - z * z), y = sin(phi) * sqrt(1 - z * z);
        return onb_transform(uvw, v3(x, y, z));
    }
    const V a = scaled(ld(o->v + 3), g.uniform());
    const V b = scaled(ld(o->v + 6), g.uniform());
    return sub(padd2(ld(o->v), a, b), origin);
}
// pdf.lisp:30-44
__device__ __forceinline__ double cosine_pdf_value(const ONB &uvw, V direction) { return lmax(0.0, dot(unit(direction), uvw.w) / kPi); }
template <class G> __device__ __forceinline__ V cosine_pdf_generate(const ONB &uvw, G &g) {
    const double r1 = g.uniform(), r2 = g.uniform();
    const double phi = (2.0 * kPi) * r1;
    const double x = cos(phi) * sqrt(r2), y = sin(phi) * sqrt(r2), z = sqrt(1.0 - r2);
    return onb_transform(uvw, v3(x, y, z));
}

// material.lisp:87-91; (expt x 5) by multiplication
__device__ __forceinline__ double reflectance(double cosine, double ri) {
    double r0 = (1.0 - ri) / (1.0 + ri);
    r0 = r0 * r0;
    const double x = 1 - cosine, x2 = x * x;
    return r0 + (1.0 - r0) * ((x2 * x2) * x);
}

// ---- camera and ray-colour --------------------------------------------------------------------------
// camera.lisp:136-163
template <class G> __device__ __forceinline__ Ray get_ray(const on_camera &c, int i, int j, int si, int sj, G &g) {
    const double ox = (si + g.uniform()) * c.recip_sqrt_spp - 0.5;
    const double oy = (sj + g.uniform()) * c.recip_sqrt_spp - 0.5;
    const V sample = padd2(ld(c.pixel00), scaled(ld(c.pixel_delta_u), i + ox), scaled(ld(c.pixel_delta_v), j + oy));
    double us, vs;
    do {
        us = 2.0 * g.uniform() - 1;
        vs = 2.0 * g.uniform() - 1;
    } while (!(us * us + vs * vs < 1) && !g.over());
    const V origin = padd2(ld(c.centre), scaled(ld(c.defocus_u), us), scaled(ld(c.defocus_v), vs));
    Ray r;
    r.o = origin;
    r.d = sub(sample, origin);
    r.time = g.uniform();
    return r;
}

// ray-colour (camera.lisp:94-134) as a loop over ray segments: the recursion's products are taken front to
// back (`thr`), so a sample's colour can differ from the recursive form's in the last bits.  A path is a
// small state that one call of path_segment carries one segment further; the render kernel keeps one per
// lane and never makes a lane wait for the other lanes' paths to end.
struct Path {
    Ray r;
    V thr, out;
    uint32_t depth;   // segments this path may still trace
};
__device__ __forceinline__ void path_begin(Path &p, const Ray &r, uint32_t max_depth) {
    p.r = r; p.thr = v3(1, 1, 1); p.out = v3(0, 0, 0); p.depth = max_depth;
}
// traces one segment; false when the path has ended (p.out is then the sample's colour)
template <class G>
__device__ __forceinline__ bool path_segment(const DScene &sc, Path &p, const double *background, const LaneStack &st, G &g) {
    if (p.depth == 0) return false;   // the recursion limit: no more light (camera.lisp:131-134)
    --p.depth;
    const Chain none = {0, 0, 0, 0, 0};
    const Ray r = p.r;
    V &thr = p.thr, &out = p.out;
    // A world that is a sequence (scene.lisp:92-99) is taken member by member by all lanes together, each member with
    // what the members before it left of the interval -- the order of the Lisp's loop -- so that the lanes enter
    // every member's walk at the same time and in the same kind of object, instead of drifting apart down the list.
    Found f;
    bool found = false;
    {
        const DObj *w = sc.objects + sc.world;
        const bool sequence = w->kind == ON_OBJ_LIST;
        double hi = HUGE_VAL;
        for (int32_t pos = w->a;; ++pos) {
            const int32_t member = sequence ? __builtin_amdgcn_readfirstlane(sc.children[pos]) : sc.world;
            if (member < 0) break;
            if (walk<true>(sc, member, r, none, kShadowAcne, hi, st, 0, g, f)) { found = true; hi = f.t; }   // (a walk writes `f` only for a hit before `hi`)
            if (!sequence) break;
        }
    }
    if (!found) {
        out = v3(out.x + thr.x * background[0], out.y + thr.y * background[1], out.z + thr.z * background[2]);
        p.depth = 0;
        return true;
    }
    const Hit h = hit_record(sc, r, f, false);
    const on_material_desc *m = sc.materials + h.material;
    const uint32_t kind = m->kind;
    if (kind == ON_MAT_DIFFUSE_LIGHT) {   // material.lisp:113-124: emits on its front face, scatters nothing
        if (h.front) {
            const V e = texture_value(sc, m->texture, h.u, h.v, h.p);
            out = v3(out.x + thr.x * e.x, out.y + thr.y * e.y, out.z + thr.z * e.z);
        }
        p.depth = 0;
        return true;
    }
    Ray next = {h.p, v3(0, 0, 0), r.time};
    if (kind == ON_MAT_METAL) {   // material.lisp:72-78
        next.d = add2(unit(reflect(r.d, h.normal)), scaled(random_unit(g), m->fuzz));
        thr = v3(thr.x * m->albedo[0], thr.y * m->albedo[1], thr.z * m->albedo[2]);
    } else if (kind == ON_MAT_DIELECTRIC) {   // material.lisp:93-106
        const double ri = h.front ? 1.0 / m->refraction_index : m->refraction_index;
        const V ud = unit(r.d);
        const double cos_t = lmin(dot(neg(ud), h.normal), 1.0);
        const double sin_t = sqrt(1.0 - cos_t * cos_t);
        if (ri * sin_t > 1.0 || reflectance(cos_t, ri) > g.uniform()) next.d = reflect(ud, h.normal);
        else next.d = refract(ud, h.normal, ri);
    } else {
        // lambertian (material.lisp:52-64) and isotropic (:131-141): a density to sample, mixed with the lights' (pdf.lisp:48-71)
        const V att = texture_value(sc, m->texture, h.u, h.v, h.p);
        const bool cosine = kind == ON_MAT_LAMBERTIAN;
        ONB uvw = {v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)};
        if (cosine) uvw = make_onb(h.normal);
        const bool use_lights = sc.lights >= 0;
        V direction;
        if (use_lights && g.uniform() < 0.5) direction = object_random_direction(sc, sc.lights, h.p, g);
        else direction = cosine ? cosine_pdf_generate(uvw, g) : random_unit(g);
        const double own = cosine ? cosine_pdf_value(uvw, direction) : 1.0 / (4 * kPi);
        const double pdf_value = use_lights ? 0.5 * object_pdf_value(sc, sc.lights, h.p, direction) + 0.5 * own : own;
        double spdf;
        if (cosine) {
            const double c = dot(h.normal, unit(direction));
            spdf = c < 0 ? 0.0 : c / kPi;
        } else {
            spdf = 1.0 / (4.0 * kPi);
        }
        if (!(pdf_value > 0)) { p.depth = 0; return true; }   // (camera.lisp:129 would divide by zero)
        const double w = spdf / pdf_value;
        thr = v3((thr.x * att.x) * w, (thr.y * att.y) * w, (thr.z * att.z) * w);
        next.d = direction;
    }
    p.r = next;
    return true;
}

}  // namespace ond
