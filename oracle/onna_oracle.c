/*
 * onna_oracle.c -- CPU restatement (plain C, double precision, recursive as the original) of the
 * reference's Common Lisp tracer, /root/reference/onna/: hit-test over spheres, quads, sequences,
 * bvh nodes, translate, rotate-y and constant media; textures; materials; PDFs; the camera and the
 * importance-sampling ray-colour.  Citations are file:line under onna/.
 *
 * THIS IS TEST INFRASTRUCTURE: the checker the HIP path (libonna_hip.so) is compared against.  It is
 * never shipped, never linked into the product libraries and never used as a fallback.
 *
 * Parity status: PARITY UNPINNED.  No Common Lisp implementation is in this image (no sbcl), the
 * reference holds no test, fixture or rendered image for this tracer, and its random numbers come from
 * the `random-state` library.  What pins this file is the text of the Lisp source alone: every
 * expression below keeps the Lisp expression's operand order (n-ary + and * associate to the left,
 * vec3+ sums from zero, unit-vec3 multiplies by the reciprocal of the length) and its numeric
 * contagion (single-float literals such as 0.001 and 0.0001 widen to double exactly).  Stated
 * deviations, each where the Lisp text cannot be followed to the letter:
 *   - random numbers: every camera sample draws from its own counter-based stream (the GPU library's
 *     Philox -> xoshiro128++), `(random 1d0)` = one 53-bit uniform, `(random n)` = floor(uniform * n);
 *   - perlin-noise masks its lattice indices with +perlin-point-count+ = 256 (texture.lisp:89-93), which
 *     indexes entry 256 of 256-entry tables whenever bit 8 is set; restated with 255, the mask the same
 *     author's C++ tracer uses (c++/perlin.cpp:61-68);
 *   - IEEE arithmetic without traps: a zero direction component, log 0 or 0/0 give infinities and NaNs
 *     here where SBCL would signal; min / max keep the old bound on an unordered comparison;
 *   - (expt x 5) in reflectance (material.lisp:91) is x*x * x*x * x, not libm's pow;
 *   - an image texture's u = 1 or v = 0 indexes one past the last column / row (texture.lisp:42-44):
 *     clamped to the last;
 *   - a scattering direction whose mixture density is not positive ends the path with its emission
 *     only (camera.lisp:129 would divide by zero);
 *   - a world without lights samples the material's own PDF alone (camera.lisp:113-116 would call
 *     (random 0)).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "onna_oracle.h"

/* ------------------------------------------------------------------ vectors ---- */
typedef struct { double x, y, z; } V;

static V v3(double x, double y, double z) { V r = {x, y, z}; return r; }
static V ld(const double *p) { return v3(p[0], p[1], p[2]); }
/* vec.lisp:27-33: vec3+ sums its arguments into a zero vector */
static V add2(V a, V b) { return v3((0.0 + a.x) + b.x, (0.0 + a.y) + b.y, (0.0 + a.z) + b.z); }
static V add3(V a, V b, V c) { return v3(((0.0 + a.x) + b.x) + c.x, ((0.0 + a.y) + b.y) + c.y, ((0.0 + a.z) + b.z) + c.z); }
/* vec.lisp:35-43 */
static V neg(V a) { return v3(-a.x, -a.y, -a.z); }
static V sub(V a, V b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
/* vec.lisp:45-48 */
static V scaled(V a, double d) { return v3(a.x * d, a.y * d, a.z * d); }
/* vec.lisp:50-60 */
static double dot(V u, V v) { return (u.x * v.x + u.y * v.y) + u.z * v.z; }
static V cross(V u, V v) { return v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
/* vec.lisp:18-25 */
static double lensq(V v) { return (v.x * v.x + v.y * v.y) + v.z * v.z; }
static double len(V v) { return sqrt(lensq(v)); }
/* vec.lisp:62-64 */
static V unit(V v) { return scaled(v, 1.0 / len(v)); }
/* vec.lisp:94-105: point3+ adds the sum of the vectors to the point; point3- */
static V padd1(V p, V a) { V s = v3(0.0 + a.x, 0.0 + a.y, 0.0 + a.z); return v3(p.x + s.x, p.y + s.y, p.z + s.z); }
static V padd2(V p, V a, V b) { V s = add2(a, b); return v3(p.x + s.x, p.y + s.y, p.z + s.z); }
/* vec.lisp:74-76 */
static V reflect(V v, V n) { return sub(v, scaled(n, 2 * dot(v, n))); }
/* vec.lisp:78-83 */
static double lmin(double a, double b) { return b < a ? b : a; }
static double lmax(double a, double b) { return b > a ? b : a; }
static V refract(V uv, V n, double eta) {
    const double cos_theta = lmin(dot(neg(uv), n), 1.0);
    const V perp = scaled(add2(uv, scaled(n, cos_theta)), eta);
    const V par = scaled(n, -sqrt(fabs(1.0 - lensq(perp))));
    return add2(perp, par);
}

typedef struct { V o, d; double time; } R;
/* ray.lisp:9-11 */
static V at(const R *r, double t) { return padd1(r->o, scaled(r->d, t)); }

#define PI 3.14159265358979323846
static const double kShadowAcne = (double)0.001f;   /* camera.lisp:92: a single-float literal */
static const double kMediumStep = (double)0.0001f;  /* scene.lisp:432 */

/* ------------------------------------------------------------------- draws ---- */
double onc_uniform(odraw *g) {
    if (g->tape) {
        double u = 0.0;
        if (g->pos < g->len) u = g->tape[g->pos]; else g->over = 1;
        g->pos++;
        return u;
    }
    return orc_stream_canonical(&g->st);
}
static double uniform2(odraw *g) { return 2.0 * onc_uniform(g); }   /* (random 2d0) */
/* vec.lisp:66-72 */
static V random_unit(odraw *g) {
    for (;;) {
        V p;
        p.x = -1.0 + uniform2(g);
        p.y = -1.0 + uniform2(g);
        p.z = -1.0 + uniform2(g);
        const double l2 = lensq(p);
        if ((1e-160 < l2 && l2 <= 1.0) || g->over) return scaled(p, 1.0 / sqrt(l2));
    }
}

/* ---------------------------------------------------------------- hit-test ---- */
typedef struct {
    double t;
    V p, outward;   /* in the space of the object that was hit, then carried outward by the instances */
    V normal;
    int front, object, material;
    double u, v;
} H;

/* scene.lisp:5-10: front-face and the normal facing the ray */
static void finish(H *h, const R *r) {
    h->front = dot(r->d, h->outward) < 0;
    h->normal = h->front ? h->outward : neg(h->outward);
}

/* aabb.lisp:38-59 */
static int hit_aabb(const double *b, const R *r, double lo, double hi) {
    const double o[3] = {r->o.x, r->o.y, r->o.z}, d[3] = {r->d.x, r->d.y, r->d.z};
    for (int a = 0; a < 3; ++a) {
        const double inv = 1.0 / d[a];
        const double t0 = (b[2 * a] - o[a]) * inv, t1 = (b[2 * a + 1] - o[a]) * inv;
        if (t0 < t1) { lo = lmax(lo, t0); hi = lmin(hi, t1); }
        else { lo = lmax(lo, t1); hi = lmin(hi, t0); }
        if (hi <= lo) return 0;
    }
    return 1;
}

static int hit_object(const on_scene_desc *s, int obj, const R *r, double lo, double hi, odraw *g, H *out, onc_counters *cnt);

/* scene.lisp:150-179 */
static int hit_sphere(const on_object_desc *o, int id, const R *r, double lo, double hi, H *out) {
    const R centre = {ld(o->v), ld(o->v + 3), 0};
    const double radius = o->v[6];
    const V cur = at(&centre, r->time);
    const V oc = sub(cur, r->o);
    const double a = lensq(r->d), h = dot(r->d, oc), c = lensq(oc) - radius * radius;
    const double disc = h * h - a * c;
    if (disc < 0) return 0;
    const double sd = sqrt(disc);
    double root = (h - sd) / a;
    if (!(lo < root && root < hi)) {
        root = (h + sd) / a;
        if (!(lo < root && root < hi)) return 0;
    }
    out->t = root;
    out->p = at(r, root);
    out->outward = scaled(sub(out->p, cur), 1.0 / radius);
    const double theta = acos(-out->outward.y), phi = atan2(-out->outward.z, out->outward.x) + PI;
    out->u = phi / (2 * PI);
    out->v = theta / PI;
    out->object = id;
    out->material = o->material;
    finish(out, r);
    return 1;
}

/* scene.lisp:240-262 */
static int hit_quad(const on_object_desc *o, int id, const R *r, double lo, double hi, H *out) {
    const V q = ld(o->v), u = ld(o->v + 3), v = ld(o->v + 6), w = ld(o->v + 9), normal = ld(o->v + 12);
    const double denom = dot(normal, r->d);
    if (fabs(denom) < 1e-8) return 0;
    const double t = (o->v[15] - dot(normal, r->o)) / denom;
    if (!(lo <= t && t <= hi)) return 0;
    const V p = at(r, t);
    const V planar = sub(p, q);
    const double alpha = dot(w, cross(planar, v)), beta = dot(w, cross(u, planar));
    if (!(0.0 <= alpha && alpha <= 1.0 && 0.0 <= beta && beta <= 1.0)) return 0;
    out->t = t; out->p = p; out->outward = normal; out->u = alpha; out->v = beta;
    out->object = id; out->material = o->material;
    finish(out, r);
    return 1;
}

/* scene.lisp:423-452 */
static int hit_medium(const on_scene_desc *s, const on_object_desc *o, int id, const R *r, double lo, double hi, odraw *g, H *out, onc_counters *cnt) {
    H r1, r2;
    if (!hit_object(s, o->a, r, -HUGE_VAL, HUGE_VAL, g, &r1, cnt)) return 0;
    if (!hit_object(s, o->a, r, r1.t + kMediumStep, HUGE_VAL, g, &r2, cnt)) return 0;
    double t1 = lmax(r1.t, lo), t2 = lmin(r2.t, hi);
    if (t1 >= t2) return 0;
    t1 = lmax(t1, 0.0);
    const double ray_length = len(r->d);
    const double inside = (t2 - t1) * ray_length;
    const double hit_distance = o->v[0] * log(onc_uniform(g));
    const double hit_time = t1 + hit_distance / ray_length;
    if (hit_distance > inside) return 0;
    out->t = hit_time;
    out->p = at(r, hit_time);
    out->outward = v3(1, 0, 0);
    out->u = 0; out->v = 0;
    out->object = id; out->material = o->material;
    finish(out, r);
    return 1;
}

static int hit_object(const on_scene_desc *s, int obj, const R *r, double lo, double hi, odraw *g, H *out, onc_counters *cnt) {
    const on_object_desc *o = s->objects + obj;
    switch (o->kind) {
    case ON_OBJ_SPHERE: if (cnt) cnt->sphere_tests++; return hit_sphere(o, obj, r, lo, hi, out);
    case ON_OBJ_QUAD: if (cnt) cnt->quad_tests++; return hit_quad(o, obj, r, lo, hi, out);
    case ON_OBJ_LIST: {   /* scene.lisp:92-99 */
        int found = 0;
        for (int i = 0; i < o->b; ++i) {
            H h;
            if (hit_object(s, s->children[o->a + i], r, lo, hi, g, &h, cnt)) { *out = h; hi = h.t; found = 1; }
        }
        return found;
    }
    case ON_OBJ_BVH: {   /* scene.lisp:77-85 */
        if (cnt) cnt->box_tests++;
        if (!hit_aabb(o->box, r, lo, hi)) return 0;
        H hl, hr;
        const int left = hit_object(s, o->a, r, lo, hi, g, &hl, cnt);
        const int right = hit_object(s, o->b, r, lo, left ? hl.t : hi, g, &hr, cnt);
        if (right) { *out = hr; return 1; }
        if (left) { *out = hl; return 1; }
        return 0;
    }
    case ON_OBJ_TRANSLATE: {   /* scene.lisp:328-333 */
        const V off = ld(o->v);
        R moved = *r;
        moved.o = padd1(r->o, neg(off));
        if (!hit_object(s, o->a, &moved, lo, hi, g, out, cnt)) return 0;
        out->p = padd1(out->p, off);
        return 1;
    }
    case ON_OBJ_ROTATE_Y: {   /* scene.lisp:379-405 */
        const double sn = o->v[0], cs = o->v[1];
        R turned = *r;
        turned.o = v3(cs * r->o.x - sn * r->o.z, r->o.y, sn * r->o.x + cs * r->o.z);
        turned.d = v3(cs * r->d.x - sn * r->d.z, r->d.y, sn * r->d.x + cs * r->d.z);
        if (!hit_object(s, o->a, &turned, lo, hi, g, out, cnt)) return 0;
        const V p = out->p, n = out->normal;
        out->p = v3(cs * p.x + sn * p.z, p.y, (-sn) * p.x + cs * p.z);
        out->normal = v3(cs * n.x + sn * n.z, n.y, (-sn) * n.x + cs * n.z);
        return 1;
    }
    case ON_OBJ_MEDIUM: return hit_medium(s, o, obj, r, lo, hi, g, out, cnt);
    }
    return 0;
}

/* ---------------------------------------------------------------- textures ---- */
/* texture.lisp:71-97 (lattice indices masked with 255: see the header) */
static double perlin_noise(const on_perlin_desc *p, V pt) {
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
                /* alexandria:lerp v a b = a + v (b - a) */
                const double li = (1 - uu) + di * (uu - (1 - uu)), lj = (1 - vv) + dj * (vv - (1 - vv)), lk = (1 - ww) + dk * (ww - (1 - ww));
                sum_k = sum_k + ((li * lj) * lk) * dot(c, weight);
            }
            sum_j = sum_j + sum_k;
        }
        total = total + sum_j;
    }
    return total;
}
/* texture.lisp:99-104 */
static double perlin_turbulence(const on_perlin_desc *p, V pt, int depth) {
    double weight = 1, sum = 0;
    for (int i = 0; i < depth; ++i) {
        sum = sum + weight * perlin_noise(p, pt);
        weight = weight * 0.5;
        pt = scaled(pt, 2.0);
    }
    return fabs(sum);
}

static void texture_value(const on_scene_desc *s, int tex, double u, double v, V p, double out[3]) {
    const on_texture_desc *t = s->textures + tex;
    while (t->kind == ON_TEX_CHEQUER) {   /* texture.lisp:22-30 */
        const double x = floor(t->scale * p.x), y = floor(t->scale * p.y), z = floor(t->scale * p.z);
        const double sum = x + y + z;
        const int even = fmod(sum, 2.0) == 0.0;
        t = s->textures + (even ? t->even : t->odd);
    }
    if (t->kind == ON_TEX_IMAGE) {   /* texture.lisp:39-44, image.lisp:12-13 */
        const on_image_desc *im = s->images + t->table;
        v = 1.0 - v;
        double fi = floor(u * im->width), fj = floor(v * im->height);
        if (!(fi >= 0)) fi = 0;
        if (!(fj >= 0)) fj = 0;
        if (fi > im->width - 1) fi = im->width - 1;
        if (fj > im->height - 1) fj = im->height - 1;
        const uint8_t *px = s->pixels + im->offset + 3 * ((size_t)fj * im->width + (size_t)fi);
        for (int c = 0; c < 3; ++c) out[c] = (double)(px[c] * px[c]) / 65536.0;
        return;
    }
    if (t->kind == ON_TEX_PERLIN) {   /* texture.lisp:106-111 */
        const double n = sin(t->scale * p.z + perlin_turbulence(s->perlins + t->table, p, 7) * 10);
        const double c = 0.5 * (1 + n);
        out[0] = out[1] = out[2] = c;
        return;
    }
    out[0] = t->colour[0]; out[1] = t->colour[1]; out[2] = t->colour[2];
}

/* -------------------------------------------------------------------- PDFs ---- */
/* onb.lisp:5-16 */
typedef struct { V u, v, w; } ONB;
static ONB make_onb(V n) {
    ONB o;
    o.w = unit(n);
    const V a = fabs(o.w.x) > 0.9 ? v3(0, 1, 0) : v3(1, 0, 0);
    o.v = unit(cross(o.w, a));
    o.u = unit(cross(o.w, o.v));
    return o;
}
/* onb.lisp:18-21 */
static V onb_transform(const ONB *o, V v) { return add3(scaled(o->u, v.x), scaled(o->v, v.y), scaled(o->w, v.z)); }

/* scene.lisp:106-111, :184-196, :267-283 */
static double object_pdf_value(const on_scene_desc *s, int obj, V origin, V direction) {
    const on_object_desc *o = s->objects + obj;
    const R r = {origin, direction, 0};
    H h;
    if (o->kind == ON_OBJ_LIST) {
        const double weight = 1.0 / (double)o->b;
        double sum = 0;
        for (int i = 0; i < o->b; ++i) sum = sum + weight * object_pdf_value(s, s->children[o->a + i], origin, direction);
        return sum;
    }
    if (o->kind == ON_OBJ_SPHERE) {
        if (!hit_sphere(o, obj, &r, kShadowAcne, HUGE_VAL, &h)) return 0;
        const V dir = sub(ld(o->v), origin);   /* the centre at time 0 */
        const double d2 = lensq(dir), radius = o->v[6];
        const double cos_max = sqrt(1 - (radius * radius) / d2);
        const double solid = (2 * PI) * (1 - cos_max);
        return 1.0 / solid;
    }
    if (o->kind == ON_OBJ_QUAD) {
        if (!hit_quad(o, obj, &r, kShadowAcne, HUGE_VAL, &h)) return 0;
        const double d2 = (h.t * h.t) * lensq(direction);
        const double cosine = fabs(dot(direction, h.normal) / len(direction));
        return d2 / (cosine * o->v[16]);
    }
    return 0;
}

/* scene.lisp:113-115, :198-211, :285-290 */
static V object_random_direction(const on_scene_desc *s, int obj, V origin, odraw *g) {
    const on_object_desc *o = s->objects + obj;
    if (o->kind == ON_OBJ_LIST) {
        int i = (int)floor(onc_uniform(g) * o->b);
        if (i >= o->b) i = o->b - 1;
        return object_random_direction(s, s->children[o->a + i], origin, g);
    }
    if (o->kind == ON_OBJ_SPHERE) {
        const V dir = sub(ld(o->v), origin);
        const double d2 = lensq(dir), radius = o->v[6];
        const ONB uvw = make_onb(dir);
        const double r1 = onc_uniform(g), r2 = onc_uniform(g);
        const double z = 1 + r2 * (sqrt(1 - (radius * radius) / d2) - 1);
        const double phi = (2 * PI) * r1;
        const double x = cos(phi) * sqrt(1 - z * z), y = sin(phi) * sqrt(1 - z * z);
        return onb_transform(&uvw, v3(x, y, z));
    }
    /* quad */
    const V a = scaled(ld(o->v + 3), onc_uniform(g));
    const V b = scaled(ld(o->v + 6), onc_uniform(g));
    return sub(padd2(ld(o->v), a, b), origin);
}

/* pdf.lisp:30-44 */
static double cosine_pdf_value(const ONB *uvw, V direction) { return lmax(0.0, dot(unit(direction), uvw->w) / PI); }
static V cosine_pdf_generate(const ONB *uvw, odraw *g) {
    const double r1 = onc_uniform(g), r2 = onc_uniform(g);
    const double phi = (2.0 * PI) * r1;
    const double x = cos(phi) * sqrt(r2), y = sin(phi) * sqrt(r2), z = sqrt(1.0 - r2);
    return onb_transform(uvw, v3(x, y, z));
}

/* --------------------------------------------------------------- materials ---- */
/* material.lisp:87-91 */
static double reflectance(double cosine, double ri) {
    double r0 = (1.0 - ri) / (1.0 + ri);
    r0 = r0 * r0;
    const double x = 1 - cosine, x2 = x * x;
    return r0 + (1.0 - r0) * ((x2 * x2) * x);
}

/* camera.lisp:94-134 */
static void ray_colour(const on_scene_desc *s, const R *r, int depth, const double bg[3], odraw *g, double out[3], onc_counters *cnt) {
    out[0] = out[1] = out[2] = 0;
    if (!(depth > 0)) return;
    if (cnt) cnt->rays++;
    H h;
    if (!hit_object(s, s->world, r, kShadowAcne, HUGE_VAL, g, &h, cnt)) { out[0] = bg[0]; out[1] = bg[1]; out[2] = bg[2]; return; }
    const on_material_desc *m = s->materials + h.material;
    double emit[3] = {0, 0, 0};
    if (m->kind == ON_MAT_DIFFUSE_LIGHT) {   /* material.lisp:113-124 */
        if (h.front) texture_value(s, m->texture, h.u, h.v, h.p, emit);
        out[0] = emit[0]; out[1] = emit[1]; out[2] = emit[2];
        return;
    }
    if (m->kind == ON_MAT_METAL || m->kind == ON_MAT_DIELECTRIC) {
        R next = {h.p, v3(0, 0, 0), r->time};
        double att[3] = {1, 1, 1};
        if (m->kind == ON_MAT_METAL) {   /* material.lisp:72-78 */
            next.d = add2(unit(reflect(r->d, h.normal)), scaled(random_unit(g), m->fuzz));
            att[0] = m->albedo[0]; att[1] = m->albedo[1]; att[2] = m->albedo[2];
        } else {   /* material.lisp:93-106 */
            const double ri = h.front ? 1.0 / m->refraction_index : m->refraction_index;
            const V ud = unit(r->d);
            const double cos_t = lmin(dot(neg(ud), h.normal), 1.0);
            const double sin_t = sqrt(1.0 - cos_t * cos_t);
            if (ri * sin_t > 1.0 || reflectance(cos_t, ri) > onc_uniform(g)) next.d = reflect(ud, h.normal);
            else next.d = refract(ud, h.normal, ri);
        }
        double in[3];
        ray_colour(s, &next, depth - 1, bg, g, in, cnt);
        for (int c = 0; c < 3; ++c) out[c] = in[c] * att[c];   /* camera.lisp:110-111 */
        return;
    }
    /* lambertian (material.lisp:52-64) and isotropic (:131-141): a density to sample */
    double att[3];
    texture_value(s, m->texture, h.u, h.v, h.p, att);
    const int cosine = m->kind == ON_MAT_LAMBERTIAN;
    ONB uvw;
    memset(&uvw, 0, sizeof uvw);
    if (cosine) uvw = make_onb(h.normal);
    /* pdf.lisp:64-71 over the lights (pdf.lisp:52-56) and the material's density */
    V direction;
    const int use_lights = s->lights >= 0;
    if (use_lights && onc_uniform(g) < 0.5) direction = object_random_direction(s, s->lights, h.p, g);
    else direction = cosine ? cosine_pdf_generate(&uvw, g) : random_unit(g);
    const double own = cosine ? cosine_pdf_value(&uvw, direction) : 1.0 / (4 * PI);
    const double pdf_value = use_lights ? 0.5 * object_pdf_value(s, s->lights, h.p, direction) + 0.5 * own : own;
    const R scattered = {h.p, direction, r->time};
    double spdf;
    if (cosine) {
        const double c = dot(h.normal, unit(scattered.d));
        spdf = c < 0 ? 0.0 : c / PI;
    } else {
        spdf = 1.0 / (4.0 * PI);
    }
    if (!(pdf_value > 0)) return;   /* (emission of these materials is zero) */
    double in[3];
    ray_colour(s, &scattered, depth - 1, bg, g, in, cnt);
    const double w = spdf / pdf_value;
    /* camera.lisp:112-130: emission + (next * attenuation) * (c c c) */
    for (int c = 0; c < 3; ++c) out[c] = (0.0 + emit[c]) + (in[c] * att[c]) * w;
}

/* camera.lisp:136-163 */
static R get_ray(const on_camera *c, int i, int j, int si, int sj, odraw *g) {
    const double ox = (si + onc_uniform(g)) * c->recip_sqrt_spp - 0.5;
    const double oy = (sj + onc_uniform(g)) * c->recip_sqrt_spp - 0.5;
    const V sample = padd2(ld(c->pixel00), scaled(ld(c->pixel_delta_u), i + ox), scaled(ld(c->pixel_delta_v), j + oy));
    double us, vs;
    do {
        us = uniform2(g) - 1;
        vs = uniform2(g) - 1;
    } while (!(us * us + vs * vs < 1) && !g->over);
    const V origin = padd2(ld(c->centre), scaled(ld(c->defocus_u), us), scaled(ld(c->defocus_v), vs));
    R r;
    r.o = origin;
    r.d = sub(sample, origin);
    r.time = onc_uniform(g);
    return r;
}

/* ------------------------------------------------------- flat entry points ---- */
static void store_hit(on_hit *o, int found, const H *h) {
    memset(o, 0, sizeof *o);
    o->object = -1; o->material = -1;
    if (!found) return;
    o->hit = 1; o->object = h->object; o->material = h->material; o->front_face = h->front;
    o->t = h->t; o->u = h->u; o->v = h->v;
    o->p[0] = h->p.x; o->p[1] = h->p.y; o->p[2] = h->p.z;
    o->normal[0] = h->normal.x; o->normal[1] = h->normal.y; o->normal[2] = h->normal.z;
}

void onc_hit_batch(const on_scene_desc *s, int32_t object, const double *rays, size_t n, double t_min, double t_max,
                   const double *uniforms, uint32_t k, on_hit *out) {
    for (size_t i = 0; i < n; ++i) {
        const double *q = rays + 7 * i;
        const R r = {v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), q[6]};
        odraw g;
        memset(&g, 0, sizeof g);
        static const double none = 0;
        g.tape = uniforms ? uniforms + (size_t)k * i : &none; g.len = uniforms ? (int)k : 0;
        H h;
        const int found = hit_object(s, object, &r, t_min, t_max, &g, &h, NULL);
        store_hit(out + i, found, &h);
    }
}

void onc_texture_batch(const on_scene_desc *s, int32_t texture, const double *uvp, size_t n, double *rgb) {
    for (size_t i = 0; i < n; ++i) texture_value(s, texture, uvp[5 * i], uvp[5 * i + 1], ld(uvp + 5 * i + 2), rgb + 3 * i);
}

void onc_pdf_value_batch(const on_scene_desc *s, int32_t object, const double *od, size_t n, double *out) {
    for (size_t i = 0; i < n; ++i) out[i] = object_pdf_value(s, object, ld(od + 6 * i), ld(od + 6 * i + 3));
}

void onc_random_direction_batch(const on_scene_desc *s, int32_t object, const double *origins, const double *tape, uint32_t k, size_t n, double *out) {
    for (size_t i = 0; i < n; ++i) {
        odraw g;
        memset(&g, 0, sizeof g);
        g.tape = tape + (size_t)k * i; g.len = (int)k;
        const V d = object_random_direction(s, object, ld(origins + 3 * i), &g);
        out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
    }
}

/* render (camera.lisp:165-181): per-pixel SUMS of the samples [sample_begin, sample_end), top row first */
void onc_render(const on_scene_desc *s, const on_camera *c, const on_render_params *p, double *out_sum, onc_counters *cnt) {
    if (cnt) memset(cnt, 0, sizeof *cnt);
    for (uint32_t j = 0; j < c->image_height; ++j)
        for (uint32_t i = 0; i < c->image_width; ++i) {
            double sum[3] = {0, 0, 0};
            const uint32_t pixel = j * c->image_width + i;
            for (uint32_t smp = p->sample_begin; smp < p->sample_end; ++smp) {
                odraw g;
                memset(&g, 0, sizeof g);
                orc_stream_init(&g.st, p->seed, pixel, smp);
                const R r = get_ray(c, (int)i, (int)j, (int)(smp % c->sqrt_spp), (int)(smp / c->sqrt_spp), &g);
                double col[3];
                ray_colour(s, &r, (int)c->max_depth, c->background, &g, col, cnt);
                if (cnt) cnt->paths++;
                for (int k = 0; k < 3; ++k) sum[k] += col[k];
            }
            for (int k = 0; k < 3; ++k) out_sum[3 * (size_t)pixel + k] = sum[k];
        }
}

/* one camera ray with its draws (tests) */
void onc_camera_ray(const on_camera *c, int i, int j, int si, int sj, uint64_t seed, uint32_t pixel, uint32_t sample, double out[7]) {
    odraw g;
    memset(&g, 0, sizeof g);
    orc_stream_init(&g.st, seed, pixel, sample);
    const R r = get_ray(c, i, j, si, sj, &g);
    out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z; out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z; out[6] = r.time;
}

/* colour.lisp:12-18,31-37; image.lisp:30-40 */
void onc_resolve(const double *sums, size_t n_values, uint32_t spp, uint8_t *rgb8) {
    for (size_t i = 0; i < n_values; ++i) {
        const double mean = sums[i] / (double)spp;
        double x = sqrt(mean);
        x = x < 0 ? 0 : (x > (double)0.999f ? (double)0.999f : x);
        if (!(x == x)) x = 0;
        rgb8[i] = (uint8_t)floor(256 * x);
    }
}
