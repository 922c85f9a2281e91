/*
 * rt_oracle.c -- CPU restatement (plain C, double precision) of the reference C++
 * tracer's hot path.  TEST INFRASTRUCTURE ONLY: see rt_oracle.h.
 *
 * Build with -ffp-contract=off: the reference is built for x86-64 without FMA, so
 * every product and sum below rounds separately, in the order the reference's
 * expressions evaluate (left to right, component-wise vec3 operators of vec3.cpp).
 *
 * Citations are file:line under /root/reference/c++/.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vec3 ---- */
/* vec3.cpp:40-74: operators work component by component; scalar ops multiply or
 * DIVIDE each component (no reciprocal). */
static inline ovec3 v3(double x, double y, double z) { ovec3 r = {x, y, z}; return r; }
static inline ovec3 vadd(ovec3 a, ovec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ovec3 vsub(ovec3 a, ovec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ovec3 vmul(ovec3 a, double d) { return v3(a.x * d, a.y * d, a.z * d); }
static inline ovec3 vdiv(ovec3 a, double d) { return v3(a.x / d, a.y / d, a.z / d); }
static inline ovec3 vneg(ovec3 a) { return v3(-a.x, -a.y, -a.z); }
/* vec3.cpp:24-26 */
static inline double vdot(ovec3 a, ovec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* vec3.cpp:16-22 */
static inline ovec3 vcross(ovec3 a, ovec3 b) {
    return v3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
/* vec3.cpp:28-38 */
static inline double vlen(ovec3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline ovec3 vunit(ovec3 a) { return vdiv(a, vlen(a)); }

static ocounters g_cnt;
void orc_counters_reset(void) { memset(&g_cnt, 0, sizeof g_cnt); }
void orc_counters_get(ocounters *out) { *out = g_cnt; }

/* ------------------------------------------------------------- mt19937 ------ */
/* std::mt19937 (32-bit Mersenne twister, standard parameters). */
void orc_mt_seed(omt19937 *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

uint32_t orc_mt_next(omt19937 *g) {
    if (g->idx >= 624) {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (g->mt[k] & 0x80000000u) | (g->mt[(k + 1) % 624] & 0x7fffffffu);
            g->mt[k] = g->mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* libstdc++ 11 generate_canonical<double, 53>(mt19937) (bits/random.tcc:3348-3380):
 * two 32-bit draws, low word first, summed in double, divided by 2^64; a result that
 * rounds up to 1 is replaced by the largest double below 1. */
double orc_mt_canonical(omt19937 *g) {
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; ++k) {
        sum += (double)orc_mt_next(g) * tmp;
        tmp *= 4294967296.0;
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

/* libstdc++ 11 uniform_int_distribution<int>{0,2} on a 32-bit generator
 * (bits/uniform_int_dist.h:241-270,311-317): Lemire's nearly divisionless method.
 * Used by the BVH builder for the split axis (bvh_node.cpp:6,9). */
int orc_mt_uniform_int_0_2(omt19937 *g) {
    const uint32_t range = 3u;
    uint64_t product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)orc_mt_next(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (int)(product >> 32);
}

/* ------------------------------------------------- per-path stream (GPU RNG) - */
void orc_philox4x32_10(const uint32_t key_in[2], const uint32_t ctr_in[4], uint32_t out[4]) {
    uint32_t k0 = key_in[0], k1 = key_in[1];
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_stream_init(ostream *st, uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {pixel, sample, 0x52544957u, 0u};
    orc_philox4x32_10(key, ctr, st->s);
    if ((st->s[0] | st->s[1] | st->s[2] | st->s[3]) == 0) st->s[0] = 1u;
}

static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

/* xoshiro128++ */
uint32_t orc_stream_next(ostream *st) {
    uint32_t *s = st->s;
    uint32_t result = rotl32(s[0] + s[3], 7) + s[0];
    uint32_t t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl32(s[3], 11);
    return result;
}

double orc_stream_canonical(ostream *st) {
    uint64_t hi = orc_stream_next(st);
    uint64_t lo = orc_stream_next(st);
    return (double)(((hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

/* uniform_real_distribution<double>{a,b}: canonical * (b - a) + a (bits/random.h). */
double orc_uniform(orng *g, double a, double b) {
    double u;
    switch (g->kind) {
    case ORC_RNG_MT: u = orc_mt_canonical(g->mt); break;
    case ORC_RNG_STREAM: u = orc_stream_canonical(&g->st); break;
    default:
        if (g->tape_pos < g->tape_len) u = g->tape[g->tape_pos];
        else { u = 0.0; g->tape_overrun = 1; }
        g->tape_pos++;
        break;
    }
    return (u * (b - a)) + a;
}

/* ------------------------------------------------------------------ aabb ---- */
static inline double ffmin(double a, double b) { return a < b ? a : b; }  /* aabb.hpp:6 */
static inline double ffmax(double a, double b) { return a > b ? a : b; }  /* aabb.hpp:7 */

/* aabb.cpp:8-57: three slab tests with early exit; NaN from 0*inf leaves the bound
 * unchanged through ffmin/ffmax. */
int orc_aabb_hit(const oaabb *bx, const oray *r, double tmin, double tmax) {
    g_cnt.box_tests++;
    const double o[3] = {r->o.x, r->o.y, r->o.z}, d[3] = {r->d.x, r->d.y, r->d.z};
    const double mn[3] = {bx->mn.x, bx->mn.y, bx->mn.z}, mx[3] = {bx->mx.x, bx->mx.y, bx->mx.z};
    for (int a = 0; a < 3; ++a) {
        double invD = 1.0 / d[a];
        double t0 = (mn[a] - o[a]) * invD, t1 = (mx[a] - o[a]) * invD;
        if (invD < 0) { double t = t0; t0 = t1; t1 = t; }
        tmin = ffmax(t0, tmin);
        tmax = ffmin(t1, tmax);
        if (tmax <= tmin) return 0;
    }
    return 1;
}

/* aabb.cpp:59-64 */
oaabb orc_surrounding_box(oaabb a, oaabb b) {
    oaabb r;
    r.mn = v3(fmin(a.mn.x, b.mn.x), fmin(a.mn.y, b.mn.y), fmin(a.mn.z, b.mn.z));
    r.mx = v3(fmax(a.mx.x, b.mx.x), fmax(a.mx.y, b.mx.y), fmax(a.mx.z, b.mx.z));
    return r;
}

/* ------------------------------------------------------------- primitives --- */
/* moving_sphere.cpp:3-5 */
ovec3 orc_moving_centre(const oprim *p, double time) {
    return vadd(p->c0, vmul(vsub(p->c1, p->c0), (time - p->t0) / (p->t1 - p->t0)));
}

/* sphere.cpp:9-45 and moving_sphere.cpp:13-49 (identical apart from the centre) */
static int sphere_hit_at(ovec3 centre, double radius, const oray *r, double tmin, double tmax, ohit *out) {
    ovec3 oc = vsub(r->o, centre);
    double a = vdot(r->d, r->d);
    double b = vdot(oc, r->d);
    double c = vdot(oc, oc) - radius * radius;
    double disc = b * b - a * c;
    double root = sqrt(disc);
    if (disc > 0) {
        double t = (-b - root) / a;
        if (t < tmax && t > tmin) {
            out->t = t;
            out->p = vadd(r->o, vmul(r->d, t));          /* ray.cpp:6 */
            out->n = vdiv(vsub(out->p, centre), radius);
            return 1;
        }
        t = (-b + root) / a;
        if (t < tmax && t > tmin) {
            out->t = t;
            out->p = vadd(r->o, vmul(r->d, t));
            out->n = vdiv(vsub(out->p, centre), radius);
            return 1;
        }
    }
    return 0;
}

int orc_prim_hit(const oscene *s, int prim, const oray *r, double tmin, double tmax, ohit *out) {
    const oprim *p = &s->prim[prim];
    int h;
    if (p->kind == ORC_PRIM_SPHERE) {
        g_cnt.sphere_tests++;
        h = sphere_hit_at(p->c0, p->r, r, tmin, tmax, out);
    } else {
        g_cnt.moving_tests++;
        h = sphere_hit_at(orc_moving_centre(p, r->time), p->r, r, tmin, tmax, out);
    }
    if (h) out->prim = prim;
    return h;
}

/* sphere.cpp:47-50; moving_sphere.cpp:51-60 (the second box's max corner uses the
 * centre at t0, as the reference does) */
oaabb orc_prim_bbox(const oprim *p, double t0, double t1) {
    ovec3 rv = v3(p->r, p->r, p->r);
    oaabb b;
    if (p->kind == ORC_PRIM_SPHERE) {
        b.mn = vsub(p->c0, rv); b.mx = vadd(p->c0, rv);
        return b;
    }
    ovec3 c0 = orc_moving_centre(p, t0), c1 = orc_moving_centre(p, t1);
    oaabb b0, b1;
    b0.mn = vsub(c0, rv); b0.mx = vadd(c0, rv);
    b1.mn = vsub(c1, rv); b1.mx = vadd(c0, rv);
    return orc_surrounding_box(b0, b1);
}

/* ------------------------------------------------- libstdc++ std::sort ------ */
/* std::ranges::sort (bvh_node.cpp:10) forwards to std::sort: introsort with
 * median-of-three pivot to the first slot, threshold 16, heap-sort fallback at depth
 * 2*floor(log2 n), then a final insertion sort (bits/stl_algo.h:1799-1959,
 * bits/stl_heap.h).  The sort is not stable, so the order of equal keys -- every
 * small sphere of the random scene has the same box minimum in y -- depends on these
 * exact steps. */
typedef int (*less_fn)(int, int, void *);

static void sift_push(int *f, long hole, long top, int value, less_fn less, void *ctx) {
    long parent = (hole - 1) / 2;
    while (hole > top && less(f[parent], value, ctx)) {
        f[hole] = f[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    f[hole] = value;
}

static void adjust_heap(int *f, long hole, long len, int value, less_fn less, void *ctx) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (less(f[child], f[child - 1], ctx)) child--;
        f[hole] = f[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        f[hole] = f[child - 1];
        hole = child - 1;
    }
    sift_push(f, hole, top, value, less, ctx);
}

static void heap_sort_all(int *f, long len, less_fn less, void *ctx) {
    /* __partial_sort(first, last, last): make_heap over everything, then sort_heap */
    if (len >= 2) {
        long parent = (len - 2) / 2;
        for (;;) {
            int value = f[parent];
            adjust_heap(f, parent, len, value, less, ctx);
            if (parent == 0) break;
            parent--;
        }
    }
    long last = len;
    while (last > 1) {
        --last;
        int value = f[last];
        f[last] = f[0];
        adjust_heap(f, 0, last, value, less, ctx);
    }
}

static void swap_int(int *a, int *b) { int t = *a; *a = *b; *b = t; }

static void median_to_first(int *result, int *a, int *b, int *c, less_fn less, void *ctx) {
    if (less(*a, *b, ctx)) {
        if (less(*b, *c, ctx)) swap_int(result, b);
        else if (less(*a, *c, ctx)) swap_int(result, c);
        else swap_int(result, a);
    } else if (less(*a, *c, ctx)) swap_int(result, a);
    else if (less(*b, *c, ctx)) swap_int(result, c);
    else swap_int(result, b);
}

static int *unguarded_partition(int *first, int *last, int *pivot, less_fn less, void *ctx) {
    for (;;) {
        while (less(*first, *pivot, ctx)) ++first;
        --last;
        while (less(*pivot, *last, ctx)) --last;
        if (!(first < last)) return first;
        swap_int(first, last);
        ++first;
    }
}

static void introsort_loop(int *first, int *last, long depth_limit, less_fn less, void *ctx) {
    while (last - first > 16) {
        if (depth_limit == 0) {
            heap_sort_all(first, last - first, less, ctx);
            return;
        }
        --depth_limit;
        int *mid = first + (last - first) / 2;
        median_to_first(first, first + 1, mid, last - 1, less, ctx);
        int *cut = unguarded_partition(first + 1, last, first, less, ctx);
        introsort_loop(cut, last, depth_limit, less, ctx);
        last = cut;
    }
}

static void unguarded_linear_insert(int *last, less_fn less, void *ctx) {
    int val = *last;
    int *next = last - 1;
    while (less(val, *next, ctx)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

static void insertion_sort(int *first, int *last, less_fn less, void *ctx) {
    if (first == last) return;
    for (int *i = first + 1; i != last; ++i) {
        if (less(*i, *first, ctx)) {
            int val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(int));
            *first = val;
        } else {
            unguarded_linear_insert(i, less, ctx);
        }
    }
}

void orc_introsort(int *v, int n, less_fn less, void *ctx) {
    if (n <= 0) return;
    long lg = 0;
    for (unsigned long m = (unsigned long)n; m > 1; m >>= 1) lg++;
    introsort_loop(v, v + n, lg * 2, less, ctx);
    if (n > 16) {
        insertion_sort(v, v + 16, less, ctx);
        for (int *i = v + 16; i != v + n; ++i) unguarded_linear_insert(i, less, ctx);
    } else {
        insertion_sort(v, v + n, less, ctx);
    }
}

/* ------------------------------------------------------------------- bvh ---- */
typedef struct { const oscene *s; int axis; double t0, t1; } sortctx;

/* object_list::bounding_box (object_list.cpp:28-48): the first member's box, joined with each
 * further member's in order (a list inside a bvh group is never empty: the recipe reader refuses one) */
static oaabb list_bbox(const oscene *s, const olist *l, double t0, double t1) {
    oaabb b = orc_prim_bbox(&s->prim[l->first], t0, t1);
    for (int k = 1; k < l->n; ++k) b = orc_surrounding_box(b, orc_prim_bbox(&s->prim[l->first + k], t0, t1));
    return b;
}

/* bounding box of an object of a bvh group: a primitive or a list */
static oaabb object_bbox(const oscene *s, int obj, double t0, double t1) {
    return obj >= ORC_LIST_BASE ? list_bbox(s, &s->list[obj - ORC_LIST_BASE], t0, t1) : orc_prim_bbox(&s->prim[obj], t0, t1);
}

/* object_list::hit (object_list.cpp:12-26) for a list below a bvh_node: linear scan, the upper
 * bound narrows to the closest hit so far, a later equal t does not replace it */
static int list_hit(const oscene *s, const olist *l, const oray *r, double tmin, double tmax, ohit *out) {
    int any = 0;
    double closest = tmax;
    ohit h;
    for (int k = 0; k < l->n; ++k)
        if (orc_prim_hit(s, l->first + k, r, tmin, closest, &h)) { any = 1; closest = h.t; *out = h; }
    return any;
}

/* bvh_node.cpp:10-28: compare the box minimum on the chosen axis */
static int box_min_less(int l, int r, void *vctx) {
    sortctx *c = (sortctx *)vctx;
    oaabb bl = object_bbox(c->s, l, c->t0, c->t1);
    oaabb br = object_bbox(c->s, r, c->t0, c->t1);
    switch (c->axis) {
    case 0: return bl.mn.x < br.mn.x;
    case 1: return bl.mn.y < br.mn.y;
    default: return bl.mn.z < br.mn.z;
    }
}

static int alloc_node(oscene *s) {
    if (s->n_node == s->cap_node) {
        s->cap_node = s->cap_node ? s->cap_node * 2 : 64;
        s->node = (onode *)realloc(s->node, (size_t)s->cap_node * sizeof(onode));
    }
    return s->n_node++;
}

static oaabb child_box(const oscene *s, int ref, double t0, double t1) {
    return ref >= 0 ? s->node[ref].box : object_bbox(s, ~ref, t0, t1);
}

/* bvh_node.cpp:8-50.  Draw the axis, sort the span in place, then: one object ->
 * both children are that object; two -> one each; otherwise build the left half
 * completely, then the right half (so the generator's draws run in pre-order). */
int orc_bvh_build(oscene *s, int *prims, int n, omt19937 *mt, double t0, double t1) {
    sortctx ctx = {s, orc_mt_uniform_int_0_2(mt), t0, t1};
    orc_introsort(prims, n, box_min_less, &ctx);
    int me = alloc_node(s);
    int left, right;
    if (n == 1) {
        left = right = ~prims[0];
    } else if (n == 2) {
        left = ~prims[0];
        right = ~prims[1];
    } else {
        left = orc_bvh_build(s, prims, n / 2, mt, t0, t1);
        right = orc_bvh_build(s, prims + n / 2, n - n / 2, mt, t0, t1);
    }
    s->node[me].left = left;
    s->node[me].right = right;
    s->node[me].box = orc_surrounding_box(child_box(s, left, t0, t1), child_box(s, right, t0, t1));
    return me;
}

static int child_hit(const oscene *s, int ref, const oray *r, double tmin, double tmax, ohit *out) {
    if (ref >= 0) return orc_bvh_hit(s, ref, r, tmin, tmax, out);
    if (~ref >= ORC_LIST_BASE) return list_hit(s, &s->list[~ref - ORC_LIST_BASE], r, tmin, tmax, out);
    return orc_prim_hit(s, ~ref, r, tmin, tmax, out);
}

/* bvh_node.cpp:52-69: box test, then BOTH children with the same interval; the
 * nearer wins and a tie goes to the right child. */
int orc_bvh_hit(const oscene *s, int node, const oray *r, double tmin, double tmax, ohit *out) {
    const onode *nd = &s->node[node];
    if (!orc_aabb_hit(&nd->box, r, tmin, tmax)) return 0;
    ohit hl, hr;
    int l = child_hit(s, nd->left, r, tmin, tmax, &hl);
    int rr = child_hit(s, nd->right, r, tmin, tmax, &hr);
    if (l && rr) { *out = hl.t < hr.t ? hl : hr; return 1; }
    if (l) { *out = hl; return 1; }
    if (rr) { *out = hr; return 1; }
    return 0;
}

/* object_list.cpp:12-26: linear scan, the upper bound narrows to the closest hit */
int orc_world_hit(const oscene *s, const oray *r, double tmin, double tmax, ohit *out) {
    g_cnt.rays++;
    int any = 0;
    double closest = tmax;
    ohit h;
    for (int i = 0; i < s->n_item; ++i) {
        const oitem *it = &s->item[i];
        int got = it->is_bvh ? orc_bvh_hit(s, it->index, r, tmin, closest, &h)
                             : orc_prim_hit(s, it->index, r, tmin, closest, &h);
        if (got) { any = 1; closest = h.t; *out = h; }
    }
    return any;
}

/* ---------------------------------------------------------------- perlin ---- */
/* glibc drand48 with its default state (no srand48 call): X <- (0x5DEECE66D X + 0xB)
 * mod 2^48 from X = 0 (glibc zero-initialises the state); the result is X / 2^48. */
static uint64_t g_drand_state;
static double lcg48(void) {
    g_drand_state = (g_drand_state * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL;
    return (double)g_drand_state / 281474976710656.0;
}

static double g_ranfloat[256];
static uint8_t g_perm[3][256];
static int g_perlin_ready;

/* perlin.cpp:8-31 and the static-initialisation order of perlin.cpp:50-53 */
static void perlin_init(void) {
    if (g_perlin_ready) return;
    g_drand_state = 0;
    for (int i = 0; i < 256; ++i) g_ranfloat[i] = lcg48();
    for (int t = 0; t < 3; ++t) {
        uint8_t *p = g_perm[t];
        for (int i = 0; i < 256; ++i) p[i] = (uint8_t)i;
        for (int i = 255; i > 0; --i) {
            int target = (int)(lcg48() * (i + 1));
            uint8_t tmp = p[i]; p[i] = p[target]; p[target] = tmp;
        }
    }
    g_perlin_ready = 1;
}

void orc_perlin_tables(double ranfloat[256], uint8_t px[256], uint8_t py[256], uint8_t pz[256]) {
    perlin_init();
    memcpy(ranfloat, g_ranfloat, sizeof g_ranfloat);
    memcpy(px, g_perm[0], 256); memcpy(py, g_perm[1], 256); memcpy(pz, g_perm[2], 256);
}

/* perlin.cpp:33-47,55-73: plain trilinear blend of lattice values, no smoothing */
double orc_perlin_noise(ovec3 p) {
    perlin_init();
    double u = p.x - floor(p.x), v = p.y - floor(p.y), w = p.z - floor(p.z);
    int i = (int)floor(p.x), j = (int)floor(p.y), k = (int)floor(p.z);
    double c[2][2][2];
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk)
                c[di][dj][dk] = g_ranfloat[g_perm[0][(i + di) & 255] ^ g_perm[1][(j + dj) & 255] ^
                                           g_perm[2][(k + dk) & 255]];
    double accum = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int cc = 0; cc < 2; ++cc)
                accum += (a * u + (1 - a) * (1 - u)) * (b * v + (1 - b) * (1 - v)) *
                         (cc * w + (1 - cc) * (1 - w)) * c[a][b][cc];
    return accum;
}

/* texture.hpp:19 (constant), texture.cpp:10-13 (chequer), texture.cpp:15-18 (perlin) */
void orc_texture_value(const oscene *s, int tex, ovec3 p, double out[3]) {
    const otexture *t = &s->tex[tex];
    switch (t->kind) {
    case ORC_TEX_CONSTANT:
        out[0] = t->c[0]; out[1] = t->c[1]; out[2] = t->c[2];
        return;
    case ORC_TEX_CHEQUER: {
        double sines = sin(10 * p.x) * sin(10 * p.y) * sin(10 * p.z);
        orc_texture_value(s, sines < 0 ? t->odd : t->even, p, out);
        return;
    }
    default: {
        double grey = orc_perlin_noise(p);
        out[0] = out[1] = out[2] = grey;
        return;
    }
    }
}

/* -------------------------------------------------------------- materials --- */
/* material.cpp:35-40, distribution {-1,1} (material.cpp:42) */
static ovec3 random_in_unit_sphere(orng *g) {
    if (g->kind == ORC_RNG_STREAM) {
        /* The per-path stream of the GPU library (rt_device.hpp, StreamRng::in_unit_sphere), operation for
         * operation: a point uniform in the unit sphere without rejection.  Direction: two outputs give a
         * point of the square [-1, 1]^2 (24 bits each), mapped onto the sphere by the equal-area octahedral
         * map (Clarberg 2008) in single precision; radius: the largest of three 32-bit uniforms (density
         * 3 r^2), shortened by 2^-21 so that the rounding of the direction cannot carry the point outside.
         * Every operation is a correctly rounded IEEE one on both sides (this file is built with
         * -ffp-contract=off). */
        const uint32_t ha = orc_stream_next(&g->st), hb = orc_stream_next(&g->st);
        const float a = fmaf((float)(ha >> 8), 0x1p-23f, 0x1p-24f - 1.0f), b = fmaf((float)(hb >> 8), 0x1p-23f, 0x1p-24f - 1.0f);
        const float aa = fabsf(a), ab = fabsf(b);
        const float d = 1.0f - (aa + ab), r = 1.0f - fabsf(d);
        const float th = (r > 0 ? (ab - aa) / r : 0.0f) * 0.78539816339744830962f, q = th * th;
        const float f = r * sqrtf(fmaf(-r, r, 2.0f)) * 0.70710678118654752440f;
        float cs = 2.4801587e-05f;
        cs = fmaf(cs, q, -0.0013888889f); cs = fmaf(cs, q, 0.041666668f); cs = fmaf(cs, q, -0.5f); cs = fmaf(cs, q, 1.0f);
        float sn = 2.7557319e-06f;
        sn = fmaf(sn, q, -0.00019841270f); sn = fmaf(sn, q, 0.0083333333f); sn = fmaf(sn, q, -0.16666667f); sn = fmaf(sn * q, th, th);
        const float fx = copysignf(f * (cs - sn), a), fy = copysignf(f * (cs + sn), b), fz = copysignf(fmaf(-r, r, 1.0f), d);
        const uint32_t w0 = orc_stream_next(&g->st), w1 = orc_stream_next(&g->st), w2 = orc_stream_next(&g->st);
        const uint32_t w = w0 > w1 ? (w0 > w2 ? w0 : w2) : (w1 > w2 ? w1 : w2);
        const double rho = fma((double)w, 0x1p-32 * (1.0 - 0x1p-21), 0x1p-33);
        ovec3 c;
        c.x = (double)fx * rho; c.y = (double)fy * rho; c.z = (double)fz * rho;
        return c;
    }
    for (;;) {
        ovec3 c;
        c.x = orc_uniform(g, -1, 1);
        c.y = orc_uniform(g, -1, 1);
        c.z = orc_uniform(g, -1, 1);
        if (c.x * c.x + c.y * c.y + c.z * c.z < 1) return c;
        if (g->kind == ORC_RNG_TAPE && g->tape_overrun) return c;
    }
}

/* n points of random_in_unit_sphere over one per-path stream (test access to the stream's sampler) */
void orc_stream_unit_sphere(uint64_t seed, uint32_t pixel, uint32_t sample, size_t n, double *out) {
    orng g;
    memset(&g, 0, sizeof g);
    g.kind = ORC_RNG_STREAM;
    orc_stream_init(&g.st, seed, pixel, sample);
    for (size_t i = 0; i < n; ++i) {
        ovec3 c = random_in_unit_sphere(&g);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

/* material.cpp:10-12 */
static ovec3 reflect(ovec3 v, ovec3 n) { return vsub(v, vmul(n, 2 * vdot(v, n))); }

/* material.cpp:14-24 */
static int refract(ovec3 v, ovec3 n, double ni_over_nt, ovec3 *out) {
    ovec3 uv = vunit(v);
    double dt = vdot(uv, n);
    double disc = 1 - ni_over_nt * ni_over_nt * (1 - dt * dt);
    if (disc > 0) {
        *out = vsub(vmul(vsub(uv, vmul(n, dt)), ni_over_nt), vmul(n, sqrt(disc)));
        return 1;
    }
    return 0;
}

/* material.cpp:26-30 */
static double schlick(double cosine, double ri) {
    double r0 = (1 - ri) / (1 + ri);
    double r1 = r0 * r0;
    return r1 + (1 - r1) * pow(1 - cosine, 5);
}

int orc_scatter(const oscene *s, int mat, const oray *r, const ohit *h, orng *g, double att[3], oray *sc) {
    const omaterial *m = &s->mat[mat];
    switch (m->kind) {
    case ORC_MAT_LAMBERTIAN: { /* material.cpp:47-52 */
        ovec3 target = vadd(vadd(h->p, h->n), random_in_unit_sphere(g));
        orc_texture_value(s, m->tex, h->p, att);
        sc->o = h->p; sc->d = vsub(target, h->p); sc->time = r->time;
        return 1;
    }
    case ORC_MAT_METAL: { /* material.cpp:59-67 */
        ovec3 reflected = reflect(vunit(r->d), h->n);
        sc->o = h->p;
        sc->d = vadd(reflected, vmul(random_in_unit_sphere(g), m->fuzz));
        sc->time = r->time;
        att[0] = m->albedo[0]; att[1] = m->albedo[1]; att[2] = m->albedo[2];
        return vdot(sc->d, h->n) > 0;
    }
    default: { /* material.cpp:77-98 */
        double dn = vdot(r->d, h->n);
        int simple = dn > 0;
        ovec3 outward = simple ? vneg(h->n) : h->n;
        ovec3 reflected = reflect(r->d, h->n);
        double ni_over_nt = simple ? m->ri : 1 / m->ri;
        double cosine = simple ? m->ri * vdot(r->d, h->n) / vlen(r->d) : -vdot(r->d, h->n) / vlen(r->d);
        ovec3 refracted;
        int has = refract(r->d, outward, ni_over_nt, &refracted);
        double reflect_prob = has ? schlick(cosine, m->ri) : 1.0;
        att[0] = att[1] = att[2] = 1;
        sc->o = h->p; sc->time = r->time;
        sc->d = orc_uniform(g, 0, 1) < reflect_prob ? reflected : refracted;
        return 1;
    }
    }
}

/* ----------------------------------------------------------------- camera --- */
/* camera.cpp:6-14,25-36 */
void orc_camera_init(ocamera *c, ovec3 from, ovec3 at, ovec3 up, double vfov, double aspect,
                     double aperture, double focus_dist, double t0, double t1) {
    double theta = vfov * M_PI / 180;
    double hh = tan(theta / 2);
    double hw = aspect * hh;
    c->origin = from;
    c->w = vunit(vsub(from, at));
    c->u = vunit(vcross(up, c->w));
    c->v = vcross(c->w, c->u);
    c->llc = vsub(vsub(vsub(from, vmul(c->u, hw * focus_dist)), vmul(c->v, hh * focus_dist)),
                  vmul(c->w, focus_dist));
    c->horizontal = vmul(c->u, 2 * hw * focus_dist);
    c->vertical = vmul(c->v, 2 * hh * focus_dist);
    c->lens_radius = aperture / 2;
    c->t0 = t0; c->t1 = t1;
}

/* camera.cpp:16-21,38-48: the lens disk is sampled even when the radius is zero, and
 * the shutter time reuses the {-1,1} distribution */
oray orc_camera_get_ray(const ocamera *c, double s, double t, orng *g) {
    ovec3 p = {0, 0, 0};
    /* The GPU's per-path streams skip the lens sample when the radius is zero (its value is
     * multiplied by the radius); the shared-mt19937 and tape modes draw it as the reference does. */
    const int skip = g->kind == ORC_RNG_STREAM && c->lens_radius == 0;
    while (!skip) {
        p.x = orc_uniform(g, -1, 1);
        p.y = orc_uniform(g, -1, 1);
        p.z = 0;
        if (p.x * p.x + p.y * p.y + p.z * p.z < 1) break;
        if (g->kind == ORC_RNG_TAPE && g->tape_overrun) break;
    }
    ovec3 rd = vmul(p, c->lens_radius);
    ovec3 offset = vadd(vmul(c->u, rd.x), vmul(c->v, rd.y));
    double time = c->t0 + orc_uniform(g, -1, 1) * (c->t1 - c->t0);
    oray r;
    r.o = vadd(c->origin, offset);
    r.d = vsub(vsub(vadd(vadd(c->llc, vmul(c->horizontal, s)), vmul(c->vertical, t)), c->origin), offset);
    r.time = time;
    return r;
}

static void camera_from_array(ocamera *c, const double *a) {
    orc_camera_init(c, v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), v3(a[6], a[7], a[8]),
                    a[9], a[10], a[11], a[12], a[13], a[14]);
}

/* ------------------------------------------------------------- integrator --- */
/* main.cpp:30-50: scatter is evaluated before the depth test; the product with the
 * attenuation is taken on the way back up the recursion. */
static int g_max_depth = 50;   /* the constant of main.cpp:34 */
void orc_set_max_depth(int depth) { g_max_depth = depth; }

void orc_ray_colour(const oscene *s, const oray *r, int depth, orng *g, double out[3]) {
    ohit h;
    if (orc_world_hit(s, r, 0.001, DBL_MAX, &h)) {
        double att[3];
        oray sc;
        int scattered = orc_scatter(s, s->prim[h.prim].mat, r, &h, g, att, &sc);
        if (depth < g_max_depth && scattered) {
            double next[3];
            orc_ray_colour(s, &sc, depth + 1, g, next);
            out[0] = next[0] * att[0]; out[1] = next[1] * att[1]; out[2] = next[2] * att[2];
        } else {
            out[0] = out[1] = out[2] = 0;
        }
    } else {
        ovec3 unit = vunit(r->d);
        double t = 0.5 * (unit.y + 1.0);
        out[0] = (1.0 - t) * 1.0 + t * 0.5;
        out[1] = (1.0 - t) * 1.0 + t * 0.7;
        out[2] = (1.0 - t) * 1.0 + t * 1.0;
    }
}

/* main.cpp:126-141 + supersampler.cpp:5-10 */
void orc_render(const oscene *s, const double *cam, int nx, int ny, int s_begin, int s_end,
                uint64_t seed, int mode, double *out_sum) {
    ocamera c;
    camera_from_array(&c, cam);
    omt19937 mt;
    orng g;
    memset(&g, 0, sizeof g);
    g.kind = mode;
    if (mode == ORC_RNG_MT) { orc_mt_seed(&mt, (uint32_t)seed); g.mt = &mt; }
    for (int j = ny - 1; j >= 0; --j) {
        for (int i = 0; i < nx; ++i) {
            size_t pix = (size_t)(ny - 1 - j) * (size_t)nx + (size_t)i;
            double sr = 0, sg = 0, sb = 0;
            for (int smp = s_begin; smp < s_end; ++smp) {
                if (mode == ORC_RNG_STREAM) orc_stream_init(&g.st, seed, (uint32_t)pix, (uint32_t)smp);
                double ii = i + orc_uniform(&g, 0, 1);
                double jj = j + orc_uniform(&g, 0, 1);
                double u = ii / (double)nx, v = jj / (double)ny;
                oray r = orc_camera_get_ray(&c, u, v, &g);
                double col[3];
                orc_ray_colour(s, &r, 0, &g, col);
                g_cnt.paths++;
                sr += col[0]; sg += col[1]; sb += col[2];
            }
            out_sum[pix * 3 + 0] = sr; out_sum[pix * 3 + 1] = sg; out_sum[pix * 3 + 2] = sb;
        }
    }
}

/* ---------------------------------------------------------- batch wrappers -- */
int orc_scene_counts(const oscene *s, int *n_prim, int *n_node, int *n_item) {
    if (n_prim) *n_prim = s->n_prim;
    if (n_node) *n_node = s->n_node;
    if (n_item) *n_item = s->n_item;
    return 0;
}

void orc_scene_nodes(const oscene *s, double *boxes, int32_t *children) {
    for (int i = 0; i < s->n_node; ++i) {
        const onode *n = &s->node[i];
        double *b = boxes + (size_t)i * 6;
        b[0] = n->box.mn.x; b[1] = n->box.mn.y; b[2] = n->box.mn.z;
        b[3] = n->box.mx.x; b[4] = n->box.mx.y; b[5] = n->box.mx.z;
        children[2 * i] = n->left; children[2 * i + 1] = n->right;
    }
}

static oray ray_from_array(const double *a) {
    oray r;
    r.o = v3(a[0], a[1], a[2]); r.d = v3(a[3], a[4], a[5]); r.time = a[6];
    return r;
}

static void ray_to_array(const oray *r, double *a) {
    a[0] = r->o.x; a[1] = r->o.y; a[2] = r->o.z;
    a[3] = r->d.x; a[4] = r->d.y; a[5] = r->d.z; a[6] = r->time;
}

void orc_trace_batch(const oscene *s, const double *rays, size_t n, double tmin, double tmax,
                     uint8_t *hit, int32_t *prim, double *tpn) {
    for (size_t i = 0; i < n; ++i) {
        oray r = ray_from_array(rays + i * 7);
        ohit h;
        double *o = tpn + i * 7;
        if (orc_world_hit(s, &r, tmin, tmax, &h)) {
            hit[i] = 1; prim[i] = h.prim;
            o[0] = h.t; o[1] = h.p.x; o[2] = h.p.y; o[3] = h.p.z; o[4] = h.n.x; o[5] = h.n.y; o[6] = h.n.z;
        } else {
            hit[i] = 0; prim[i] = -1;
            for (int k = 0; k < 7; ++k) o[k] = 0;
        }
    }
}

void orc_aabb_batch(const double *cases, size_t n, uint8_t *hit) {
    for (size_t i = 0; i < n; ++i) {
        const double *c = cases + i * 14;
        oaabb b;
        b.mn = v3(c[0], c[1], c[2]); b.mx = v3(c[3], c[4], c[5]);
        oray r;
        r.o = v3(c[6], c[7], c[8]); r.d = v3(c[9], c[10], c[11]); r.time = 0;
        hit[i] = (uint8_t)orc_aabb_hit(&b, &r, c[12], c[13]);
    }
}

void orc_texture_batch(const oscene *s, int tex, const double *p, size_t n, double *out) {
    for (size_t i = 0; i < n; ++i) orc_texture_value(s, tex, v3(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]), out + i * 3);
}

void orc_scatter_batch(const oscene *s, const int32_t *prim, const double *rays, const double *hits,
                       const double *tape, int k, size_t n, uint8_t *scattered, double *att,
                       double *out_rays, int32_t *used) {
    for (size_t i = 0; i < n; ++i) {
        oray r = ray_from_array(rays + i * 7);
        const double *hh = hits + i * 7;
        ohit h;
        h.t = hh[0]; h.p = v3(hh[1], hh[2], hh[3]); h.n = v3(hh[4], hh[5], hh[6]); h.prim = prim[i];
        orng g;
        memset(&g, 0, sizeof g);
        g.kind = ORC_RNG_TAPE; g.tape = tape + i * (size_t)k; g.tape_len = k;
        oray sc;
        memset(&sc, 0, sizeof sc);
        scattered[i] = (uint8_t)orc_scatter(s, s->prim[prim[i]].mat, &r, &h, &g, att + i * 3, &sc);
        ray_to_array(&sc, out_rays + i * 7);
        used[i] = g.tape_overrun ? -1 : g.tape_pos;
    }
}

void orc_camera_batch(const double *cam, const double *st, const double *tape, int k, size_t n,
                      double *out_rays, int32_t *used) {
    ocamera c;
    camera_from_array(&c, cam);
    for (size_t i = 0; i < n; ++i) {
        orng g;
        memset(&g, 0, sizeof g);
        g.kind = ORC_RNG_TAPE; g.tape = tape + i * (size_t)k; g.tape_len = k;
        oray r = orc_camera_get_ray(&c, st[i * 2], st[i * 2 + 1], &g);
        ray_to_array(&r, out_rays + i * 7);
        used[i] = g.tape_overrun ? -1 : g.tape_pos;
    }
}
