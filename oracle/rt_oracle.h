/*
 * rt_oracle.h -- CPU restatement (plain C, double precision) of the reference C++
 * tracer's per-pixel path-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE.  It is the checker the HIP path is compared against;
 * it is never shipped, never linked into the product libraries, and never used as a
 * fallback.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against the
 * compiled reference (oracle/_ref/ref_capture, built from /root/reference/c++ by
 * oracle/Makefile) through the golden vectors committed under tests/golden/.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/c++/).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double x, y, z; } ovec3;                 /* vec3.hpp:3-24 */
typedef struct { ovec3 o, d; double time; } oray;         /* ray.hpp:5-15 */
typedef struct { ovec3 mn, mx; } oaabb;                   /* aabb.hpp:9-20 */
typedef struct { double t; ovec3 p, n; int prim; } ohit;  /* object.hpp:10-15; prim stands for the material pointer */

enum { ORC_TEX_CONSTANT = 0, ORC_TEX_CHEQUER = 1, ORC_TEX_PERLIN = 2 };
enum { ORC_MAT_LAMBERTIAN = 0, ORC_MAT_METAL = 1, ORC_MAT_DIELECTRIC = 2 };
enum { ORC_PRIM_SPHERE = 0, ORC_PRIM_MOVING_SPHERE = 1 };

typedef struct { int kind; double c[3]; int even, odd; } otexture;
typedef struct { int kind; int tex; double albedo[3]; double fuzz; double ri; } omaterial;
typedef struct { int kind; ovec3 c0, c1; double t0, t1, r; int mat; int list; /* object_list it belongs to inside a bvh group, -1 = none */ } oprim;
/* an object_list handed to a bvh_node as one of its objects (object_list.hpp:8-17): primitives [first, first + n) */
typedef struct { int first, n; } olist;
/* An object of a bvh group is a primitive (its index) or a list (ORC_LIST_BASE + its index).
 * child reference of a node: >= 0 -> node index, < 0 -> ~object */
enum { ORC_LIST_BASE = 1 << 30 };
typedef struct { oaabb box; int left, right; } onode;
/* one entry of the top-level object_list: a primitive or the root of a bvh_node */
typedef struct { int is_bvh; int index; uint32_t seed; double t0, t1; int first_prim, n_prims; } oitem;

typedef struct {
    otexture *tex;  int n_tex;
    omaterial *mat; int n_mat;
    oprim *prim;    int n_prim;
    olist *list;    int n_list;
    onode *node;    int n_node, cap_node;
    oitem *item;    int n_item;
} oscene;

/* per-run work counters (reference-semantics traversal), used for the roofline's
 * algorithmic bytes per path */
typedef struct {
    uint64_t box_tests, sphere_tests, moving_tests, rays, paths;
} ocounters;

/* ---- random sources -------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } omt19937;
void     orc_mt_seed(omt19937 *g, uint32_t seed);
uint32_t orc_mt_next(omt19937 *g);
double   orc_mt_canonical(omt19937 *g);     /* libstdc++ generate_canonical<double,53> */
int      orc_mt_uniform_int_0_2(omt19937 *g);/* libstdc++ uniform_int_distribution<int>{0,2} */

/* counter-based per-path stream used by the GPU path: Philox4x32-10 keyed by the
 * render seed over the counter (pixel, sample) gives the 128-bit start state of a
 * xoshiro128++ generator; doubles take 53 bits of two successive outputs. */
typedef struct { uint32_t s[4]; } ostream;
void     orc_philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]);
void     orc_stream_init(ostream *st, uint64_t seed, uint32_t pixel, uint32_t sample);
uint32_t orc_stream_next(ostream *st);
double   orc_stream_canonical(ostream *st);
void     orc_stream_unit_sphere(uint64_t seed, uint32_t pixel, uint32_t sample, size_t n, double *out /*[n][3]*/);

/* a draw source: shared mt19937, per-path stream, or a tape of given uniforms */
enum { ORC_RNG_MT = 0, ORC_RNG_STREAM = 1, ORC_RNG_TAPE = 2 };
typedef struct {
    int kind;
    omt19937 *mt;
    ostream st;
    const double *tape; int tape_len, tape_pos; int tape_overrun;
} orng;
double orc_uniform(orng *g, double a, double b);   /* uniform_real_distribution<double>{a,b} */

/* ---- geometry -------------------------------------------------------------- */
int   orc_aabb_hit(const oaabb *b, const oray *r, double tmin, double tmax);   /* aabb.cpp:8-57 */
oaabb orc_surrounding_box(oaabb a, oaabb b);                                   /* aabb.cpp:59-64 */
ovec3 orc_moving_centre(const oprim *p, double time);                          /* moving_sphere.cpp:3-5 */
int   orc_prim_hit(const oscene *s, int prim, const oray *r, double tmin, double tmax, ohit *out);
oaabb orc_prim_bbox(const oprim *p, double t0, double t1);                     /* sphere.cpp:47-50, moving_sphere.cpp:51-60 */
int   orc_bvh_build(oscene *s, int *prims, int n, omt19937 *mt, double t0, double t1); /* bvh_node.cpp:8-50 */
int   orc_bvh_hit(const oscene *s, int node, const oray *r, double tmin, double tmax, ohit *out); /* bvh_node.cpp:52-69 */
int   orc_world_hit(const oscene *s, const oray *r, double tmin, double tmax, ohit *out);        /* object_list.cpp:12-26 */
void  orc_introsort(int *v, int n, int (*less)(int, int, void *), void *ctx);  /* libstdc++ std::sort */

/* ---- shading --------------------------------------------------------------- */
void   orc_perlin_tables(double ranfloat[256], uint8_t px[256], uint8_t py[256], uint8_t pz[256]); /* perlin.cpp:8-31,50-53 */
double orc_perlin_noise(ovec3 p);                                              /* perlin.cpp:55-73 */
void   orc_texture_value(const oscene *s, int tex, ovec3 p, double out[3]);    /* texture.hpp:19, texture.cpp:10-18 */
int    orc_scatter(const oscene *s, int mat, const oray *r, const ohit *h, orng *g,
                   double att[3], oray *scattered);                            /* material.cpp:47-98 */

/* ---- camera / integrator --------------------------------------------------- */
typedef struct {
    ovec3 origin, w, u, v, llc, horizontal, vertical;
    double lens_radius, t0, t1;
} ocamera;
void orc_camera_init(ocamera *c, ovec3 from, ovec3 at, ovec3 up, double vfov, double aspect,
                     double aperture, double focus_dist, double t0, double t1);  /* camera.cpp:25-36 */
oray orc_camera_get_ray(const ocamera *c, double s, double t, orng *g);          /* camera.cpp:38-48 */
void orc_ray_colour(const oscene *s, const oray *r, int depth, orng *g, double out[3]); /* main.cpp:30-50 */

/* ---- scene recipes (text) --------------------------------------------------- */
oscene *orc_scene_load(const char *path, char *err, size_t errlen);
void    orc_scene_free(oscene *s);

/* ---- flat-array entry points for ctypes ------------------------------------- */
void orc_counters_reset(void);
void orc_counters_get(ocounters *out);
int  orc_scene_counts(const oscene *s, int *n_prim, int *n_node, int *n_item);
/* dumps nodes in index order: box[6] (min xyz, max xyz), left, right */
void orc_scene_nodes(const oscene *s, double *boxes, int32_t *children);
void orc_trace_batch(const oscene *s, const double *rays /*[n][7]*/, size_t n, double tmin, double tmax,
                     uint8_t *hit, int32_t *prim, double *tpn /*[n][7]*/);
void orc_aabb_batch(const double *cases /*[n][14]: box6 ray o3 d3 tmin tmax*/, size_t n, uint8_t *hit);
void orc_texture_batch(const oscene *s, int tex, const double *p /*[n][3]*/, size_t n, double *out /*[n][3]*/);
void orc_scatter_batch(const oscene *s, const int32_t *prim, const double *rays /*[n][7]*/,
                       const double *hits /*[n][7] t p n*/, const double *tape /*[n][k]*/, int k, size_t n,
                       uint8_t *scattered, double *att /*[n][3]*/, double *out_rays /*[n][7]*/, int32_t *used);
void orc_camera_batch(const double *cam /*[10]: from at up? see .c*/, const double *st /*[n][2]*/,
                      const double *tape, int k, size_t n, double *out_rays /*[n][7]*/, int32_t *used);
/* The bounce limit of main.cpp:34 (50 there); settable so that tests can compare renders at
 * other limits.  Process-wide. */
void orc_set_max_depth(int depth);
/* render samples [s_begin, s_end) of every pixel.
 * mode ORC_RNG_MT: one shared mt19937(seed), draw order of main.cpp:126-141 (s_begin must be 0).
 * mode ORC_RNG_STREAM: per-(pixel, sample) streams, pixel = row-major index with the top row first.
 * out: per-pixel SUMS (not means) of the samples, top row first, rgb doubles. */
void orc_render(const oscene *s, const double *cam /*[16]*/, int nx, int ny, int s_begin, int s_end,
                uint64_t seed, int mode, double *out_sum);

#ifdef __cplusplus
}
#endif
#endif
