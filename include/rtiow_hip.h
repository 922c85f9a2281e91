/*
 * rtiow_hip.h -- C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * The reference tracer (hgiddens/ray-tracing-in-one-weekend, c++/) has no plugin or
 * FFI layer: its hot path is reached through in-process C++ virtual calls.  This
 * header is the boundary placed directly beneath that class surface.  Each entry
 * point names the reference interface it stands in for (file:line under c++/).
 *
 * Conventions: plain C structs, sizes and pointers; opaque scene handle; every
 * function returns 0 on success and a non-zero RT_E_* code on failure, with the
 * message available from rt_last_error() (thread-local).  No exceptions cross the
 * boundary.  A scene lives on one GPU (chosen at rt_scene_create); one host thread
 * or process per GPU may use its own scenes concurrently.  Wherever a function takes a
 * `device`, -1 stands for the calling thread's current HIP device.
 *
 * All arithmetic is IEEE double precision, evaluated in the reference's operation
 * order without fused multiply-add, so hit/miss decisions, primitive ids, t, p and
 * normals are bit-identical to the reference's on the same inputs.
 */
#ifndef RTIOW_HIP_H
#define RTIOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RT_OK = 0,
    RT_E_INVALID = 1,      /* bad argument or malformed scene description */
    RT_E_DEVICE = 2,       /* HIP runtime error */
    RT_E_UNSUPPORTED = 3,  /* scene shape outside what the kernels handle */
    RT_E_TAPE = 4          /* a uniform tape was too short for at least one item */
};

enum { RT_TEX_CONSTANT = 0, RT_TEX_CHEQUER = 1, RT_TEX_PERLIN = 2 };      /* texture.hpp:15-33 */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2 };  /* material.hpp:31-53 */
enum { RT_PRIM_SPHERE = 0, RT_PRIM_MOVING_SPHERE = 1 };                   /* sphere.hpp, moving_sphere.hpp */
enum { RT_ACCUM_F32 = 0, RT_ACCUM_F64 = 1 };

/* ray.hpp:5-15 -- the direction is not normalised */
typedef struct rt_ray { double origin[3]; double direction[3]; double time; } rt_ray;

/* object.hpp:10-15 -- `prim` (index of the primitive in the scene description) takes
 * the place of the material pointer: every primitive owns exactly one material */
typedef struct rt_hit { int32_t hit; int32_t prim; double t; double p[3]; double normal[3]; } rt_hit;

/* texture.hpp:15-33; even/odd index earlier entries of the texture table */
typedef struct rt_texture_desc { uint32_t kind; uint32_t even, odd; uint32_t pad; double colour[3]; } rt_texture_desc;

/* material.hpp:31-53 */
typedef struct rt_material_desc { uint32_t kind; uint32_t texture; double albedo[3]; double fuzziness; double refractive_index; } rt_material_desc;

/* sphere.hpp:8-18, moving_sphere.hpp:8-21; a static sphere uses centre0 only */
typedef struct rt_prim_desc {
    uint32_t kind; uint32_t material;
    double centre0[3]; double centre1[3];
    double time0, time1, radius;
} rt_prim_desc;

/* bvh_node.hpp:14-15 -- the host-built tree, node by node.  A child >= 0 is a node
 * index, a child < 0 is ~primitive index.  A node built over one object has
 * left == right (bvh_node.cpp:31-34). */
typedef struct rt_node_desc { double box_min[3]; double box_max[3]; int32_t left, right; } rt_node_desc;

/* object_list.hpp:8-17 -- the world is a list of items, each a primitive or the root
 * node of a bvh */
typedef struct rt_item_desc { uint32_t is_bvh; int32_t index; } rt_item_desc;

typedef struct rt_scene_desc {
    const rt_texture_desc *textures;   uint32_t n_textures;
    const rt_material_desc *materials; uint32_t n_materials;
    const rt_prim_desc *prims;         uint32_t n_prims;
    const rt_node_desc *nodes;         uint32_t n_nodes;
    const rt_item_desc *items;         uint32_t n_items;
    /* perlin.hpp:10-11 -- 256 lattice values and three byte permutations */
    const double *perlin_ranfloat;
    const uint8_t *perlin_perm_x, *perlin_perm_y, *perlin_perm_z;
} rt_scene_desc;

/* camera.hpp:10-18 -- the state camera::camera computes on the host (camera.cpp:25-36) */
typedef struct rt_camera {
    double origin[3], u[3], v[3], w[3];
    double lower_left_corner[3], horizontal[3], vertical[3];
    double lens_radius, time0, time1;
} rt_camera;

/* the constants of main.cpp:23,34,108 and the sample range this GPU renders */
typedef struct rt_render_params {
    uint32_t nx, ny;
    uint32_t sample_begin, sample_end;  /* global sample indices [begin, end) of every pixel */
    uint32_t max_depth;                 /* 50 in the reference (main.cpp:34) */
    uint32_t accum_type;                /* RT_ACCUM_F32 or RT_ACCUM_F64 */
    double t_min;                       /* 0.001 in the reference (main.cpp:23) */
    uint64_t seed;                      /* keys the per-(pixel, sample) random streams */
} rt_render_params;

typedef struct rt_stats { uint64_t paths; uint64_t rays; double kernel_ms; } rt_stats;

typedef struct rt_scene rt_scene;

int rt_device_count(void);
const char *rt_last_error(void);

/* Copies the description to `device`, laid out for the kernels.  Replaces building
 * the object graph that main.cpp:113 hands to ray_colour. */
int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);

/* object::hit over a batch (object.hpp:20; object_list.cpp:12-26, bvh_node.cpp:52-69,
 * sphere.cpp:9-45, moving_sphere.cpp:13-49).  Host buffers. */
int rt_trace_batch(rt_scene *scene, const rt_ray *rays, size_t n, double t_min, double t_max, rt_hit *out);

/* aabb::hit over a batch (aabb.cpp:8-57): boxes[n][6] = min xyz, max xyz; host buffers */
int rt_aabb_hit_batch(int device, const double *boxes, const rt_ray *rays, const double *t_min,
                      const double *t_max, size_t n, uint8_t *out);

/* texture::value(0, 0, p) over a batch (texture.hpp:12; texture.cpp:10-18); host buffers */
int rt_texture_batch(rt_scene *scene, uint32_t texture, const double *p, size_t n, double *rgb);

/* material::scatter over a batch (material.hpp:28; material.cpp:47-98).  Item i scatters
 * off the material of primitive prim[i]; the uniforms it consumes are read in order
 * from tape[i*k .. i*k+k) (canonical values in [0,1), as std::generate_canonical
 * yields them).  used[i] is the count consumed, or -1 when k was too small (the call
 * then returns RT_E_TAPE).  Host buffers. */
int rt_scatter_batch(rt_scene *scene, const int32_t *prim, const rt_ray *rays, const rt_hit *hits,
                     const double *tape, uint32_t k, size_t n, uint8_t *scattered,
                     double *attenuation, rt_ray *out, int32_t *used);

/* camera::get_ray over a batch (camera.hpp:27; camera.cpp:16-21,38-48); st[n][2];
 * tape as in rt_scatter_batch.  Host buffers. */
int rt_camera_batch(int device, const rt_camera *camera, const double *st, const double *tape,
                    uint32_t k, size_t n, rt_ray *out, int32_t *used);

/* The pixel/sample loop of main.cpp:126-141 with ray_colour (main.cpp:30-50) and the
 * supersampler's running sums (supersampler.cpp:5-10).  ADDS the radiance of samples
 * [sample_begin, sample_end) of every pixel into accum (DEVICE memory on the scene's
 * GPU, nx*ny*3 values of accum_type, row-major, top image row first), asynchronously
 * on `stream` (a hipStream_t, or NULL).  stats may be NULL; when given, the call
 * synchronises the stream and fills it.  At most 2^24 samples per pixel per call.  A scene
 * carries the frame's tile queue and tile order: renders of ONE scene must not overlap in
 * time (use one scene per stream if they must).  Frames of 32 samples per pixel or more are
 * preceded by a short costing pass on the same stream (tiles are then rendered dearest
 * first); its two small device buffers are allocated on the first such call for an image
 * size, which may synchronise the device.  A frame of the same camera, size and depth limit
 * as the scene's last costed frame reuses that frame's tile order and skips the pass.  The
 * image does not depend on any of this: a pixel's samples are summed as 64-bit integers in
 * units of 2^-50 (exact and order-independent; one rounding, when the sum is added to accum),
 * so tile sizes, kernel shapes and schedules all give the same bits.  Calls deeper than 4096
 * samples per pixel are split into several launches.  A scene with an albedo or texture
 * colour outside [0, 1] -- radiance then has no bound -- is summed in doubles instead, in an
 * order that is fixed for a given GPU.  A sample whose radiance is not finite leaves its
 * pixel's sum undefined (as main.cpp:26-28 is for such a pixel). */
int rt_render(rt_scene *scene, const rt_camera *camera, const rt_render_params *params,
              void *accum_dev, void *stream, rt_stats *stats);

/* The pixel/sample loop of main.cpp:126-141 on several GPUs of one node, split by sample index
 * as the reference's README proposes for threads (README.md:21-23: every worker renders the
 * whole image with its own samples, the results are merged).  scenes[g] is the same world
 * uploaded to the g-th GPU of the caller's choice (rt_scene_create with that device); scene g
 * renders the samples [b + g S / n, b + (g + 1) S / n) of every pixel, b = sample_begin and
 * S = sample_end - sample_begin, on its own host thread and stream.  The per-GPU sums are
 * added into the first scene's GPU with one RCCL reduction (ncclReduce, sum, over xGMI) and
 * copied to accum_host (nx*ny*3 sums of params->accum_type; what it held before is
 * overwritten).  The random streams are keyed by (pixel, global sample index): the set of
 * paths is the one a single GPU traces for the whole range.  Scenes that share a GPU render
 * one after the other into that GPU's buffer; with a single GPU in the list no collective
 * is run and RCCL is not loaded (librccl.so.1 is opened on first use).
 * stats: paths and rays summed over the scenes, kernel_ms the largest per-GPU render time;
 * may be NULL. */
int rt_render_multi(rt_scene *const *scenes, uint32_t n_scenes, const rt_camera *camera,
                    const rt_render_params *params, void *accum_host, rt_stats *stats);

/* rt_render for callers that cannot allocate.  It has no counterpart in the reference: it
 * is for code under HIP graph capture, tight serving loops and harnesses that forbid
 * hipMalloc.  Same arguments and same image as rt_render, bit for bit, without the
 * statistics (no events are created, nothing is synchronised).  The frame's working memory
 * -- the four frame counters, the tile costs and tile order, and the deep stack entries of
 * trees too large for LDS -- lies in a 32 MiB array that is part of the library's device
 * image (every GPU the library runs on holds it until the process ends), so the call makes
 * no hipMalloc / hipFree.  Limits, reported as RT_E_UNSUPPORTED: frames of 32 or more samples
 * per pixel (those whose tiles are ordered by cost) may have at most 2^19 - 8 tiles, counted in
 * tiles of the size the frame is rendered in -- 4x4 pixels unless the RTIOW_TILE_SHIFT switch
 * says otherwise, i.e. about 8.3 M pixels: 3840x2160 fits -- and the scene's stack spill at
 * most 28 MiB less 256 bytes (what the array has left beside the counters and the 4 MiB of tile
 * tables; the 10^6-sphere tree needs 17 MiB).  All calls
 * on one GPU share the array, whichever scene they render: calls that overlap in time on one
 * GPU race, also with each other across streams; rt_render calls do not use it. */
int rt_render_noalloc(rt_scene *scene, const rt_camera *camera, const rt_render_params *params,
                      void *accum_dev, void *stream);

/* Same as rt_render, into a host buffer that is zeroed first (allocates and frees device memory). */
int rt_render_host(rt_scene *scene, const rt_camera *camera, const rt_render_params *params,
                   void *accum_host, rt_stats *stats);

/* supersampler::value, colour::gamma2 and convert (supersampler.cpp:12-19,
 * colour.cpp:19-21, main.cpp:26-28): rgb8 = int(255.99 * sqrt(sum / spp)).  Device
 * buffers on `device`; float sums in, bytes out. */
int rt_resolve(int device, const float *accum_dev, size_t npix, uint32_t spp, uint8_t *rgb8_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif
