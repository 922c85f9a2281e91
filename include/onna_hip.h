/*
 * onna_hip.h -- C ABI of the MI355X (gfx950) renderer for the features that exist only in the
 * reference's Common Lisp tracer (hgiddens/ray-tracing-in-one-weekend, onna/): quadrilaterals and
 * boxes, translate / rotate-y instances, constant-density media, image and gradient-noise textures,
 * emissive materials, and the importance-sampling integrator (scattering weighted by mixture PDFs
 * over a list of lights).  Citations are file:line under onna/.
 *
 * The Lisp tracer has no FFI either: its hot path is the generic functions hit-test
 * (scene.lisp:22), scatter / emitted / scattering-pdf (material.lisp:13-45), texture-value
 * (texture.lisp:3), pdf-value / pdf-generate (pdf.lisp:3-7) and the loops of render and ray-colour
 * (camera.lisp:94-181).  This header is the boundary beneath them: the host builds the object graph
 * (bounding boxes, bvh, derived quad constants) and hands it over as flat tables.
 *
 * Conventions as in rtiow_hip.h: plain structs and pointers, an opaque scene handle, 0 on success
 * or an RT_E_* code with the text in on_last_error() (thread-local), no exceptions, `device` = -1
 * for the calling thread's current HIP device.  All arithmetic is IEEE double precision in the
 * operation order of the Lisp expressions, without fused multiply-add.
 */
#ifndef ONNA_HIP_H
#define ONNA_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "rtiow_hip.h" /* rt_ray, rt_stats, RT_E_* */

#ifdef __cplusplus
extern "C" {
#endif

/* scene.lisp: sphere :129, quad :215, sequences :92, bvh-node :47, translate :319, rotate-y :340, constant-medium :412 */
enum { ON_OBJ_SPHERE = 0, ON_OBJ_QUAD = 1, ON_OBJ_LIST = 2, ON_OBJ_BVH = 3, ON_OBJ_TRANSLATE = 4, ON_OBJ_ROTATE_Y = 5, ON_OBJ_MEDIUM = 6 };
/* material.lisp: lambertian :49, metal :68, dielectric :82, diffuse-light :110, isotropic :128 */
enum { ON_MAT_LAMBERTIAN = 0, ON_MAT_METAL = 1, ON_MAT_DIELECTRIC = 2, ON_MAT_DIFFUSE_LIGHT = 3, ON_MAT_ISOTROPIC = 4 };
/* texture.lisp: a colour :8, chequer :14, image-texture :36, perlin :50 */
enum { ON_TEX_SOLID = 0, ON_TEX_CHEQUER = 1, ON_TEX_IMAGE = 2, ON_TEX_PERLIN = 3 };

/* One object of the graph.  v[] by kind:
 *   sphere     v[0..2] centre at time 0, v[3..5] centre's motion per unit time (zero: static), v[6] radius
 *              (scene.lisp:129-148: the centre is a ray, from -> to)
 *   quad       v[0..2] q, v[3..5] u, v[6..8] v, and what make-quad derives (scene.lisp:219-228):
 *              v[9..11] w = n / (n.n), v[12..14] normal = unit(n), v[15] d = normal.q, v[16] area = |n|, n = u x v
 *   list       a = first entry in `children`, b = count (a Lisp sequence of objects)
 *   bvh        a = left, b = right (scene.lisp:47-75; one object twice for a span of one)
 *   translate  a = object, v[0..2] offset
 *   rotate-y   a = object, v[0] sin theta, v[1] cos theta
 *   medium     a = boundary object, material = its isotropic phase function, v[0] = -1 / density
 * material: index into `materials`, or -1 (the stand-ins that make up a list of lights have none).
 * box: the object's bounding-box (min x, max x, min y, max y, min z, max z); only bvh nodes test it. */
typedef struct on_object_desc {
    uint32_t kind;
    int32_t material;
    int32_t a, b;
    double v[18];
    double box[6];
} on_object_desc;

/* chequer: even / odd index textures, scale holds 1/scale (texture.lisp:14-30); image: `table`
 * indexes `images`; perlin: `table` indexes `perlins`, scale as texture.lisp:69 */
typedef struct on_texture_desc { uint32_t kind; int32_t even, odd, table; double colour[3]; double scale; } on_texture_desc;

/* an image as image.lisp:10-28 reads it: 8-bit RGB rows, top row first; a channel's value is (x / 256)^2 */
typedef struct on_image_desc { uint32_t width, height; uint64_t offset; /* of its first byte in `pixels` */ } on_image_desc;

/* texture.lisp:50-69: 256 random unit vectors and three permutations of 0..255 */
typedef struct on_perlin_desc { double vec[256][3]; int32_t perm[3][256]; } on_perlin_desc;

typedef struct on_material_desc { uint32_t kind; int32_t texture; double albedo[3]; double fuzz; double refraction_index; } on_material_desc;

typedef struct on_scene_desc {
    const on_object_desc *objects;     uint32_t n_objects;
    const int32_t *children;           uint32_t n_children;   /* the members of the lists, list after list */
    const on_material_desc *materials; uint32_t n_materials;
    const on_texture_desc *textures;   uint32_t n_textures;
    const on_image_desc *images;       uint32_t n_images;
    const uint8_t *pixels;             uint64_t n_pixel_bytes;
    const on_perlin_desc *perlins;     uint32_t n_perlins;
    int32_t world;    /* the object render traces */
    int32_t lights;   /* a sphere, a quad or a list of those (camera.lisp:165 `lights`), or -1: none */
} on_scene_desc;

/* what make-camera computes (camera.lisp:19-90) */
typedef struct on_camera {
    double centre[3], pixel_delta_u[3], pixel_delta_v[3], pixel00[3], defocus_u[3], defocus_v[3], background[3];
    double recip_sqrt_spp;              /* 1 / sqrt_spp, correctly rounded */
    uint32_t image_width, image_height;
    uint32_t sqrt_spp;                  /* samples per pixel = sqrt_spp^2, stratified (camera.lisp:34,169-174) */
    uint32_t max_depth;
} on_camera;

/* Sample s of a pixel is the stratum sj = s / sqrt_spp, si = s % sqrt_spp, in the order render
 * collects them (camera.lisp:172-174).  A GPU renders [sample_begin, sample_end) of every pixel. */
typedef struct on_render_params { uint32_t sample_begin, sample_end; uint64_t seed; } on_render_params;

/* hit-record (scene.lisp:5-20); `object` is the sphere, quad or medium that was hit */
typedef struct on_hit {
    int32_t hit, object, material, front_face;
    double t, p[3], normal[3], u, v;
} on_hit;

typedef struct on_scene on_scene;

const char *on_last_error(void);

int on_scene_create(const on_scene_desc *desc, int device, on_scene **out);
void on_scene_destroy(on_scene *scene);

/* hit-test of `object` over a batch of rays with the interval (t_min, t_max) (scene.lisp:22; methods
 * :77, :92, :150, :240, :328, :379, :423).  A constant-medium draws one uniform per visit that finds
 * the ray inside its boundary (scene.lisp:443): ray i takes them in order from uniforms[i*k .. i*k+k)
 * (values in [0, 1)); k = 0 and uniforms = NULL when no medium lies under `object`.  Host buffers. */
int on_hit_batch(on_scene *scene, int32_t object, const rt_ray *rays, size_t n, double t_min, double t_max,
                 const double *uniforms, uint32_t k, on_hit *out);

/* texture-value (texture.lisp:3; methods :8, :22, :39, :106): uvp[n][5] = u, v, p; host buffers */
int on_texture_batch(on_scene *scene, int32_t texture, const double *uvp, size_t n, double *rgb);

/* object-pdf-value of `object` -- a sphere, a quad or a list of them -- for rays origin -> direction
 * (scene.lisp:34; methods :106, :184, :267); od[n][6]; host buffers */
int on_pdf_value_batch(on_scene *scene, int32_t object, const double *od, size_t n, double *out);

/* render (camera.lisp:165-181) with ray-colour (camera.lisp:94-134): ADDS the radiance of samples
 * [sample_begin, sample_end) of every pixel to accum_dev (device memory on the scene's GPU,
 * image_height * image_width * 3 doubles, row-major, top row first), asynchronously on `stream`
 * (a hipStream_t, or NULL).  Every sample draws from its own random stream keyed by (seed, pixel,
 * sample) and leaves its colour in a slot of its own; a pixel's slots are then added in sample
 * order and that sum is added to accum_dev -- so the frame does not depend on how the work was
 * scheduled, and disjoint sample ranges add up to the whole frame's sums (a rounding per range
 * apart).  The slots live in a buffer of the scene's (image_height * image_width * 3 doubles per
 * sample, at most 4 GiB; allocated or grown by the first call that needs it, which may synchronise
 * the device); calls with more samples than it holds go through in several passes, each pass's
 * sum added to accum_dev in turn.  Renders of ONE scene must not overlap
 * in time.  stats may be NULL; when given, the call waits for the frame and fills it. */
int on_render(on_scene *scene, const on_camera *camera, const on_render_params *params, double *accum_dev,
              void *stream, rt_stats *stats);

/* the same into a host buffer, zeroed first */
int on_render_host(on_scene *scene, const on_camera *camera, const on_render_params *params, double *accum_host,
                   rt_stats *stats);

/* blend-colours, gamma-2 and colour-8bit (colour.lisp:12-18,20-37; image.lisp:30-40):
 * rgb8 = floor(256 * clamp(sqrt(sum / spp), 0, 0.999)).  Device buffers on `device`. */
int on_resolve(int device, const double *accum_dev, size_t npix, uint32_t spp, uint8_t *rgb8_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif
