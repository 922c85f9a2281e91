/*
 * onna_oracle.h -- CPU restatement of the reference's Common Lisp tracer (onna/), see onna_oracle.c.
 *
 * THIS IS TEST INFRASTRUCTURE (the checker for libonna_hip.so); PARITY UNPINNED: no Lisp is in this
 * image and the reference holds no fixtures for this tracer, so the restatement is pinned by the text
 * of the Lisp source alone.  It reads the flat scene tables of include/onna_hip.h (the boundary's data
 * layout, shared so that one description can be handed to both sides).
 */
#ifndef ONNA_ORACLE_H
#define ONNA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/onna_hip.h"
#include "rt_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* a source of (random 1d0): a per-sample stream, or a tape of given uniforms when `tape` is set */
typedef struct { ostream st; const double *tape; int len, pos, over; } odraw;
double onc_uniform(odraw *g);

typedef struct { uint64_t paths, rays, box_tests, sphere_tests, quad_tests; } onc_counters;

void onc_hit_batch(const on_scene_desc *s, int32_t object, const double *rays /*[n][7]*/, size_t n, double t_min, double t_max,
                   const double *uniforms, uint32_t k, on_hit *out);
void onc_texture_batch(const on_scene_desc *s, int32_t texture, const double *uvp /*[n][5]*/, size_t n, double *rgb);
void onc_pdf_value_batch(const on_scene_desc *s, int32_t object, const double *od /*[n][6]*/, size_t n, double *out);
void onc_random_direction_batch(const on_scene_desc *s, int32_t object, const double *origins /*[n][3]*/, const double *tape, uint32_t k,
                                size_t n, double *out /*[n][3]*/);
void onc_render(const on_scene_desc *s, const on_camera *c, const on_render_params *p, double *out_sum, onc_counters *cnt);
void onc_camera_ray(const on_camera *c, int i, int j, int si, int sj, uint64_t seed, uint32_t pixel, uint32_t sample, double out[7]);
void onc_resolve(const double *sums, size_t n_values, uint32_t spp, uint8_t *rgb8);

#ifdef __cplusplus
}
#endif
#endif
