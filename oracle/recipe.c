/*
 * recipe.c -- reader for the text scene recipes used by the oracle and by the
 * reference capture tool.  TEST INFRASTRUCTURE ONLY (see rt_oracle.h).
 *
 * A recipe lists what the reference's scene builders (main.cpp:52-99) pass to the
 * public constructors, one object per line, in construction order:
 *
 *   rtiow-recipe 1
 *   texture constant R G B | texture chequer EVEN ODD | texture perlin
 *   material lambertian TEX | material metal R G B FUZZ | material dielectric RI
 *   sphere MAT CX CY CZ R
 *   moving_sphere MAT C0X C0Y C0Z C1X C1Y C1Z T0 T1 R
 *   bvh_begin SEED T0 T1   ... primitives ...   bvh_end
 *   list_begin   ... primitives ...   list_end            (inside a bvh group only)
 *
 * Textures, materials and primitives are numbered from 0 in order of appearance.
 * The top level is an object_list; a bvh group is one bvh_node built over its
 * objects with a fresh std::mt19937(SEED).  A list group inside a bvh group is one
 * object_list handed to that bvh_node as a single object (object_list.cpp:28-48 gives
 * it a bounding box for exactly that).
 */
#include "rt_oracle.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void fail(char *err, size_t errlen, const char *fmt, ...) {
    if (!err || !errlen) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, errlen, fmt, ap);
    va_end(ap);
}

#define GROW(arr, n, type)                                                              \
    do {                                                                                \
        if (((n) & ((n) - 1)) == 0)                                                     \
            (arr) = (type *)realloc((arr), (size_t)((n) ? (n) * 2 : 1) * sizeof(type)); \
    } while (0)

void orc_scene_free(oscene *s) {
    if (!s) return;
    free(s->tex); free(s->mat); free(s->prim); free(s->list); free(s->node); free(s->item);
    free(s);
}

oscene *orc_scene_load(const char *path, char *err, size_t errlen) {
    FILE *f = fopen(path, "r");
    if (!f) { fail(err, errlen, "cannot open %s", path); return NULL; }
    oscene *s = (oscene *)calloc(1, sizeof(oscene));
    char line[1024];
    int lineno = 0, in_bvh = 0, in_list = 0, header = 0;
    oitem group;
    memset(&group, 0, sizeof group);
    while (fgets(line, sizeof line, f)) {
        lineno++;
        char kw[64], sub[64];
        if (sscanf(line, "%63s", kw) != 1 || kw[0] == '#') continue;
        if (!header) {
            int ver = 0;
            if (strcmp(kw, "rtiow-recipe") != 0 || sscanf(line, "%*s %d", &ver) != 1 || ver != 1) {
                fail(err, errlen, "%s:%d: not a version-1 recipe", path, lineno);
                goto bad;
            }
            header = 1;
            continue;
        }
        if (strcmp(kw, "texture") == 0) {
            otexture t;
            memset(&t, 0, sizeof t);
            if (sscanf(line, "%*s %63s", sub) != 1) goto syntax;
            if (strcmp(sub, "constant") == 0) {
                t.kind = ORC_TEX_CONSTANT;
                if (sscanf(line, "%*s %*s %lf %lf %lf", &t.c[0], &t.c[1], &t.c[2]) != 3) goto syntax;
            } else if (strcmp(sub, "chequer") == 0) {
                t.kind = ORC_TEX_CHEQUER;
                if (sscanf(line, "%*s %*s %d %d", &t.even, &t.odd) != 2) goto syntax;
                if (t.even < 0 || t.even >= s->n_tex || t.odd < 0 || t.odd >= s->n_tex) goto syntax;
            } else if (strcmp(sub, "perlin") == 0) {
                t.kind = ORC_TEX_PERLIN;
            } else goto syntax;
            GROW(s->tex, s->n_tex, otexture);
            s->tex[s->n_tex++] = t;
        } else if (strcmp(kw, "material") == 0) {
            omaterial m;
            memset(&m, 0, sizeof m);
            if (sscanf(line, "%*s %63s", sub) != 1) goto syntax;
            if (strcmp(sub, "lambertian") == 0) {
                m.kind = ORC_MAT_LAMBERTIAN;
                if (sscanf(line, "%*s %*s %d", &m.tex) != 1 || m.tex < 0 || m.tex >= s->n_tex) goto syntax;
            } else if (strcmp(sub, "metal") == 0) {
                m.kind = ORC_MAT_METAL;
                if (sscanf(line, "%*s %*s %lf %lf %lf %lf", &m.albedo[0], &m.albedo[1], &m.albedo[2], &m.fuzz) != 4)
                    goto syntax;
            } else if (strcmp(sub, "dielectric") == 0) {
                m.kind = ORC_MAT_DIELECTRIC;
                if (sscanf(line, "%*s %*s %lf", &m.ri) != 1) goto syntax;
            } else goto syntax;
            GROW(s->mat, s->n_mat, omaterial);
            s->mat[s->n_mat++] = m;
        } else if (strcmp(kw, "sphere") == 0 || strcmp(kw, "moving_sphere") == 0) {
            oprim p;
            memset(&p, 0, sizeof p);
            if (kw[0] == 's') {
                p.kind = ORC_PRIM_SPHERE;
                if (sscanf(line, "%*s %d %lf %lf %lf %lf", &p.mat, &p.c0.x, &p.c0.y, &p.c0.z, &p.r) != 5) goto syntax;
                p.c1 = p.c0; p.t0 = 0; p.t1 = 1;
            } else {
                p.kind = ORC_PRIM_MOVING_SPHERE;
                if (sscanf(line, "%*s %d %lf %lf %lf %lf %lf %lf %lf %lf %lf", &p.mat, &p.c0.x, &p.c0.y, &p.c0.z,
                           &p.c1.x, &p.c1.y, &p.c1.z, &p.t0, &p.t1, &p.r) != 10)
                    goto syntax;
            }
            if (p.mat < 0 || p.mat >= s->n_mat) goto syntax;
            p.list = in_list ? s->n_list - 1 : -1;
            if (in_list) s->list[s->n_list - 1].n++;
            GROW(s->prim, s->n_prim, oprim);
            s->prim[s->n_prim++] = p;
            if (in_bvh) {
                group.n_prims++;
            } else {
                oitem it;
                memset(&it, 0, sizeof it);
                it.is_bvh = 0; it.index = s->n_prim - 1; it.first_prim = it.index; it.n_prims = 1;
                GROW(s->item, s->n_item, oitem);
                s->item[s->n_item++] = it;
            }
        } else if (strcmp(kw, "bvh_begin") == 0) {
            unsigned long seed;
            if (in_bvh) goto syntax;
            memset(&group, 0, sizeof group);
            if (sscanf(line, "%*s %lu %lf %lf", &seed, &group.t0, &group.t1) != 3) goto syntax;
            group.is_bvh = 1; group.seed = (uint32_t)seed; group.first_prim = s->n_prim;
            in_bvh = 1;
        } else if (strcmp(kw, "list_begin") == 0) {
            if (!in_bvh || in_list) goto syntax;
            olist l = {s->n_prim, 0};
            GROW(s->list, s->n_list, olist);
            s->list[s->n_list++] = l;
            in_list = 1;
        } else if (strcmp(kw, "list_end") == 0) {
            if (!in_list || s->list[s->n_list - 1].n == 0) goto syntax;
            in_list = 0;
        } else if (strcmp(kw, "bvh_end") == 0) {
            if (!in_bvh || in_list || group.n_prims == 0) goto syntax;
            in_bvh = 0;
            GROW(s->item, s->n_item, oitem);
            s->item[s->n_item++] = group;
        } else goto syntax;
        continue;
    syntax:
        fail(err, errlen, "%s:%d: syntax error: %s", path, lineno, line);
        goto bad;
    }
    if (!header || in_bvh) { fail(err, errlen, "%s: truncated recipe", path); goto bad; }
    fclose(f);
    /* build each bvh group as the reference constructor does */
    for (int i = 0; i < s->n_item; ++i) {
        oitem *it = &s->item[i];
        if (!it->is_bvh) continue;
        /* the group's objects: its primitives, a list counting as one */
        int *ids = (int *)malloc((size_t)it->n_prims * sizeof(int));
        int n = 0;
        for (int k = 0; k < it->n_prims; ++k) {
            const oprim *p = &s->prim[it->first_prim + k];
            if (p->list < 0) ids[n++] = it->first_prim + k;
            else if (s->list[p->list].first == it->first_prim + k) ids[n++] = ORC_LIST_BASE + p->list;
        }
        omt19937 mt;
        orc_mt_seed(&mt, it->seed);
        it->index = orc_bvh_build(s, ids, n, &mt, it->t0, it->t1);
        free(ids);
    }
    return s;
bad:
    fclose(f);
    orc_scene_free(s);
    return NULL;
}
