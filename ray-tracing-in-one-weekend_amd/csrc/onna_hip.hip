// onna_hip.hip -- kernels and C ABI (include/onna_hip.h) of the MI355X renderer for the features of the
// reference's Common Lisp tracer (onna/).  Host side: validates a flat scene description, uploads it and
// launches.  There is no CPU implementation of any entry point: without a HIP device every call fails.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "onna_device.hpp"
#include "onna_hip.h"

using namespace ond;

namespace {

constexpr int kBlock = 256;
// k_render's runs of samples: each takes 35 % of what is left of its window, at least 4 samples and at most `longest`
constexpr uint32_t kShortestRun = 4, kRunShare = 35;
__host__ __device__ inline uint32_t run_length(uint32_t left, uint32_t longest) {
    uint32_t len = (uint32_t)(((unsigned long long)left * kRunShare + 99) / 100);
    if (len < kShortestRun) len = kShortestRun;
    if (len > longest) len = longest;
    return len < left ? len : left;
}
constexpr size_t kSlotBudgetBytes = (size_t)4 << 30;   // of a scene's buffer of per-sample colours (see k_render)

__device__ __forceinline__ Ray load_ray(const rt_ray *r) {
    Ray o;
    o.o = v3(r->origin[0], r->origin[1], r->origin[2]);
    o.d = v3(r->direction[0], r->direction[1], r->direction[2]);
    o.time = r->time;
    return o;
}

// hit-test over a batch, one ray per lane
__global__ __launch_bounds__(kBlock) void k_hit_batch(DScene sc, int32_t object, const rt_ray *rays, size_t n, double t_min, double t_max,
                                                     const double *uniforms, uint32_t k, on_hit *out) {
    __shared__ int32_t stacks[kStackCap * kBlock];
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const LaneStack st = {stacks + threadIdx.x, kBlock};
    const Ray r = load_ray(rays + i);
    TapeDraw g = {uniforms ? uniforms + i * (size_t)k : nullptr, uniforms ? (int)k : 0, 0, false};
    const Chain none = {0, 0, 0, 0, 0};
    Found f;
    on_hit h;
    memset(&h, 0, sizeof h);
    h.object = -1; h.material = -1;
    if (walk<true>(sc, object, r, none, t_min, t_max, st, 0, g, f)) {
        const Hit rec = hit_record(sc, r, f);
        h.hit = 1; h.object = rec.object; h.material = rec.material; h.front_face = rec.front ? 1 : 0;
        h.t = rec.t; h.u = rec.u; h.v = rec.v;
        h.p[0] = rec.p.x; h.p[1] = rec.p.y; h.p[2] = rec.p.z;
        h.normal[0] = rec.normal.x; h.normal[1] = rec.normal.y; h.normal[2] = rec.normal.z;
    }
    out[i] = h;
}

__global__ __launch_bounds__(kBlock) void k_texture_batch(DScene sc, int32_t texture, const double *uvp, size_t n, double *rgb) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const V c = texture_value(sc, texture, uvp[5 * i], uvp[5 * i + 1], ld(uvp + 5 * i + 2));
    rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
}

__global__ __launch_bounds__(kBlock) void k_pdf_value_batch(DScene sc, int32_t object, const double *od, size_t n, double *out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    out[i] = object_pdf_value(sc, object, ld(od + 6 * i), ld(od + 6 * i + 3));
}

constexpr uint32_t kMaxRuns = 30;
// render (camera.lisp:165-181).  A wave (a block) owns 8x8 pixels and a run of samples [s0, s1) of them, 64 (s1 - s0) paths in
// all, numbered pixel-fastest (path k is pixel k % 64, sample s0 + k / 64).  One loop over ray segments: a lane whose path has
// ended takes the next unstarted path of the wave at once -- one ballot and a count of the lanes below -- whatever the other
// lanes are in the middle of, so no lane waits for a neighbour whose pixel is dearer than its own; what is left to wait for
// at the wave's end is one path.  A path's colour goes to its own slot, slots[sample - w_begin][pixel][3], and k_combine adds
// a pixel's slots in sample order (blend-colours sums a pixel's samples in the order they were collected: colour.lisp:20-29):
// which lane traced what, and when, is not in the result.
// (161 vector registers: three waves per SIMD.  The walk waits on its own loads, so occupancy is what its speed hangs on:
// changes that push the count past 168 -- two waves -- have cost 20-30 % whatever else they did.)
// run r of a window is its samples [start[r], start[r + 1]); n = 0: the runs follow run_length (the default; the table is for tests)
struct SampleRuns { uint32_t n; uint32_t start[kMaxRuns + 1]; };
__global__ __launch_bounds__(64) void k_render(DScene sc, on_camera cam, uint32_t w_begin, uint32_t w_len, uint32_t longest, SampleRuns runs, uint64_t seed,
                                                      double *slots, unsigned long long *counters) {
    __shared__ int32_t stacks[kStackCap * 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t tiles_x = (cam.image_width + 7u) / 8u, tiles = tiles_x * ((cam.image_height + 7u) / 8u);
    const uint32_t tile = blockIdx.x % tiles, part = blockIdx.x / tiles;
    const uint32_t x0 = (tile % tiles_x) * 8u, y0 = (tile / tiles_x) * 8u;
    uint32_t s0, s1;
    if (runs.n) {
        s0 = w_begin + runs.start[part]; s1 = w_begin + runs.start[part + 1];
    } else {   // (the same for the whole wave: a few scalar operations per run before this one)
        uint32_t start = 0, left = w_len, len;
        for (uint32_t i = 0;; ++i) {
            len = run_length(left, longest);
            if (i == part) break;
            start += len; left -= len;
        }
        s0 = w_begin + start; s1 = s0 + len;
    }
    const uint32_t n_paths = 64u * (s1 - s0);
    const size_t n_values = (size_t)cam.image_width * cam.image_height * 3;
    const LaneStack st = {stacks + lane, 64};
    unsigned long long segments = 0;
    uint32_t next = 0;   // the same in every lane
    uint32_t pixel = 0, sample = 0;
    bool running = false, done = false;
    StreamDraw g;
    Path path;
    for (;;) {
        for (;;) {   // free lanes take paths; a path of a pixel beyond the image's edge is dropped and the lane asks again
            const bool want = !running && !done;
            const unsigned long long need = __ballot(want);
            if (need == 0) break;
            if (want) {
                const uint32_t k = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                if (k >= n_paths) {
                    done = true;
                } else {
                    const uint32_t x = x0 + (k & 7u), y = y0 + ((k >> 3) & 7u);
                    if (x < cam.image_width && y < cam.image_height) {
                        pixel = y * cam.image_width + x;
                        sample = s0 + (k >> 6);
                        g.g.init(seed, pixel, sample);
                        path_begin(path, get_ray(cam, (int)x, (int)y, (int)(sample % cam.sqrt_spp), (int)(sample / cam.sqrt_spp), g), cam.max_depth);
                        running = true;
                    }
                }
            }
            next += (uint32_t)__popcll(need);
        }
        if (__ballot(running) == 0) break;
        if (running) {
            if (path_segment(sc, path, cam.background, st, g)) {
                ++segments;
            } else {
                double *o = slots + (size_t)(sample - w_begin) * n_values + 3 * (size_t)pixel;
                o[0] = path.out.x; o[1] = path.out.y; o[2] = path.out.z;
                running = false;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) segments += __shfl_xor(segments, off);
    if (lane == 0) atomicAdd(counters, segments);
}

// a pixel's slots added in sample order, then to the image
__global__ __launch_bounds__(kBlock) void k_combine(const double *slots, size_t n_values, uint32_t samples, double *accum) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_values) return;
    double sum = 0;
    for (uint32_t s = 0; s < samples; ++s) sum += slots[(size_t)s * n_values + i];
    accum[i] += sum;
}

// blend-colours, gamma-2, colour-8bit (colour.lisp:12-37; image.lisp:30-40)
__global__ __launch_bounds__(kBlock) void k_resolve(const double *accum, size_t n_values, uint32_t spp, uint8_t *rgb8) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_values) return;
    double x = sqrt(accum[i] / (double)spp);
    x = x < 0 ? 0 : (x > (double)0.999f ? (double)0.999f : x);
    if (!(x == x)) x = 0;
    rgb8[i] = (uint8_t)floor(256 * x);
}

// ------------------------------------------------------------------ host side ----
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(RT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class T> struct dev_buf {
    T *p = nullptr;
    ~dev_buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)); }
    hipError_t upload(const T *h, size_t n) {
        hipError_t e = alloc(n);
        if (e != hipSuccess || n == 0) return e;
        return hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice);
    }
};

unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

int use_device(int device, int *chosen) {
    if (device < 0) {
        HIP_TRY(hipGetDevice(&device));
    } else {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RT_E_DEVICE, "no HIP device available");
        if (device >= ndev) return fail(RT_E_INVALID, "device %d out of range (have %d)", device, ndev);
        HIP_TRY(hipSetDevice(device));
    }
    if (chosen) *chosen = device;
    return RT_OK;
}

// What the walk needs of a description, worked out once: per object the stack entries a visit uses
// beyond the one it was popped from, whether a medium lies under it, and the deepest nest of instances.
struct shape_check {
    const on_scene_desc *d;
    std::vector<int> need, chain;      // -1 = not yet known, -2 = being worked out (a cycle)
    std::vector<char> media, plain;    // a medium below / only spheres, quads and lists of them below
    const char *error = nullptr;

    bool visit(int32_t id, int depth) {
        if (error) return false;
        if (id < 0 || (uint32_t)id >= d->n_objects) { error = "object index out of range"; return false; }
        if (need[id] == -2 || depth > 256) { error = "the object graph has a cycle (or is deeper than 256 levels)"; return false; }
        if (need[id] >= 0) return true;
        need[id] = -2;
        const on_object_desc &o = d->objects[id];
        int n = 0, c = 0;
        bool m = false, p = true;
        auto child = [&](int32_t k, int extra) {
            if (!visit(k, depth + 1)) return false;
            n = std::max(n, extra + need[k]); c = std::max(c, chain[k]); m = m || media[k]; p = p && plain[k];
            return true;
        };
        switch (o.kind) {
        case ON_OBJ_SPHERE: case ON_OBJ_QUAD:
            if (o.material < -1 || (o.material >= 0 && (uint32_t)o.material >= d->n_materials)) { error = "material index out of range"; return false; }
            break;
        case ON_OBJ_LIST:
            if (o.a < 0 || o.b < 0 || (uint64_t)o.a + (uint64_t)o.b > d->n_children) { error = "list members out of range"; return false; }
            n = 1;
            for (int i = 0; i < o.b; ++i)
                if (!child(d->children[o.a + i], 1)) return false;
            break;
        case ON_OBJ_BVH:
            n = 2; p = false;
            if (!child(o.a, 1) || !child(o.b, 0)) return false;
            break;
        case ON_OBJ_TRANSLATE: case ON_OBJ_ROTATE_Y:
            n = 2; p = false;
            if (!child(o.a, 1)) return false;
            c += 1;
            break;
        case ON_OBJ_MEDIUM:
            p = false;
            if (o.material < 0 || (uint32_t)o.material >= d->n_materials) { error = "a medium needs its phase function's material"; return false; }
            if (!child(o.a, 1)) return false;
            if (media[o.a]) { error = "a constant-medium inside the boundary of another is not supported"; return false; }
            m = true;
            break;
        default: error = "unknown object kind"; return false;
        }
        need[id] = n; chain[id] = c; media[id] = m; plain[id] = p;
        return true;
    }
};

}  // namespace

struct on_scene {
    int device = 0;
    DScene d{};
    dev_buf<DObj> objects;
    dev_buf<int32_t> children;
    dev_buf<on_material_desc> materials;
    dev_buf<on_texture_desc> textures;
    dev_buf<on_image_desc> images;
    dev_buf<uint8_t> pixels;
    dev_buf<on_perlin_desc> perlins;
    dev_buf<unsigned long long> counter;
    dev_buf<double> slots;              // the colours of a window's samples, [sample][pixel][3] (see k_render)
    size_t slot_capacity = 0;
    int n_cus = 256;
    uint32_t n_objects = 0, n_textures = 0;
    std::vector<char> media_below, plain, walkable;   // per object
    bool renderable = false;
    const char *why_not = "";
};

extern "C" {

const char *on_last_error(void) { return g_err; }

static int scene_create(const on_scene_desc *desc, int device, on_scene **out) {
    if (!desc) return fail(RT_E_INVALID, "null scene description");
    if ((desc->n_objects && !desc->objects) || (desc->n_children && !desc->children) || (desc->n_materials && !desc->materials) ||
        (desc->n_textures && !desc->textures) || (desc->n_images && !desc->images) || (desc->n_pixel_bytes && !desc->pixels) ||
        (desc->n_perlins && !desc->perlins))
        return fail(RT_E_INVALID, "scene description has a null table with a non-zero count");
    if (device < -1) return fail(RT_E_INVALID, "device %d: want an index, or -1 for the current device", device);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const on_texture_desc &t = desc->textures[i];
        if (t.kind > ON_TEX_PERLIN) return fail(RT_E_INVALID, "texture %u: unknown kind %u", i, t.kind);
        if (t.kind == ON_TEX_CHEQUER && (t.even < 0 || t.odd < 0 || (uint32_t)t.even >= i || (uint32_t)t.odd >= i))
            return fail(RT_E_INVALID, "texture %u: a chequer's textures must come before it in the table", i);
        if (t.kind == ON_TEX_IMAGE) {
            if (t.table < 0 || (uint32_t)t.table >= desc->n_images) return fail(RT_E_INVALID, "texture %u: image %d out of range", i, t.table);
            const on_image_desc &im = desc->images[t.table];
            if (im.width == 0 || im.height == 0 || im.offset > desc->n_pixel_bytes || (uint64_t)im.width * im.height * 3 > desc->n_pixel_bytes - im.offset)
                return fail(RT_E_INVALID, "texture %u: image %d lies outside the pixel table", i, t.table);
        }
        if (t.kind == ON_TEX_PERLIN) {
            if (t.table < 0 || (uint32_t)t.table >= desc->n_perlins) return fail(RT_E_INVALID, "texture %u: perlin tables %d out of range", i, t.table);
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < 256; ++k)
                    if (desc->perlins[t.table].perm[a][k] < 0 || desc->perlins[t.table].perm[a][k] > 255) return fail(RT_E_INVALID, "texture %u: a permutation entry outside 0..255", i);
        }
    }
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const on_material_desc &m = desc->materials[i];
        if (m.kind > ON_MAT_ISOTROPIC) return fail(RT_E_INVALID, "material %u: unknown kind %u", i, m.kind);
        const bool textured = m.kind == ON_MAT_LAMBERTIAN || m.kind == ON_MAT_DIFFUSE_LIGHT || m.kind == ON_MAT_ISOTROPIC;
        if (textured && (m.texture < 0 || (uint32_t)m.texture >= desc->n_textures)) return fail(RT_E_INVALID, "material %u: texture %d out of range", i, m.texture);
    }
    shape_check sh;
    sh.d = desc;
    sh.need.assign(desc->n_objects, -1); sh.chain.assign(desc->n_objects, 0); sh.media.assign(desc->n_objects, 0); sh.plain.assign(desc->n_objects, 0);
    for (uint32_t i = 0; i < desc->n_objects; ++i)
        if (!sh.visit((int32_t)i, 0)) return fail(RT_E_INVALID, "object %u: %s", i, sh.error);
    if (desc->world < 0 || (uint32_t)desc->world >= desc->n_objects) return fail(RT_E_INVALID, "world %d is not an object", desc->world);
    if (desc->lights < -1 || (desc->lights >= 0 && (uint32_t)desc->lights >= desc->n_objects)) return fail(RT_E_INVALID, "lights %d is not an object", desc->lights);
    if (desc->lights >= 0) {   // object-pdf-value and random-direction exist for spheres, quads and sequences (scene.lisp:106-115, 184-211, 267-290)
        const on_object_desc &l = desc->objects[desc->lights];
        bool ok = l.kind == ON_OBJ_SPHERE || l.kind == ON_OBJ_QUAD || (l.kind == ON_OBJ_LIST && l.b > 0);
        if (l.kind == ON_OBJ_LIST)
            for (int i = 0; i < l.b; ++i) { const uint32_t k = desc->objects[desc->children[l.a + i]].kind; ok = ok && (k == ON_OBJ_SPHERE || k == ON_OBJ_QUAD); }
        if (!ok) return fail(RT_E_UNSUPPORTED, "lights must be a sphere, a quad or a non-empty list of spheres and quads");
    }

    std::unique_ptr<on_scene> owner(new on_scene);
    on_scene *s = owner.get();
    s->n_objects = desc->n_objects; s->n_textures = desc->n_textures;
    s->media_below.assign(sh.media.begin(), sh.media.end());
    s->plain.assign(sh.plain.begin(), sh.plain.end());
    s->walkable.assign(desc->n_objects, 0);
    // a walk from an object needs the entry it starts with, what its visit takes, and -- a medium's boundary
    // being walked from where the medium was popped -- the same again for boundaries
    for (uint32_t i = 0; i < desc->n_objects; ++i) {
        int need = 1 + sh.need[i];
        if (sh.media[i]) need += 1 + sh.need[i];   // (generous: a boundary's walk starts above the medium's place)
        s->walkable[i] = need <= kStackCap && sh.chain[i] <= kMaxChain;
    }
    // render follows every hit to its material
    s->renderable = s->walkable[desc->world] != 0;
    if (!s->renderable) s->why_not = "the world is nested deeper than the walk's stack (32 entries) or its instance chain (4) allows";
    for (uint32_t i = 0; i < desc->n_objects && s->renderable; ++i) {
        const on_object_desc &o = desc->objects[i];
        if ((o.kind == ON_OBJ_SPHERE || o.kind == ON_OBJ_QUAD) && o.material < 0) {
            // stand-ins of the lights list are fine as long as the world cannot reach them; cheap test: they are only allowed when they are not under the world
            std::vector<char> seen(desc->n_objects, 0);
            std::vector<int32_t> todo{desc->world};
            bool reached = false;
            while (!todo.empty() && !reached) {
                const int32_t k = todo.back(); todo.pop_back();
                if (seen[k]) continue;
                seen[k] = 1;
                if ((uint32_t)k == i) { reached = true; break; }
                const on_object_desc &q = desc->objects[k];
                if (q.kind == ON_OBJ_LIST) for (int j = 0; j < q.b; ++j) todo.push_back(desc->children[q.a + j]);
                else if (q.kind == ON_OBJ_BVH) { todo.push_back(q.a); todo.push_back(q.b); }
                else if (q.kind >= ON_OBJ_TRANSLATE) todo.push_back(q.a);
            }
            if (reached) { s->renderable = false; s->why_not = "an object of the world has no material"; }
        }
    }

    // which materials' textures read a hit's (u, v): those with an image somewhere in them (a chequer's come before it)
    std::vector<char> texture_reads_uv(desc->n_textures, 0), material_reads_uv(desc->n_materials, 0);
    for (uint32_t i = 0; i < desc->n_textures; ++i) {
        const on_texture_desc &t = desc->textures[i];
        texture_reads_uv[i] = t.kind == ON_TEX_IMAGE || (t.kind == ON_TEX_CHEQUER && (texture_reads_uv[t.even] || texture_reads_uv[t.odd]));
    }
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const on_material_desc &m = desc->materials[i];
        const bool textured = m.kind == ON_MAT_LAMBERTIAN || m.kind == ON_MAT_DIFFUSE_LIGHT || m.kind == ON_MAT_ISOTROPIC;
        material_reads_uv[i] = textured && texture_reads_uv[m.texture];
    }
    // device tables; every list's members are followed by a terminator
    std::vector<DObj> objects(desc->n_objects);
    for (uint32_t i = 0; i < desc->n_objects; ++i) {
        const on_object_desc &in = desc->objects[i];
        DObj &o = objects[i];
        memset(&o, 0, sizeof o);
        o.kind = in.kind; o.material = in.material; o.a = in.a; o.b = in.b;
        memcpy(o.box, in.box, sizeof o.box);
        memcpy(o.v, in.v, sizeof o.v);
        if (in.kind == ON_OBJ_SPHERE && in.material >= 0 && material_reads_uv[in.material]) o.flags |= kObjUsesUV;
    }
    std::vector<int32_t> children;
    for (DObj &o : objects)
        if (o.kind == ON_OBJ_LIST) {
            const int32_t first = (int32_t)children.size();
            children.insert(children.end(), desc->children + o.a, desc->children + o.a + o.b);
            children.push_back(-1);
            o.a = first;
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RT_E_DEVICE, "no HIP device available");
    if (int rc = use_device(device, &device)) return rc;
    s->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->n_cus = prop.multiProcessorCount;
    }
    hipError_t e = s->objects.upload(objects.data(), objects.size());
    if (e == hipSuccess) e = s->children.upload(children.data(), children.size());
    if (e == hipSuccess) e = s->materials.upload(desc->materials, desc->n_materials);
    if (e == hipSuccess) e = s->textures.upload(desc->textures, desc->n_textures);
    if (e == hipSuccess) e = s->images.upload(desc->images, desc->n_images);
    if (e == hipSuccess) e = s->pixels.upload(desc->pixels, desc->n_pixel_bytes);
    if (e == hipSuccess) e = s->perlins.upload(desc->perlins, desc->n_perlins);
    if (e == hipSuccess) e = s->counter.alloc(1);
    if (e != hipSuccess) return fail(RT_E_DEVICE, "scene upload: %s", hipGetErrorString(e));
    s->d.objects = s->objects.p; s->d.children = s->children.p; s->d.materials = s->materials.p; s->d.textures = s->textures.p;
    s->d.images = s->images.p; s->d.pixels = s->pixels.p; s->d.perlins = s->perlins.p;
    s->d.world = desc->world; s->d.lights = desc->lights;
    *out = owner.release();
    return RT_OK;
}

int on_scene_create(const on_scene_desc *desc, int device, on_scene **out) {
    if (!out) return fail(RT_E_INVALID, "null output handle");
    *out = nullptr;
    try {
        return scene_create(desc, device, out);
    } catch (const std::bad_alloc &) {
        return fail(RT_E_INVALID, "on_scene_create: out of host memory");
    } catch (...) {
        return fail(RT_E_DEVICE, "on_scene_create: unknown failure");
    }
}

void on_scene_destroy(on_scene *scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    delete scene;
}

int on_hit_batch(on_scene *scene, int32_t object, const rt_ray *rays, size_t n, double t_min, double t_max, const double *uniforms, uint32_t k,
                 on_hit *out) {
    if (!scene || (n && (!rays || !out))) return fail(RT_E_INVALID, "on_hit_batch: null argument");
    if (object < 0 || (uint32_t)object >= scene->n_objects) return fail(RT_E_INVALID, "on_hit_batch: object %d out of range", object);
    if (!scene->walkable[object]) return fail(RT_E_UNSUPPORTED, "on_hit_batch: object %d is nested deeper than the walk's stack (%d entries) or instance chain (%d) allows", object, kStackCap, kMaxChain);
    if (scene->media_below[object] && (!uniforms || k == 0)) return fail(RT_E_INVALID, "on_hit_batch: a constant-medium lies under object %d: uniforms are needed", object);
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(scene->device));
    dev_buf<rt_ray> d_rays;
    dev_buf<double> d_uniforms;
    dev_buf<on_hit> d_out;
    HIP_TRY(d_rays.upload(rays, n));
    if (uniforms && k) HIP_TRY(d_uniforms.upload(uniforms, n * (size_t)k));
    HIP_TRY(d_out.alloc(n));
    hipLaunchKernelGGL(k_hit_batch, dim3(grid_for(n)), dim3(kBlock), 0, 0, scene->d, object, d_rays.p, n, t_min, t_max,
                       (uniforms && k) ? d_uniforms.p : (const double *)nullptr, k, d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, n * sizeof(on_hit), hipMemcpyDeviceToHost));
    return RT_OK;
}

int on_texture_batch(on_scene *scene, int32_t texture, const double *uvp, size_t n, double *rgb) {
    if (!scene || (n && (!uvp || !rgb))) return fail(RT_E_INVALID, "on_texture_batch: null argument");
    if (texture < 0 || (uint32_t)texture >= scene->n_textures) return fail(RT_E_INVALID, "on_texture_batch: texture %d out of range", texture);
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(scene->device));
    dev_buf<double> d_in, d_out;
    HIP_TRY(d_in.upload(uvp, n * 5));
    HIP_TRY(d_out.alloc(n * 3));
    hipLaunchKernelGGL(k_texture_batch, dim3(grid_for(n)), dim3(kBlock), 0, 0, scene->d, texture, d_in.p, n, d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rgb, d_out.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return RT_OK;
}

int on_pdf_value_batch(on_scene *scene, int32_t object, const double *od, size_t n, double *out) {
    if (!scene || (n && (!od || !out))) return fail(RT_E_INVALID, "on_pdf_value_batch: null argument");
    if (object < 0 || (uint32_t)object >= scene->n_objects) return fail(RT_E_INVALID, "on_pdf_value_batch: object %d out of range", object);
    if (!scene->plain[object]) return fail(RT_E_UNSUPPORTED, "on_pdf_value_batch: object %d is not a sphere, a quad or a list of those", object);
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(scene->device));
    dev_buf<double> d_in, d_out;
    HIP_TRY(d_in.upload(od, n * 6));
    HIP_TRY(d_out.alloc(n));
    hipLaunchKernelGGL(k_pdf_value_batch, dim3(grid_for(n)), dim3(kBlock), 0, 0, scene->d, object, d_in.p, n, d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return RT_OK;
}

int on_render(on_scene *scene, const on_camera *camera, const on_render_params *params, double *accum_dev, void *stream_v, rt_stats *stats) {
    if (!scene || !camera || !params || !accum_dev) return fail(RT_E_INVALID, "on_render: null argument");
    if (!scene->renderable) return fail(RT_E_UNSUPPORTED, "on_render: %s", scene->why_not);
    if (camera->image_width == 0 || camera->image_height == 0 || camera->sqrt_spp == 0) return fail(RT_E_INVALID, "on_render: empty image or no samples");
    if ((uint64_t)camera->image_width * camera->image_height > 0xFFFFFFFFull / 3) return fail(RT_E_INVALID, "on_render: image too large");
    if (params->sample_end < params->sample_begin || params->sample_end > camera->sqrt_spp * camera->sqrt_spp)
        return fail(RT_E_INVALID, "on_render: samples [%u, %u) outside the camera's %u strata", params->sample_begin, params->sample_end, camera->sqrt_spp * camera->sqrt_spp);
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t stream = (hipStream_t)stream_v;
    struct event_pair {
        hipEvent_t a = nullptr, b = nullptr;
        ~event_pair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIP_TRY(hipMemsetAsync(scene->counter.p, 0, sizeof(unsigned long long), stream));
    if (stats) {
        HIP_TRY(hipEventCreate(&ev.a));
        HIP_TRY(hipEventCreate(&ev.b));
        HIP_TRY(hipEventRecord(ev.a, stream));
    }
    const uint32_t spp = params->sample_end - params->sample_begin;
    const size_t n_values = (size_t)camera->image_width * camera->image_height * 3;
    {
        // The samples go through in windows of as many as the slot buffer holds (4 GiB: all of them for frames up to
        // about 1.7 million pixels x 100 samples), each window added to the image in sample order by k_combine.
        const uint32_t tiles8 = ((camera->image_width + 7u) / 8u) * ((camera->image_height + 7u) / 8u);
        uint32_t window = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(spp, (kSlotBudgetBytes / sizeof(double)) / n_values));
        if (const char *e = getenv("ONNA_WINDOW")) window = (uint32_t)std::max(1, std::min<int>(atoi(e), (int)std::max(1u, spp)));   // (tests)
        if (spp && scene->slot_capacity < n_values * window) {   // (grown on demand; the allocation may synchronise the device)
            if (scene->slots.p) { (void)hipFree(scene->slots.p); scene->slots.p = nullptr; scene->slot_capacity = 0; }
            HIP_TRY(scene->slots.alloc(n_values * window));
            scene->slot_capacity = n_values * window;
        }
        for (uint32_t w0 = params->sample_begin; w0 < params->sample_end;) {
            const uint32_t most = std::min(window, params->sample_end - w0);
            // A wave takes a run of the window's samples.  Long runs keep the lanes busy (a wave idles for about one path at its
            // end, whatever its length); short ones at the end of the launch -- blocks start in index order -- keep the tail
            // short: each run takes a share of what is left, down to kShortestRun samples.  A frame of few tiles cannot afford
            // long runs at all -- its dearest tile's run would outlast everything else -- so a run is at most 1/8 of a wave
            // slot's share of the window.
            const uint64_t slots_for_waves = (uint64_t)scene->n_cus * 12;
            const uint32_t longest = (uint32_t)std::max<uint64_t>(kShortestRun, std::min<uint64_t>(64, (uint64_t)most * tiles8 / (slots_for_waves * 8)));
            const char *forced = getenv("ONNA_RUNS");   // (tests, A/B) "40,25,15,...": the last length repeats; at most kMaxRuns runs a pass
            SampleRuns runs;
            runs.n = 0; runs.start[0] = 0;
            uint32_t n_runs = 0, covered = most;
            if (forced) {
                uint32_t len = 1;
                for (uint32_t left = most; left && runs.n < kMaxRuns;) {
                    if (*forced) { len = (uint32_t)std::max(1l, strtol(forced, const_cast<char **>(&forced), 10)); if (*forced == ',') ++forced; }
                    const uint32_t take = std::min(len, left);
                    runs.start[runs.n + 1] = runs.start[runs.n] + take;
                    ++runs.n;
                    left -= take;
                }
                n_runs = runs.n; covered = runs.start[runs.n];
            } else {
                for (uint32_t left = most; left; ++n_runs) left -= run_length(left, longest);
            }
            const uint32_t w1 = w0 + covered;
            hipLaunchKernelGGL(k_render, dim3(tiles8 * n_runs), dim3(64), 0, stream, scene->d, *camera, w0, covered, longest, runs, params->seed, scene->slots.p,
                               scene->counter.p);
            hipLaunchKernelGGL(k_combine, dim3(grid_for(n_values)), dim3(kBlock), 0, stream, scene->slots.p, n_values, w1 - w0, accum_dev);
            w0 = w1;
        }
    }
    HIP_TRY(hipGetLastError());
    if (stats) {
        HIP_TRY(hipEventRecord(ev.b, stream));
        HIP_TRY(hipEventSynchronize(ev.b));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        unsigned long long rays = 0;
        HIP_TRY(hipMemcpy(&rays, scene->counter.p, sizeof rays, hipMemcpyDeviceToHost));
        stats->kernel_ms = ms;
        stats->rays = rays;
        stats->paths = (uint64_t)camera->image_width * camera->image_height * (params->sample_end - params->sample_begin);
    }
    return RT_OK;
}

int on_render_host(on_scene *scene, const on_camera *camera, const on_render_params *params, double *accum_host, rt_stats *stats) {
    if (!scene || !camera || !params || !accum_host) return fail(RT_E_INVALID, "on_render_host: null argument");
    HIP_TRY(hipSetDevice(scene->device));
    const size_t n = (size_t)camera->image_width * camera->image_height * 3;
    dev_buf<double> d;
    HIP_TRY(d.alloc(n));
    HIP_TRY(hipMemset(d.p, 0, n * sizeof(double)));
    if (int rc = on_render(scene, camera, params, d.p, nullptr, stats)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(accum_host, d.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return RT_OK;
}

int on_resolve(int device, const double *accum_dev, size_t npix, uint32_t spp, uint8_t *rgb8_dev, void *stream) {
    if (npix && (!accum_dev || !rgb8_dev)) return fail(RT_E_INVALID, "on_resolve: null argument");
    if (spp == 0) return fail(RT_E_INVALID, "on_resolve: no samples");
    if (npix == 0) return RT_OK;
    if (int rc = use_device(device, nullptr)) return rc;
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(npix * 3)), dim3(kBlock), 0, (hipStream_t)stream, accum_dev, npix * 3, spp, rgb8_dev);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // extern "C"
