// rtiow_hip.hip -- kernels and C ABI (include/rtiow_hip.h) of the MI355X path tracer.
//
// Host side: validates a scene description, lays it out for the kernels (nodes with
// both child boxes, primitives in left-to-right leaf order) and launches.  There is no
// CPU implementation of any entry point: without a HIP device every call fails.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is opened on first use (rccl_api)
#include <dlfcn.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <array>
#include <future>
#include <map>
#include <string>
#include <thread>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "rt_device.hpp"
#include "bvh_device.hpp"
#include "rtiow_hip.h"

using namespace rtd;

// ------------------------------------------------------------------- kernels ----
namespace {

constexpr int kBlock = 256;                 // batch kernels
constexpr size_t kLdsBudget = 160 * 1024;   // LDS per CU on gfx950; the render kernel runs one block per CU

__device__ __forceinline__ Ray load_ray(const rt_ray *r) {
    Ray o;
    o.o = mk(r->origin[0], r->origin[1], r->origin[2]);
    o.d = mk(r->direction[0], r->direction[1], r->direction[2]);
    o.time = r->time;
    return o;
}

__device__ __forceinline__ void store_ray(rt_ray *r, const Ray &o) {
    r->origin[0] = o.o.x; r->origin[1] = o.o.y; r->origin[2] = o.o.z;
    r->direction[0] = o.d.x; r->direction[1] = o.d.y; r->direction[2] = o.d.z;
    r->time = o.time;
}

// object::hit for a batch of rays, one ray per lane
__global__ __launch_bounds__(kBlock) void k_trace_batch(DScene sc, const rt_ray *rays, size_t n, double tmin,
                                                       double tmax, rt_hit *out) {
    extern __shared__ int lds_stack[];
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Ray r = load_ray(rays + i);
    RayCtx ctx = make_ctx(r, sc.far, sc.reach_a);
    double t;
    int prim;
    rt_hit h;
    memset(&h, 0, sizeof h);
    h.prim = -1;
    if (world_closest(sc, ctx, tmin, tmax, lds_stack + threadIdx.x, kBlock, t, prim)) {
        v3 p, nn;
        hit_point(sc, prim, ctx, t, p, nn);
        h.hit = 1;
        h.prim = sc.prims[prim].orig;
        h.t = t;
        h.p[0] = p.x; h.p[1] = p.y; h.p[2] = p.z;
        h.normal[0] = nn.x; h.normal[1] = nn.y; h.normal[2] = nn.z;
    }
    out[i] = h;
}

// aabb::hit for a batch
__global__ __launch_bounds__(kBlock) void k_aabb_batch(const double *boxes, const rt_ray *rays, const double *tmin,
                                                      const double *tmax, size_t n, uint8_t *out) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const FarSlack none = {0, 0, 0, 0, 0};
    RayCtx ctx = make_ctx(load_ray(rays + i), none);
    const double *b = boxes + i * 6;
    double bx[6] = {b[0], b[3], b[1], b[4], b[2], b[5]};
    double entry;
    out[i] = box_test(bx, ctx, tmin[i], tmax[i], entry) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_texture_batch(DScene sc, uint32_t tex, const double *p, size_t n, double *rgb) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    v3 c = texture_value(sc, tex, mk(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]));
    rgb[i * 3] = c.x; rgb[i * 3 + 1] = c.y; rgb[i * 3 + 2] = c.z;
}

__global__ __launch_bounds__(kBlock) void k_scatter_batch(DScene sc, const uint32_t *mat, const rt_ray *rays,
                                                         const rt_hit *hits, const double *tape, uint32_t k, size_t n,
                                                         uint8_t *scattered, double *att, rt_ray *out, int32_t *used) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    TapeRng g = {tape + i * (size_t)k, (int)k, 0, false};
    Ray r = load_ray(rays + i);
    const rt_hit *h = hits + i;
    v3 a;
    Ray o;
    bool s = scatter(sc, mat[i], r, mk(h->p[0], h->p[1], h->p[2]), mk(h->normal[0], h->normal[1], h->normal[2]), g, a, o);
    scattered[i] = s ? 1 : 0;
    att[i * 3] = a.x; att[i * 3 + 1] = a.y; att[i * 3 + 2] = a.z;
    store_ray(out + i, o);
    used[i] = g.over ? -1 : g.pos;
}

__global__ __launch_bounds__(kBlock) void k_camera_batch(DCamera cam, const double *st, const double *tape, uint32_t k,
                                                        size_t n, rt_ray *out, int32_t *used) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    TapeRng g = {tape + i * (size_t)k, (int)k, 0, false};
    Ray r = camera_ray(cam, st[i * 2], st[i * 2 + 1], g);
    store_ray(out + i, r);
    used[i] = g.over ? -1 : g.pos;
}

// Adds one sample's radiance to a pixel's three sums in LDS.  The sums are integers in units of 2^-50:
// for |x| < 2, x 2^50 + 1.5 2^52 has its unit in the last place, so its bit pattern minus that of
// 1.5 2^52 is x 2^50 rounded to the nearest integer (one fma and one subtraction).  Sums of integers
// do not depend on the order of the additions.  A launch may add 4096 samples to a pixel.
// `wide`: the scene has an albedo or texture colour outside [0, 1], so that a sample's radiance is not
// bounded by the sky's 1; its sums are doubles, added in the order the wave reaches them.
__device__ __forceinline__ long long radiance_units(double x) {
    // (the low word of 1.5 2^52 is zero: the subtraction touches the high word only)
    const double v = __builtin_fma(x, 0x1p50, 0x1.8p52);
    return (long long)(((unsigned long long)((uint32_t)__double2hiint(v) - 0x43380000u) << 32) | (uint32_t)__double2loint(v));
}
__device__ __forceinline__ void add_radiance(long long *q, double cx, double cy, double cz, uint32_t stride, bool wide) {
    if (wide) {
        __hip_atomic_fetch_add(reinterpret_cast<double *>(q), cx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(reinterpret_cast<double *>(q + stride), cy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(reinterpret_cast<double *>(q + 2u * stride), cz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    __hip_atomic_fetch_add(q, radiance_units(cx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(q + stride, radiance_units(cy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(q + 2u * stride, radiance_units(cz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The pixel/sample loop (main.cpp:126-141) with ray_colour (main.cpp:30-50) unrolled
// into one loop over ray segments.
//
// Persistent blocks: the grid is one block per CU; each wave draws pixel tiles from a
// device-wide queue until the frame is done, so no CU idles behind a slow tile.  When
// the tree fits, the block first copies it into LDS (nodes and spheres, re-strided
// against bank conflicts) and every traversal step reads LDS instead of the vector cache.
//
// A wave owns its tile's pixels x (s_end - s_begin) paths, numbered k = sample * pixels +
// pixel-in-tile; a lane whose path has ended claims the next unclaimed k (lanes that
// claim together take consecutive numbers), so every lane stays busy until the tile's
// paths run out, whatever the mix of short sky paths and long glass paths in the tile --
// and then the wave opens its next tile beside the one that is still draining (see the
// ring in the kernel).  A finished path adds its radiance to its tile's sums in LDS; the
// random stream is keyed by (pixel, sample), so the set of paths -- and every path's
// arithmetic -- is independent of which lane ran it, and the sums, being integers, of the
// order in which they were added.
#ifndef RT_LEAF_QUORUM
#define RT_LEAF_QUORUM 16         // trees in LDS (measured: 8 and 24 lose 0.3-0.8 % on the 485-sphere scenes)
#endif
#ifndef RT_LEAF_QUORUM_PAIRS
#define RT_LEAF_QUORUM_PAIRS 24   // ... whose leaves hold the reference's pairs (book-2: +0.6 % over 16)
#endif
#ifndef RT_LEAF_QUORUM_LARGE
#define RT_LEAF_QUORUM_LARGE 8    // trees in device memory: a waiting leaf holds its lane up for longer walks (10^6 spheres: +2.8 % over 16)
#endif
// Shapes of the kernel.  When nodes, spheres and full-depth stacks fit in LDS every traversal
// step reads LDS; the block is 1024 threads (4 waves per SIMD, measured 4-5 % faster than 3)
// if that still fits, else 768.  Otherwise the block is 1024 threads (the walk then waits on
// device memory, which more waves hide), LDS holds the top levels of the tree -- the first
// n_lnodes nodes, which are numbered breadth-first -- and the first stack_cap entries of each
// lane's stack; deeper nodes and all spheres are read from device memory, deeper stack
// entries spill to it.
constexpr int kWavesMax = 4, kWavesMin = 3;   // waves per SIMD
constexpr size_t kCameraLds = 192;            // the block's copy of the camera
constexpr size_t kHeadLds = kCameraLds + 64 + 16 * 16;  // ... followed by the scene's FarSlack and reach, 1/nx and 1/ny, and each wave's four open tiles
constexpr int kSpillStackCap = 12;                      // LDS stack entries per lane when the tree is in device memory
#ifdef RT_LANE_STATS
constexpr size_t kPassCounters = 12;                    // of one pass: [0] rays, [1] the tile queue, [4..11] a queue position per XCD
constexpr size_t kQueueCounters = 2 * kPassCounters;    // the frame's, then the costing pass's
constexpr size_t kCounters = kQueueCounters + 16 * 16 * 1024;        // + 16 statistics slots (12 lane counts, 3 phase clocks) per wave of up to 1024 blocks
#else
constexpr size_t kPassCounters = 12;                    // of one pass: [0] rays, [1] the tile queue, [4..11] a queue position per XCD
constexpr size_t kQueueCounters = 2 * kPassCounters;    // the frame's, then the costing pass's
constexpr size_t kCounters = kQueueCounters;
#endif
// strip q of `regions` covers the tile columns [q tiles_x / regions, (q + 1) tiles_x / regions)
__host__ __device__ inline uint32_t strip_begin(uint32_t q, uint32_t tiles_x, uint32_t regions) { return (uint32_t)((uint64_t)q * tiles_x / regions); }
__host__ __device__ inline uint32_t strip_of(uint32_t x, uint32_t tiles_x, uint32_t regions) {
    uint32_t q = (uint32_t)(((uint64_t)x * regions + regions - 1) / tiles_x);   // an estimate, then exact
    if (q >= regions) q = regions - 1;
    while (q > 0 && strip_begin(q, tiles_x, regions) > x) --q;
    while (q + 1 < regions && strip_begin(q + 1, tiles_x, regions) <= x) ++q;
    return q;
}
constexpr uint32_t kOrderMinSamples = 32;               // frames at least this deep get their tiles ordered by cost
constexpr uint32_t kSamplesPerLaunch = 4096;            // samples a launch may add to a pixel's 64-bit sums (2^51 units each at most)

// COSTING: the pass that ranks the tiles -- the same walk over a few samples, but instead of
// adding to the image each tile records how much work it was (tile_cost).
// DEFER: the walk postpones the box tests of static-sphere leaves (world_closest_voted); chosen
// per scene.
// WIDE (with DEFER): the walk runs over the four-wide tree (world_closest_wide): LDS_TREE then means the
// DNode4 records and the spheres are in LDS; without it everything is read from device memory and LDS
// holds the stacks only.
template <class AccT, bool LDS_TREE, int WAVES, bool COSTING, bool DEFER, bool WIDE = false>
__global__ __launch_bounds__(256 * WAVES) void k_render(DScene sc, DCamera cam_arg, uint32_t nx, uint32_t ny, uint32_t s_begin,
                                                       uint32_t s_end, uint32_t max_depth, double t_min, uint64_t seed,
                                                       AccT *accum, unsigned long long *counters, uint32_t n_lnodes,
                                                       uint32_t n_lleaves, uint32_t stack_cap, int *spill,
                                                       const uint32_t *tile_order, uint32_t *tile_cost, uint32_t tile_shift,
                                                       uint32_t sums_per_wave, uint32_t regions) {
    constexpr int kRenderBlock = 256 * WAVES, kRenderWaves = 4 * WAVES;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // LDS: [camera][far slack][tile sums: waves x sums_per_wave bytes][stacks: stack_cap x block ints][nodes][leaves]
    // The camera's 24 doubles are needed once per path; as kernel arguments they would sit in 48
    // scalar registers for the whole kernel and push other scalars out into vector-register lanes,
    // to be fetched back with vector instructions in the middle of the hot loops.
    static_assert(sizeof(DCamera) == kCameraLds, "camera copy in LDS");
    if (tid < kCameraLds / sizeof(double)) reinterpret_cast<double *>(lds_raw)[tid] = reinterpret_cast<const double *>(&cam_arg)[tid];
    static_assert(sizeof(FarSlack) == 20 && kHeadLds - kCameraLds >= sizeof(FarSlack), "far slack copy in LDS");
    static_assert(offsetof(DScene, reach_a) == offsetof(DScene, far) + 20 && offsetof(DScene, reach_b) == offsetof(DScene, far) + 32 && kHeadLds - kCameraLds >= 44,
                  "far slack and reach are copied together");
    if (tid < 11) reinterpret_cast<float *>(lds_raw + kCameraLds)[tid] = reinterpret_cast<const float *>(&sc.far)[tid];
    long long *const sums = reinterpret_cast<long long *>(lds_raw + kHeadLds + (size_t)wave * sums_per_wave);
    char *const stacks = lds_raw + kHeadLds + (size_t)kRenderWaves * sums_per_wave;
    char *ltree = stacks + (size_t)stack_cap * kRenderBlock * sizeof(int);
    const LaneStack st = {reinterpret_cast<int *>(stacks) + tid, kRenderBlock, (int)stack_cap,
                          spill ? spill + (size_t)blockIdx.x * kRenderBlock + tid : nullptr, (int)(gridDim.x * kRenderBlock)};
    TreeView tree = {sc.nodes, sc.leaves, sc.prims, nullptr, nullptr, (int32_t)n_lnodes};
    {
        char *lnodes = ltree, *lleaves = ltree + (((size_t)n_lnodes * (WIDE ? LNODE4_STRIDE : LNODE_STRIDE) + 7) & ~(size_t)7);
        for (uint32_t i = tid; WIDE && i < n_lnodes * 28u; i += kRenderBlock) {   // 24 planes and the four child words (see wide_word)
            const uint32_t n = i / 28u, w = i % 28u;
            uint32_t v = reinterpret_cast<const uint32_t *>(sc.nodes4 + n)[w];
            if (w >= 24u) v = wide_word((int32_t)v);
            reinterpret_cast<uint32_t *>(lnodes + (size_t)n * LNODE4_STRIDE)[w] = v;
        }
        for (uint32_t i = tid; !WIDE && i < n_lnodes * 15u; i += kRenderBlock) {   // 15 words of a node record are copied
            const uint32_t n = i / 15u, w = i % 15u;
            uint32_t v = reinterpret_cast<const uint32_t *>(sc.nodes + n)[w];
            // a tree that is in LDS whole refers to its nodes by the byte offset of their records
            if (LDS_TREE && (w == 12u || w == 13u) && (int32_t)v >= 0) v *= LNODE_STRIDE;
            reinterpret_cast<uint32_t *>(lnodes + (size_t)n * LNODE_STRIDE)[w] = v;
        }
        for (uint32_t i = tid; DEFER && i < n_lleaves * 14u; i += kRenderBlock) {   // static scenes: 6 doubles of the box, centre and r*r of each sphere
            const uint32_t n = i / 14u, w = i % 14u;
            double v;
            if (w < 6u) v = sc.leaves[n].box[w];
            else {
                const DPrim *p = sc.prims + 2u * n + (w - 6u) / 4u;
                const uint32_t f = (w - 6u) % 4u;
                v = f < 3u ? p->c0[f] : p->r * p->r;   // the sphere test wants radius * radius (sphere.cpp:16)
            }
            reinterpret_cast<double *>(lleaves + (size_t)n * SLEAF_STRIDE)[w] = v;
        }
        for (uint32_t i = tid; !DEFER && i < n_lleaves * 28u; i += kRenderBlock) {   // 6 doubles of the box, 11 of each sphere record
            const uint32_t n = i / 28u, w = i % 28u;
            double v;
            if (w < 6u) v = sc.leaves[n].box[w];
            else {
                v = reinterpret_cast<const double *>(sc.prims + 2u * n + (w - 6u) / 11u)[(w - 6u) % 11u];
                if ((w - 6u) % 11u == 3u) v = v * v;   // the sphere test wants radius * radius (sphere.cpp:16)
            }
            reinterpret_cast<double *>(lleaves + (size_t)n * LLEAF_STRIDE)[w] = v;
        }
        tree.lnodes = lnodes; tree.lleaves = lleaves;
        __syncthreads();
    }

    // the denominators' part of the divisions ii / nx and jj / ny (see div_by_shared), read back where a path is claimed
    double *const rcp_size = reinterpret_cast<double *>(lds_raw + kCameraLds + 48);
    if (tid == 0) { rcp_size[0] = refined_rcp((double)nx); rcp_size[1] = refined_rcp((double)ny); }
    uint32_t *const ring_tile = reinterpret_cast<uint32_t *>(lds_raw + kCameraLds + 64) + wave * 4;   // the tile in each of this wave's ring entries
    uint32_t *const strips_done = reinterpret_cast<uint32_t *>(lds_raw + kCameraLds + 44);            // bit q: the queue of strip q is used up
    if (tid == 0) *strips_done = 0;
    __syncthreads();
    // a tile is 8x8 or 4x4 pixels (tile_shift 3 or 2)
    const uint32_t tile_side = 1u << tile_shift, tile_px = tile_side * tile_side;
    const uint32_t tiles_x = (nx + tile_side - 1) >> tile_shift, tiles_y = (ny + tile_side - 1) >> tile_shift, n_tiles = tiles_x * tiles_y;
    unsigned long long nrays = 0;
    LaneStats lane_stats;
#ifdef RT_LANE_STATS
    if (!COSTING) lane_stats.c = counters + kQueueCounters + 16 * (blockIdx.x * 16 + wave);   // zeroed by rt_render
#endif

    // A wave keeps up to `ring` tiles open at a time: one whose paths it is still claiming, the others
    // draining.  Lanes whose path has ended never wait for the last long paths of a tile: when the tile
    // in hand has no unclaimed path left the wave opens the next one beside it.  A tile's sums are kept
    // in LDS as 64-bit integers (radiance in units of 2^-50): integer sums do not depend on the order
    // of the additions, so the image does not depend on which lane, wave or tile size carried a path,
    // nor on how tiles overlapped.  Nothing is counted per round: whether an open tile is finished
    // (no lane carries a path of it) is only asked when the wave needs its place in the ring.
    const uint32_t total = tile_px * (s_end - s_begin);   // paths of a tile
    // (a scene whose radiance is not bounded keeps sums of doubles: one tile at a time then, so that the
    // order of the additions, and with it every bit of the image, is the same from run to run)
    const bool wide = sc.wide_radiance != 0;
    const uint32_t ring = (COSTING || wide) ? 1u : (sums_per_wave / (tile_px * 24u) < 4u ? sums_per_wave / (tile_px * 24u) : 4u);
    for (uint32_t i = lane; i < sums_per_wave / 8u; i += 64u) sums[i] = 0;
    uint32_t ring_used = 0;                 // bit e: entry e holds a tile that has not been written out
    uint32_t cur = 0, next = total, cur_x = 0, cur_y = 0;   // the entry being claimed from and its first unclaimed path
    bool queue_done = false;
    uint32_t work = 0;   // per lane: traversal iterations and rounds it took part in (see world_closest_voted)

    bool alive = false;
    uint32_t slot = 0;   // pixel-in-tile of the lane's current path, and (from bit 6) the ring entry of its tile
    Ray r = {mk(0, 0, 0), mk(0, 0, 1), 0};
    v3 thr = mk(1, 1, 1);
    uint32_t depth = 0;
    StreamRng g = {1, 0, 0, 0};

    // a finished tile: its sums go to the image (or, in the costing pass, its cost to the cost table)
    auto flush = [&](uint32_t e, uint32_t tile) {
        if (COSTING) {
            // the tile's cost is the work its busiest lane counted (a count, not a duration: durations
            // depend on what else the CU is running)
            uint32_t segs = work;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(segs, off); segs = o > segs ? o : segs; }
            if (lane == 0) tile_cost[tile] = segs;
            work = 0;
            return;
        }
        const uint32_t x = ((tile % tiles_x) << tile_shift) + (lane & (tile_side - 1u)), y = ((tile / tiles_x) << tile_shift) + (lane >> tile_shift);
        if (lane < tile_px) {
            long long *q = sums + (size_t)e * 3u * tile_px + lane;
            if (x < nx && y < ny) {
                AccT *a = accum + ((size_t)y * nx + x) * 3;
                auto value = [wide](long long v) { return wide ? __longlong_as_double(v) : (double)v * 0x1p-50; };
                a[0] += (AccT)value(q[0]); a[1] += (AccT)value(q[tile_px]); a[2] += (AccT)value(q[2u * tile_px]);
            }
            q[0] = 0; q[tile_px] = 0; q[2u * tile_px] = 0;
        }
    };
    // writes out every open tile that no lane works for any more (the tile in hand must be used up)
    auto retire = [&]() {
#pragma unroll
        for (uint32_t e = 0; e < 4u; ++e) {
            if (!(ring_used & (1u << e))) continue;
            if (__ballot(alive && (slot >> 6) == e) != 0) continue;
            flush(e, (uint32_t)__builtin_amdgcn_readfirstlane((int)ring_tile[e]));
            ring_used &= ~(1u << e);
        }
    };

    for (;;) {
        // Tile management, outside the loop over rounds: the tile in hand is used up (or there is none
        // yet) and some lane has nothing to do -- open the next tile, if the ring has room for it.
        if (next >= total && !queue_done) {
            if (ring_used == (1u << ring) - 1u) retire();
            if (ring_used != (1u << ring) - 1u) {
                uint32_t tile = 0;
                if (lane == 0) {
                    if (regions > 1) {
                        // A tree in device memory: the frame is cut into vertical strips, one per XCD, each with its own
                        // queue, so that the nodes a strip's rays visit gather in that XCD's L2
                        // instead of every L2 holding a thin slice of everything.  Blocks are dealt to the XCDs in turn
                        // (block b runs on XCD b % 8); a block whose strip is used up helps with the next ones.
                        // (an ordered frame's strips are listed behind the order, dearest tile first; otherwise a strip's tiles go row
                        // by row.  A strip found used up is marked in LDS, so that the block does not ask its counter again.)
                        const uint32_t *seg = tile_order ? tile_order + n_tiles : nullptr;   // where each strip starts in the order, and n_tiles last
                        tile = n_tiles;
                        const uint32_t used_up = *strips_done;
                        uint32_t q = blockIdx.x % regions;
                        for (uint32_t tries = 0; tries < regions; ++tries, q = q + 1 == regions ? 0 : q + 1) {
                            if ((used_up >> q) & 1u) continue;
                            const uint32_t x0 = strip_begin(q, tiles_x, regions), width = strip_begin(q + 1, tiles_x, regions) - x0;
                            const uint32_t from = seg ? seg[q] : 0u;
                            const uint32_t len = seg ? seg[q + 1] - from : width * tiles_y;
                            const uint32_t pos = (uint32_t)atomicAdd(counters + 4 + q, 1ull);
                            if (pos < len) { tile = seg ? tile_order[from + pos] : (pos / width) * tiles_x + x0 + pos % width; break; }
                            atomicOr(strips_done, 1u << q);
                        }
                    } else {
                        tile = (uint32_t)atomicAdd(counters + 1, 1ull);
                        if (tile < n_tiles && tile_order) tile = tile_order[tile];
                    }
                }
                tile = __builtin_amdgcn_readfirstlane(tile);
                if (tile >= n_tiles) {
                    queue_done = true;
                } else {
                    cur = (uint32_t)__builtin_ctz(~ring_used);
                    if (lane == 0) ring_tile[cur] = tile;
                    ring_used |= 1u << cur;
                    next = 0;
                    cur_x = (tile % tiles_x) << tile_shift; cur_y = (tile / tiles_x) << tile_shift;
                }
            }
        }
        if (queue_done && next >= total && !__any(alive)) break;

        // rounds: until the tile in hand runs out of paths while a lane is idle (and more tiles may come)
        for (;;) {
#ifdef RT_LANE_STATS
            const long long clk0 = clock64();
            long long clk1 = clk0, clk2 = clk0;
#endif
            // lanes without a path claim the next unclaimed ones, in lane order
            const unsigned long long want = __ballot(!alive);
            if (want != 0 && next < total) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                const uint32_t k = next + rank;
                const uint32_t claimed = (uint32_t)__popcll(want);
                next = (total - next > claimed) ? next + claimed : total;
                if (!alive && k < total) {
                    const uint32_t pix = k & (tile_px - 1u);
                    const uint32_t x = cur_x + (pix & (tile_side - 1u)), y = cur_y + (pix >> tile_shift);
                    if (x < nx && y < ny) {   // (a ragged tile's pixels beyond the image have nothing to trace)
                        slot = pix | (cur << 6);
                        const uint32_t smp = s_begin + (k >> (2u * tile_shift));
                        // (the seed passes through an opaque move: the ten pairs of round keys derived
                        // from it are then formed here by scalar additions, instead of being hoisted
                        // out of the loops into twenty scalar registers that do not exist)
                        uint64_t seed_here = seed;
                        asm volatile("" : "+s"(seed_here));
                        g.init(seed_here, y * nx + x, smp);
                        // the reference's row index runs bottom-up (main.cpp:126)
                        const double ii = (double)x + g.canonical();   // dist{0, 1}: u * 1 + 0 = u
                        const double jj = (double)(ny - 1 - y) + g.canonical();
                        // ii / nx and jj / ny (main.cpp:132-133), the denominators' part of the division done once per wave
                        const double u = div_by_shared(ii, (double)nx, rcp_size[0]), v = div_by_shared(jj, (double)ny, rcp_size[1]);
                        r = camera_ray(*reinterpret_cast<const DCamera *>(lds_raw), u, v, g);
                        thr = mk(1, 1, 1);
                        depth = 0;
                        alive = true;
                    }
                }
            }
            if (!__any(alive)) break;   // nothing to trace: more tiles, or the end
#ifdef RT_LANE_STATS
            clk1 = clock64();
#endif
            lane_stats.add(2, (uint32_t)__popcll(__ballot(alive)));
            if (want != 0) lane_stats.add(5, (uint32_t)__popcll(want));
            if (alive) {
                const float *head = reinterpret_cast<const float *>(lds_raw + kCameraLds);
                const RayCtx ctx = make_ctx(r, *reinterpret_cast<const FarSlack *>(head), DEFER ? nullptr : head + 5);
                double t;
                int prim;
                ++nrays;
                if (COSTING) work += 8;   // claiming, set-up and shading of one segment, in node-step units
                bool hit_something;
                if (WIDE) hit_something = world_closest_wide<(LDS_TREE ? RT_LEAF_QUORUM : RT_LEAF_QUORUM_LARGE), LDS_TREE, COSTING>(sc, tree, ctx, t_min, DBL_MAX, st, t, prim, work, lane_stats);
                else hit_something = world_closest_voted<(LDS_TREE ? (DEFER ? RT_LEAF_QUORUM : RT_LEAF_QUORUM_PAIRS) : RT_LEAF_QUORUM_LARGE), LDS_TREE, COSTING, DEFER>(sc, tree, ctx, t_min, DBL_MAX, st, t, prim, work, lane_stats);
#ifdef RT_LANE_STATS
                clk2 = clock64();
#endif
                if (hit_something) {
                    lane_stats.add(3, (uint32_t)__popcll(__ballot(true)));
                    v3 p, nn, att;
                    Ray nextray;
                    hit_point(sc, prim, ctx, t, p, nn);
                    // scatter first, depth test second (main.cpp:33-34)
                    const bool ok = scatter(sc, sc.prims[prim].mat, r, p, nn, g, att, nextray);
                    if (depth < max_depth && ok) {
                        thr = mk(thr.x * att.x, thr.y * att.y, thr.z * att.z);
                        r = nextray;
                        ++depth;
                    } else {
                        alive = false;  // absorbed or too deep: black (main.cpp:42)
                    }
                } else {
                    lane_stats.add(4, (uint32_t)__popcll(__ballot(true)));
                    const v3 c = sky(r.d);
                    if (!COSTING) {
                        long long *q = sums + (slot >> 6) * 3u * tile_px + (slot & 63u);
                        add_radiance(q, c.x * thr.x, c.y * thr.y, c.z * thr.z, tile_px, wide);
                    }
                    alive = false;
                }
            }
#ifdef RT_LANE_STATS
            if (lane_stats.c && lane == 0) {   // clocks of this round: claiming, set-up and walk, shading
                const long long clk3 = clock64();
                lane_stats.c[12] += (unsigned long long)(clk1 - clk0); lane_stats.c[13] += (unsigned long long)(clk2 - clk1); lane_stats.c[14] += (unsigned long long)(clk3 - clk2);
            }
#endif
            if (next >= total && !queue_done && !__all(alive)) break;   // a lane is idle and the tile in hand has nothing for it
        }
    }
    retire();   // (no lane is alive: every tile still open is written out)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nrays += __shfl_xor(nrays, off);
    if (lane == 0) atomicAdd(counters, nrays);
}

// supersampler::value, gamma2, convert (supersampler.cpp:12-19, colour.cpp:19-21, main.cpp:26-28)
__global__ __launch_bounds__(kBlock) void k_resolve(const float *accum, size_t nvals, uint32_t spp, uint8_t *rgb8) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= nvals) return;
    double mean = spp ? (double)accum[i] / (double)spp : 0.0;
    int q = (int)(255.99 * sqrt(mean));
    rgb8[i] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
}

// Orders the tiles of a frame by measured cost, dearest first (a counting sort over 1024
// cost classes in one block; ties in arbitrary order).  A wave works through a tile's whole
// sample range, and tiles differ in cost by an order of magnitude (sky against glass), so
// handing them out in image order leaves a few waves finishing the last, dear tiles while
// the rest of the chip idles; longest-first keeps the end of the frame to the cheap ones.
//
// Costs are measured on 8x8-pixel tiles (a few samples of a smaller tile would not fill a
// wave); when the frame is rendered in smaller tiles each inherits the cost of the 8x8 tile
// it lies in (`sub` = log2 of the ratio of the sides).
constexpr int kOrderBlock = 1024;
constexpr uint32_t kRegions = 8;   // vertical strips of a frame whose tree is in device memory: one per XCD (see k_render)
// `regions` > 1: the tiles of each vertical strip of the frame are ordered on their own, strip after strip, and
// order[n .. n + regions] receives where each strip starts (and, last, n).
__global__ __launch_bounds__(kOrderBlock) void k_order_tiles(const uint32_t *cost, uint32_t cost_tiles_x, uint32_t n_cost,
                                                            uint32_t tiles_x, uint32_t n, uint32_t sub, uint32_t *order, uint32_t regions) {
    __shared__ uint32_t hist[kOrderBlock];
    __shared__ uint32_t cmax, base;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { cmax = 1; base = 0; }
    __syncthreads();
    uint32_t m = 1;
    for (uint32_t i = tid; i < n_cost; i += kOrderBlock) m = cost[i] > m ? cost[i] : m;
    atomicMax(&cmax, m);
    __syncthreads();
    const unsigned long long top = cmax;
    auto klass = [&](uint32_t i) {
        const uint32_t x = i % tiles_x, y = i / tiles_x;
        return (uint32_t)((unsigned long long)cost[(y >> sub) * cost_tiles_x + (x >> sub)] * (kOrderBlock - 1) / top);
    };
    // (vertical strips: measured 45.1 ms on 10^6 spheres against 46.8 for horizontal bands, 46.4 / 46.9 for 4 x 2 / 2 x 4
    // blocks and 47.9 for one queue; the order inside a strip -- dearest first or row by row -- makes no difference)
    auto region = [&](uint32_t i) { return regions > 1 ? strip_of(i % tiles_x, tiles_x, regions) : 0u; };
    for (uint32_t r = 0; r < (regions > 1 ? regions : 1u); ++r) {
        hist[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kOrderBlock)
            if (region(i) == r) atomicAdd(&hist[klass(i)], 1u);
        __syncthreads();
        if (tid == 0) {   // start of each class in the output, dearest class first
            uint32_t at = base;
            if (regions > 1) order[n + r] = base;
            for (int b = kOrderBlock - 1; b >= 0; --b) { const uint32_t c = hist[b]; hist[b] = at; at += c; }
            base = at;
        }
        __syncthreads();
        for (uint32_t i = tid; i < n; i += kOrderBlock)
            if (region(i) == r) order[atomicAdd(&hist[klass(i)], 1u)] = i;
        __syncthreads();
    }
    if (tid == 0 && regions > 1) order[n + regions] = n;
}

}  // namespace

// ------------------------------------------------------------------ host side ----
namespace {

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

// A device argument of -1 stands for the calling thread's current HIP device.
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

}  // namespace

// The working memory of rt_render_noalloc: one array per GPU the library runs on, held until the
// process ends.  In host code the name `scratch` is a stand-in, not a device address: the address on
// each device is asked for once (scratch_on_device) and only that pointer is handed to kernels and
// to hipMemsetAsync.
constexpr size_t kScratchCounterBytes = 256;                   // the frame counters (kQueueCounters of them)
static_assert(kQueueCounters * sizeof(unsigned long long) <= kScratchCounterBytes, "the frame counters fit their part of the scratch");
constexpr size_t kScratchTiles = (size_t)1 << 20;              // tile costs + tile order: at most 2^20 entries together
constexpr size_t kScratchBytes = (size_t)32 << 20;             // in all
constexpr size_t kScratchSpillInts = (kScratchBytes - kScratchCounterBytes - kScratchTiles * sizeof(uint32_t)) / sizeof(int);
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
void *g_scratch_ptr[kMaxDevices];
std::mutex g_scratch_mutex;

// address of `scratch` on `device`, which must be the current device; looked up on the first call per device
int scratch_on_device(int device, unsigned char **out) {
    if (device < 0 || device >= kMaxDevices) return fail(RT_E_UNSUPPORTED, "device %d: only devices 0..%d have a scratch area", device, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(RT_E_DEVICE, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
        g_scratch_ptr[device] = ptr;
    }
    *out = static_cast<unsigned char *>(g_scratch_ptr[device]);
    return RT_OK;
}
}  // namespace

struct rt_scene {
    int device = 0;
    DScene d{};
    dev_buf<DNode> nodes;
    dev_buf<DNode4> nodes4;
    dev_buf<DLeaf> leaves;
    dev_buf<DPrim> prims;
    dev_buf<DItem> items;
    dev_buf<DMat> mats;
    dev_buf<DTex> texs;
    dev_buf<double> ranfloat;
    dev_buf<uint8_t> perm;
    dev_buf<unsigned long long> counter;
    std::vector<uint32_t> material_of_prim;  // by description index
    uint32_t n_prims_desc = 0, n_textures = 0;
    dev_buf<int> spill;                      // stack entries beyond the LDS part, per render lane
    size_t spill_ints = 0;
    dev_buf<uint32_t> tiles;                 // per tile: measured cost, then the render order
    size_t tiles_capacity = 0;
    // the frame whose tile order `tiles` holds: a frame of the same view and size reuses it
    // (and the stream that wrote it: a frame queued on another stream could overtake the sort)
    struct order_key { rt_camera camera; uint32_t nx, ny, max_depth, tile_shift; double t_min; void *stream; } order_of{};
    bool order_valid = false;
    // static scenes whose sphere boxes carry the far slack: frames whose camera lies within
    // far_free_radius of far_free_centre render without the per-segment slack (0: never)
    double far_free_centre[3] = {0, 0, 0}, far_free_radius = 0;
    size_t lds_bytes = 0;                    // stacks of the batch kernels
    // shape of the render kernel for this scene (see k_render)
    bool render_lds_tree = false;
    bool render_defer = false;               // the kernel variant that postpones leaf-box tests
    bool render_wide = false;                // ... walking the four-wide tree
    int render_waves = kWavesMin;            // per SIMD
    uint32_t render_lnodes = 0, render_lleaves = 0, render_stack_cap = 0;
    size_t render_lds_bytes = 0;
    uint32_t render_sums_per_wave = 0;        // bytes of LDS each wave has for the sums of its open tiles
    uint32_t n_nodes_dev = 0, n_prims_dev = 0;
    int n_cus = 256;
};

namespace {

struct flattener {
    const rt_scene_desc *desc;
    std::vector<DNode> nodes;
    std::vector<DLeaf> leaf_boxes;   // per leaf: the reference's box, tested in double
    std::vector<DPrim> prims;        // two entries per leaf (the second unused when the leaf holds one sphere)
    uint32_t max_depth = 0;
    size_t n_moving_leaves = 0;      // leaves that hold a moving sphere
    const char *error = nullptr;

    // fills a primitive entry; returns its leaf-reference bits for slot `second`
    int32_t fill_prim(DPrim &d, int32_t id, bool second, bool &safe) const {
        const rt_prim_desc &p = desc->prims[id];
        memset(&d, 0, sizeof d);
        for (int k = 0; k < 3; ++k) { d.c0[k] = p.centre0[k]; d.dc[k] = p.centre1[k] - p.centre0[k]; }
        d.r = p.radius; d.t0 = p.time0; d.t1 = p.time1;
        d.mat = p.material;
        d.orig = id;
        const bool moving = p.kind == RT_PRIM_MOVING_SPHERE;
        if (!moving) {
            d.kind = PK_STATIC;
            for (int k = 0; k < 3; ++k) d.dc[k] = 0;
        } else {
            d.kind = (p.time0 == 0.0 && p.time1 == 1.0) ? PK_MOVING_UNIT : PK_MOVING;
        }
        // a sphere that never moves stays inside the box the reference gives it
        safe = !moving;
        return moving ? (second ? LEAF_MOVING1 : LEAF_MOVING0) : 0;
    }

    // Fills leaf number `leaf` -- its box and entries 2 leaf, 2 leaf + 1 of the primitive table, which exist
    // already -- with one sphere (prim_b == prim_a) or two; `box` is the box the reference tests before it
    // tries them (min x, max x, min y, ...; unbounded when it tries them unconditionally).  Returns the
    // leaf's reference.
    int32_t fill_leaf(size_t leaf, int32_t prim_a, int32_t prim_b, const double *box, bool &safe) {
        bool sa = true, sb = true;
        int32_t ref = LEAF_BIT | (int32_t)leaf;
        memcpy(leaf_boxes[leaf].box, box, sizeof leaf_boxes[leaf].box);
        ref |= fill_prim(prims[2 * leaf], prim_a, false, sa);
        if (prim_b != prim_a) ref |= LEAF_TWO | fill_prim(prims[2 * leaf + 1], prim_b, true, sb);
        else { DPrim &unused = prims[2 * leaf + 1]; memset(&unused, 0, sizeof unused); unused.orig = -1; }
        safe = sa && sb;
        if (safe) ref |= LEAF_SAFE;
        return ref;
    }

    // appends a leaf (see fill_leaf)
    int32_t leaf_of(int32_t prim_a, int32_t prim_b, const double *box, bool &safe) {
        if (leaf_boxes.size() >= (size_t)LEAF_INDEX_MASK) { error = "more than 2^26 leaves"; return REF_NONE; }
        const size_t leaf = leaf_boxes.size();
        leaf_boxes.emplace_back();
        prims.resize(prims.size() + 2);
        const int32_t ref = fill_leaf(leaf, prim_a, prim_b, box, safe);
        if (!safe) ++n_moving_leaves;
        return ref;
    }

    // a child's box in the node: single precision, rounded outward
    static void set_child_box(DNode &n, int c, const double *box) {
        for (int k = 0; k < 6; ++k) n.bx[c][k] = (k & 1) ? plane_up(box[k]) : plane_down(box[k]);
    }

    // returns the device reference (node index or leaf) built for description node `id`
    int32_t build(int32_t id, uint32_t depth, bool &safe) {
        if (error) return REF_NONE;
        if (depth > 64) { error = "bvh deeper than 64 levels (or cyclic)"; return REF_NONE; }
        if (nodes.size() > (1u << 26)) { error = "bvh larger than 2^26 nodes"; return REF_NONE; }
        const rt_node_desc nd = desc->nodes[id];
        const int32_t ref[2] = {nd.left, nd.right};
        for (int c = 0; c < 2; ++c) {
            if (ref[c] >= 0 ? (uint32_t)ref[c] >= desc->n_nodes : (uint32_t)~ref[c] >= desc->n_prims) {
                error = "node child out of range";
                return REF_NONE;
            }
        }
        double own[6];
        for (int a = 0; a < 3; ++a) { own[2 * a] = nd.box_min[a]; own[2 * a + 1] = nd.box_max[a]; }
        if (ref[0] < 0 && ref[1] < 0) return leaf_of(~ref[0], ~ref[1], own, safe);   // bvh_node.cpp:31-37
        // one object that is itself a subtree, as both children (bvh_node.cpp:31-34): trying it twice finds the
        // same hit twice, and this node's box is the subtree's own
        if (ref[0] == ref[1]) return build(ref[0], depth + 1, safe);

        if (depth + 1 > max_depth) max_depth = depth + 1;
        const int me = (int)nodes.size();
        nodes.emplace_back();
        DNode out;
        memset(&out, 0, sizeof out);
        bool safe_c[2] = {true, true};
        for (int c = 0; c < 2; ++c) {
            if (ref[c] >= 0) {
                const rt_node_desc &ch = desc->nodes[ref[c]];
                double cb[6];
                for (int a = 0; a < 3; ++a) { cb[2 * a] = ch.box_min[a]; cb[2 * a + 1] = ch.box_max[a]; }
                set_child_box(out, c, cb);
                out.child[c] = build(ref[c], depth + 1, safe_c[c]);
                if (error) return REF_NONE;
            } else {
                // a primitive beside a subtree: the reference tries it whenever this node's box is hit
                set_child_box(out, c, own);
                out.child[c] = leaf_of(~ref[c], ~ref[c], own, safe_c[c]);
            }
        }
        if (safe_c[0]) out.flags |= NF_SAFE0;
        if (safe_c[1]) out.flags |= NF_SAFE1;
        safe = safe_c[0] && safe_c[1];
        nodes[me] = out;
        return me;
    }

    // ---- re-clustering ------------------------------------------------------------------
    // Which leaves a ray reaches in the reference's tree depends only on the leaves' own
    // boxes: a node's box is the exact union (fmin/fmax, aabb.cpp:59-64) of its children's,
    // IEEE rounding is monotone, so whenever the slab test of a leaf's box passes, the test
    // of every box that contains it passes too (aabb.cpp:8-57 with the same ray and
    // interval).  "All ancestors' boxes hit" is therefore equivalent to "the leaf's box is
    // hit", and ANY tree whose inner boxes are exact unions of the reference's leaf boxes
    // reaches exactly the reference's leaves.  (The device keeps those unions in single
    // precision, rounded outward, with a conservative test, and repeats the reference's exact
    // test on each leaf's own box: see rt_device.hpp.)  The reference's random-axis median split
    // costs ~19 node visits per ray on the 485-sphere scene; a surface-area-heuristic tree
    // over the same leaves costs ~12.  Primitives keep the reference's left-to-right
    // numbering, which is what decides ties.
    struct leaf_rec { int32_t ref; double box[6]; bool safe; };
    std::vector<leaf_rec> leaves;

    // The leaves under description node `id`, left to right, appended to `leaves` (and to the leaf and
    // primitive tables).  Two walks, so that large subtrees can be shared out between threads: the first
    // counts the leaves under every node, the second fills them in at the places the counts assign.
    static constexpr uint32_t kParallelLeaves = 1u << 15;   // a subtree with this many leaves is worth a thread
    static constexpr uint32_t kParallelCollectDepth = 5;
    std::vector<uint32_t> leaves_under;   // by description node (filled by count_leaves)

    uint32_t count_leaves(int32_t id, uint32_t depth, const char *&problem) {
        if (depth > 64) { problem = "bvh deeper than 64 levels (or cyclic)"; return 0; }
        const rt_node_desc nd = desc->nodes[id];
        const int32_t ref[2] = {nd.left, nd.right};
        for (int c = 0; c < 2; ++c)
            if (ref[c] >= 0 ? (uint32_t)ref[c] >= desc->n_nodes : (uint32_t)~ref[c] >= desc->n_prims) { problem = "node child out of range"; return 0; }
        uint64_t n = 0;
        if (ref[0] < 0 && ref[1] < 0) n = 1;
        else if (ref[0] >= 0 && ref[1] == ref[0]) n = count_leaves(ref[0], depth + 1, problem);   // one object as both children (bvh_node.cpp:31-34): its leaves once
        else if (ref[0] >= 0 && ref[1] >= 0 && depth < kParallelCollectDepth && desc->n_nodes >= 2 * kParallelLeaves) {
            uint32_t left = 0;
            const char *left_problem = nullptr;
            bool started = false;
            std::thread other;
            try { other = std::thread([&] { left = count_leaves(ref[0], depth + 1, left_problem); }); started = true; } catch (const std::system_error &) {}
            const uint32_t right = count_leaves(ref[1], depth + 1, problem);
            if (started) other.join(); else left = count_leaves(ref[0], depth + 1, left_problem);
            if (left_problem) problem = left_problem;
            n = (uint64_t)left + right;
        } else {
            for (int c = 0; c < 2; ++c) n += ref[c] >= 0 ? count_leaves(ref[c], depth + 1, problem) : 1u;
        }
        if (n >= (uint64_t)LEAF_INDEX_MASK) { problem = "more than 2^26 leaves"; n = 0; }
        leaves_under[(size_t)id] = (uint32_t)n;
        return (uint32_t)n;
    }

    // fills leaves[at, at + leaves_under[id]) and the tables' entries from first_leaf + at on; returns how many held a moving sphere
    size_t fill_leaves(int32_t id, size_t at, size_t first_leaf) {
        const rt_node_desc nd = desc->nodes[id];
        const int32_t ref[2] = {nd.left, nd.right};
        leaf_rec lr;
        for (int a = 0; a < 3; ++a) { lr.box[2 * a] = nd.box_min[a]; lr.box[2 * a + 1] = nd.box_max[a]; }
        if (ref[0] < 0 && ref[1] < 0) {
            lr.ref = fill_leaf(first_leaf + at, ~ref[0], ~ref[1], lr.box, lr.safe);
            leaves[at] = lr;
            return lr.safe ? 0 : 1;
        }
        if (ref[0] >= 0 && ref[1] == ref[0]) return fill_leaves(ref[0], at, first_leaf);
        const size_t n_left = ref[0] >= 0 ? leaves_under[(size_t)ref[0]] : 1;
        auto child = [&](int c, size_t where) -> size_t {
            if (ref[c] >= 0) return fill_leaves(ref[c], where, first_leaf);
            leaf_rec one = lr;   // a primitive beside a subtree is tested whenever this node's box is hit
            one.ref = fill_leaf(first_leaf + where, ~ref[c], ~ref[c], one.box, one.safe);
            leaves[where] = one;
            return one.safe ? 0 : 1;
        };
        if (ref[0] >= 0 && ref[1] >= 0 && n_left >= kParallelLeaves && leaves_under[(size_t)ref[1]] >= kParallelLeaves) {
            size_t moving_left = 0;
            bool started = false;
            std::thread other;
            try { other = std::thread([&] { moving_left = child(0, at); }); started = true; } catch (const std::system_error &) {}
            const size_t moving_right = child(1, at + n_left);
            if (started) other.join(); else moving_left = child(0, at);
            return moving_left + moving_right;
        }
        return child(0, at) + child(1, at + n_left);
    }

    void collect(int32_t id) {
        if (error) return;
        leaves_under.assign(desc->n_nodes, 0);
        const char *problem = nullptr;
        const uint32_t n = count_leaves(id, 0, problem);
        if (problem) { error = problem; return; }
        const size_t first_leaf = leaf_boxes.size();
        if (first_leaf + n >= (size_t)LEAF_INDEX_MASK) { error = "more than 2^26 leaves"; return; }
        leaves.assign(n, leaf_rec{});
        leaf_boxes.resize(first_leaf + n);
        prims.resize(2 * (first_leaf + n));
        n_moving_leaves += fill_leaves(id, 0, first_leaf);
    }

    static double half_area(const double *b) {
        const double dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
        return dx * dy + dy * dz + dz * dx;
    }
    static void grow(double *b, const double *o) {
        for (int a = 0; a < 3; ++a) { b[2 * a] = std::fmin(b[2 * a], o[2 * a]); b[2 * a + 1] = std::fmax(b[2 * a + 1], o[2 * a + 1]); }
    }
    static void empty_box(double *b) { for (int a = 0; a < 3; ++a) { b[2 * a] = HUGE_VAL; b[2 * a + 1] = -HUGE_VAL; } }

    // binned surface-area heuristic over leaves[first, first+count); returns the split position.
    // Large ranges are binned by several threads, each over a slice of the range: bin boxes and
    // counts are unions and sums, which come out the same in any order.
    static constexpr int BINS = 32;
    struct bin_set {
        double box[3][BINS][6];
        size_t n[3][BINS];
        void clear() {
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < BINS; ++k) { empty_box(box[a][k]); n[a][k] = 0; }
        }
    };
    static constexpr size_t kParallelRange = (size_t)1 << 16;   // leaves in a range worth more than one thread
    static constexpr int kMaxThreads = 16;

    static int bin_of(const double *b, int a, double lo, double ext) {
        int k = (int)((0.5 * (b[2 * a] + b[2 * a + 1]) - lo) / ext * BINS);
        return k < 0 ? 0 : (k >= BINS ? BINS - 1 : k);
    }

    // runs f(slice_begin, slice_end, slice_number) over [first, first+count) cut into `slices` pieces
    template <class F> static void for_slices(size_t first, size_t count, int slices, F f) {
        std::vector<std::future<void>> jobs;
        for (int t = 1; t < slices; ++t)
            jobs.push_back(std::async(std::launch::async, f, first + count * (size_t)t / slices, first + count * (size_t)(t + 1) / slices, t));
        f(first, first + count / slices, 0);
        for (auto &j : jobs) j.get();
    }

    size_t split(size_t first, size_t count, int threads) {
        const int slices = count >= kParallelRange ? std::max(1, std::min(threads, kMaxThreads)) : 1;
        double cb[6];
        empty_box(cb);
        {
            std::vector<std::array<double, 6>> part((size_t)slices);
            for_slices(first, count, slices, [&](size_t b0, size_t b1, int t) {
                double *c = part[(size_t)t].data();
                empty_box(c);
                for (size_t i = b0; i < b1; ++i) {
                    const double *b = leaves[i].box;
                    for (int a = 0; a < 3; ++a) { const double m = 0.5 * (b[2 * a] + b[2 * a + 1]); c[2 * a] = std::fmin(c[2 * a], m); c[2 * a + 1] = std::fmax(c[2 * a + 1], m); }
                }
            });
            for (const auto &c : part) grow(cb, c.data());
        }
        bool usable[3];
        for (int a = 0; a < 3; ++a) { const double ext = cb[2 * a + 1] - cb[2 * a]; usable[a] = ext > 0 && std::isfinite(ext); }
        std::vector<bin_set> part((size_t)slices);
        for_slices(first, count, slices, [&](size_t b0, size_t b1, int t) {
            bin_set &bs = part[(size_t)t];
            bs.clear();
            for (size_t i = b0; i < b1; ++i) {
                const double *b = leaves[i].box;
                for (int a = 0; a < 3; ++a) {
                    if (!usable[a]) continue;
                    const int k = bin_of(b, a, cb[2 * a], cb[2 * a + 1] - cb[2 * a]);
                    grow(bs.box[a][k], b);
                    bs.n[a][k]++;
                }
            }
        });
        bin_set &bins = part[0];
        for (int t = 1; t < slices; ++t)
            for (int a = 0; a < 3; ++a)
                for (int k = 0; k < BINS; ++k) { if (part[(size_t)t].n[a][k]) grow(bins.box[a][k], part[(size_t)t].box[a][k]); bins.n[a][k] += part[(size_t)t].n[a][k]; }
        double best_cost = HUGE_VAL;
        int best_axis = -1, best_bin = 0;
        for (int a = 0; a < 3; ++a) {
            if (!usable[a]) continue;
            const auto &bb = bins.box[a];
            const auto &bn = bins.n[a];
            double ra[BINS];
            size_t rn[BINS];
            double acc[6];
            empty_box(acc);
            size_t n = 0;
            for (int k = BINS - 1; k > 0; --k) { if (bn[k]) grow(acc, bb[k]); n += bn[k]; ra[k] = n ? half_area(acc) : 0; rn[k] = n; }
            empty_box(acc);
            n = 0;
            for (int k = 0; k < BINS - 1; ++k) {
                if (bn[k]) grow(acc, bb[k]);
                n += bn[k];
                if (n == 0 || rn[k + 1] == 0) continue;
                const double cost = half_area(acc) * (double)n + ra[k + 1] * (double)rn[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
            }
        }
        if (best_axis < 0) return first + count / 2;   // coincident centres: any split will do
        const int a = best_axis;
        const double lo = cb[2 * a], ext = cb[2 * a + 1] - lo;
        auto mid = std::partition(leaves.begin() + (long)first, leaves.begin() + (long)(first + count),
                                  [&](const leaf_rec &l) { return bin_of(l.box, a, lo, ext) <= best_bin; });
        size_t pos = (size_t)(mid - leaves.begin());
        if (pos == first || pos == first + count) pos = first + count / 2;
        return pos;
    }

    // Builds the node over leaves[first, first+count) (count >= 2) as nodes[me]; its subtree takes
    // nodes[me, me + count - 1), numbered in pre-order, so the two halves of a large range can be
    // built by different threads.  out_box receives the node's exact union box, depth_out the
    // depth of the deepest node below (this node = `depth` + 1).
    int32_t cluster(size_t first, size_t count, uint32_t depth, size_t me, int threads, double *out_box, bool &safe, uint32_t &depth_out) {
        depth_out = depth + 1;
        // a tree this deep means the heuristic keeps peeling single leaves: fall back to halving
        const size_t pos = depth > 48 ? first + count / 2 : split(first, count, threads);
        DNode out;
        memset(&out, 0, sizeof out);
        bool safe_c[2] = {true, true};
        double cb[2][6];   // the children's boxes: exact unions of the reference's leaf boxes
        uint32_t deep[2] = {0, 0};
        const size_t beg[2] = {first, pos}, cnt[2] = {pos - first, first + count - pos};
        const size_t at[2] = {me + 1, me + cnt[0]};   // (a half of one leaf takes no node)
        auto child = [&](int c, int th) {
            if (cnt[c] == 1) {
                out.child[c] = leaves[beg[c]].ref;
                memcpy(cb[c], leaves[beg[c]].box, sizeof cb[c]);
                safe_c[c] = leaves[beg[c]].safe;
            } else {
                out.child[c] = cluster(beg[c], cnt[c], depth + 1, at[c], th, cb[c], safe_c[c], deep[c]);
            }
        };
        if (threads > 1 && std::min(cnt[0], cnt[1]) >= 4096) {
            // threads in proportion to the halves' sizes
            int t0 = (int)std::llround((double)threads * (double)cnt[0] / (double)count);
            t0 = std::max(1, std::min(threads - 1, t0));
            auto other = std::async(std::launch::async, child, 0, t0);
            child(1, threads - t0);
            other.get();
        } else {
            child(0, threads);
            child(1, threads);
        }
        for (int c = 0; c < 2; ++c) set_child_box(out, c, cb[c]);
        if (safe_c[0]) out.flags |= NF_SAFE0;
        if (safe_c[1]) out.flags |= NF_SAFE1;
        safe = safe_c[0] && safe_c[1];
        memcpy(out_box, cb[0], sizeof cb[0]);
        grow(out_box, cb[1]);
        depth_out = std::max(depth_out, std::max(deep[0], deep[1]));
        nodes[me] = out;
        return (int32_t)me;
    }

    // Scenes of static spheres only: every leaf of two spheres becomes one more node over two
    // one-sphere references, each with the sphere's own box.  The reference's leaves pair
    // neighbours along a random axis (bvh_node.cpp:9-41), so a leaf's box is mostly empty: of the rays
    // that reach one, a few per cent meet the line of either sphere.  With its own box in the way
    // a sphere is tried by about as many rays as hit it.  A ray the sphere test accepts passes
    // within the radius of the centre up to the rounding of the test, which the padding below and
    // the conservative node test's margins exceed by orders of magnitude, so no candidate of the
    // reference's is lost; the reference's test of the leaf's box is applied as before (to the winner,
    // or in the exact walk, once per sphere reached).
    void split_pairs(std::vector<DItem> &items) {
        const size_t n = nodes.size();
        bool any = false;
        auto split = [&](int32_t ref) -> int32_t {
            const int32_t leaf = ref & LEAF_INDEX_MASK;
            DNode nd;
            memset(&nd, 0, sizeof nd);
            double b[6];
            for (int k = 0; k < 2; ++k) {
                sphere_box(2 * leaf + k, b);
                set_child_box(nd, k, b);
                nd.child[k] = LEAF_BIT | LEAF_SAFE | leaf | (k ? LEAF_SECOND : 0);
            }
            nd.flags = NF_SAFE0 | NF_SAFE1;
            nodes.push_back(nd);
            any = true;
            return (int32_t)nodes.size() - 1;
        };
        for (size_t i = 0; i < n; ++i)
            for (int c = 0; c < 2; ++c) {
                const int32_t ref = nodes[i].child[c];
                if (ref < 0 && (ref & LEAF_TWO)) { const int32_t nn = split(ref); nodes[i].child[c] = nn; }
            }
        for (DItem &it : items)
            if (it.has_box && it.ref < 0 && (it.ref & LEAF_TWO)) it.ref = split(it.ref);
        if (any) ++max_depth;
    }

    // ---- four-wide tree ------------------------------------------------------------------
    // Folds the binary tree into nodes of up to four children: a node starts with the two children of
    // its binary node and, while it has room, replaces the child of largest surface area that is
    // itself a node by that node's two children.  The boxes are the binary tree's own child boxes
    // (single precision, already moved and rounded outward): each still contains every leaf box
    // below it, which is all the conservative walk asks of an inner box.
    std::vector<DNode4> nodes4;
    uint32_t depth4 = 0;
    static float area4(const float *b) {
        const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
        const float a = dx * dy + dy * dz + dz * dx;
        return a >= 0 ? a : 0;   // (NaN or inverted: no preference)
    }
    int32_t collapse(int32_t b, uint32_t depth) {
        const size_t me = nodes4.size();
        nodes4.emplace_back();
        if (depth + 1 > depth4) depth4 = depth + 1;
        struct slot { int32_t ref; float box[6]; } s[4];
        int n = 2;
        for (int c = 0; c < 2; ++c) { s[c].ref = nodes[(size_t)b].child[c]; memcpy(s[c].box, nodes[(size_t)b].bx[c], sizeof s[c].box); }
        while (n < 4) {
            int best = -1;
            float best_area = -1;
            for (int i = 0; i < n; ++i)
                if (s[i].ref >= 0 && area4(s[i].box) > best_area) { best = i; best_area = area4(s[i].box); }
            if (best < 0) break;
            const DNode &c = nodes[(size_t)s[best].ref];
            s[best].ref = c.child[0]; memcpy(s[best].box, c.bx[0], sizeof s[best].box);
            s[n].ref = c.child[1]; memcpy(s[n].box, c.bx[1], sizeof s[n].box);
            ++n;
        }
        DNode4 out;
        memset(&out, 0, sizeof out);
        for (int i = 0; i < 4; ++i) {
            out.child[i] = REF_NONE;
            for (int k = 0; k < 6; ++k) out.pl[k][i] = (k & 1) ? -HUGE_VALF : HUGE_VALF;   // an inverted box
        }
        for (int i = 0; i < n; ++i) {
            for (int k = 0; k < 6; ++k) out.pl[k][i] = s[i].box[k];
            out.child[i] = s[i].ref >= 0 ? collapse(s[i].ref, depth + 1) : s[i].ref;
        }
        nodes4[me] = out;
        return (int32_t)me;
    }
    void collapse4(std::vector<DItem> &items) {
        nodes4.clear();
        nodes4.reserve(nodes.size() / 2 + 1);
        depth4 = 0;
        for (DItem &it : items) it.ref4 = (it.has_box && it.ref >= 0) ? collapse(it.ref, 0) : it.ref;
    }

    // Renumbers the nodes so that the k nearest to the roots, breadth-first, come first (the
    // part of a large tree the render kernel keeps in LDS); the others keep their order.
    void stage_top_levels(size_t k, std::vector<DItem> &items) {
        const size_t n = nodes.size();
        if (k > n) k = n;
        std::vector<int32_t> renum(n, -1);
        std::vector<int32_t> order;
        order.reserve(k);
        for (const DItem &it : items)
            if (it.ref >= 0 && order.size() < k && renum[it.ref] < 0) { renum[it.ref] = (int32_t)order.size(); order.push_back(it.ref); }
        for (size_t head = 0; head < order.size() && order.size() < k; ++head)
            for (int c = 0; c < 2; ++c) {
                const int32_t ch = nodes[order[head]].child[c];
                if (ch >= 0 && order.size() < k && renum[ch] < 0) { renum[ch] = (int32_t)order.size(); order.push_back(ch); }
            }
        int32_t next = (int32_t)order.size();
        for (size_t i = 0; i < n; ++i)
            if (renum[i] < 0) renum[i] = next++;
        std::vector<DNode> moved(n);
        for (size_t i = 0; i < n; ++i) {
            DNode nd = nodes[i];
            for (int c = 0; c < 2; ++c)
                if (nd.child[c] >= 0) nd.child[c] = renum[nd.child[c]];
            moved[renum[i]] = nd;
        }
        nodes.swap(moved);
        for (DItem &it : items)
            if (it.ref >= 0) it.ref = renum[it.ref];
    }

    // the box a one-sphere reference gets in a scene of static spheres (see split_pairs, sphere_pad_box)
    void sphere_box(int32_t prim, double *b) const {
        const DPrim &p = prims[(size_t)prim];
        sphere_pad_box(p.c0, p.r, far_allowance, b);
    }

    bool static_scene = false;   // no moving sphere: the tree is built over single spheres

    // Static scenes: the largest distance between a ray's origin and a sphere's centre that the
    // padding of the spheres' own boxes covers (0: none is built in, every segment computes its slack).
    double far_allowance = 0;
    double far_pad(double radius) const { return far_slack_pad(radius, far_allowance); }

    // Large scenes of static spheres: the tree over the single spheres is built on the GPU (bvh_device.hpp).
    // The host then only lists the spheres (device_refs: one-sphere leaf references) and the box around
    // their centres; the nodes are numbered from 0 on the device and never exist on the host.
    bool device_build = false;
    std::vector<int32_t> device_refs;
    double device_lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, device_hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};

    // the whole bvh item rooted at description node `id`, re-clustered
    int32_t build_clustered(int32_t id, bool &safe) {
        const bool verbose = getenv("RTIOW_VERBOSE") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            const auto t1 = std::chrono::steady_clock::now();
            if (verbose) fprintf(stderr, "rtiow:   %s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
            t0 = t1;
        };
        leaves.clear();
        collect(id);
        lap("reference leaves collected");
        if (error) return REF_NONE;
        if (static_scene && device_build && leaves.size() >= 2 && nodes.empty()) {
            device_refs.reserve(2 * leaves.size());
            for (const leaf_rec &lr : leaves) {
                const int32_t leaf = lr.ref & LEAF_INDEX_MASK;
                for (int k = 0; k < ((lr.ref & LEAF_TWO) ? 2 : 1); ++k) {
                    device_refs.push_back(LEAF_BIT | LEAF_SAFE | leaf | (k ? LEAF_SECOND : 0));
                    const double *c = prims[2 * (size_t)leaf + (size_t)k].c0;
                    for (int a = 0; a < 3; ++a) { device_lo[a] = std::fmin(device_lo[a], c[a]); device_hi[a] = std::fmax(device_hi[a], c[a]); }
                }
            }
            leaves.clear();
            lap("spheres listed for the device build");
            safe = true;
            return 0;   // the device numbers its root 0
        }
        device_build = false;   // (not this item: the host clusters it)
        if (static_scene) {
            // cluster the spheres themselves rather than the reference's pairs of them: the pairs are
            // neighbours along a random axis, and a tree over their mostly empty boxes stays loose
            std::vector<leaf_rec> singles;
            singles.reserve(2 * leaves.size());
            for (const leaf_rec &lr : leaves) {
                const int32_t leaf = lr.ref & LEAF_INDEX_MASK;
                for (int k = 0; k < ((lr.ref & LEAF_TWO) ? 2 : 1); ++k) {
                    leaf_rec one;
                    one.ref = LEAF_BIT | LEAF_SAFE | leaf | (k ? LEAF_SECOND : 0);
                    sphere_box(2 * leaf + k, one.box);
                    one.safe = true;
                    singles.push_back(one);
                }
            }
            leaves.swap(singles);
            lap("single-sphere boxes");
        }
        if (leaves.size() == 1) { safe = leaves[0].safe; return leaves[0].ref; }
        double box[6];
        const size_t me = nodes.size();
        if (me + leaves.size() - 1 > ((size_t)1 << 26)) { error = "bvh larger than 2^26 nodes"; return REF_NONE; }
        nodes.resize(me + leaves.size() - 1);   // a binary tree over n leaves has n - 1 nodes
        int threads = std::max(1u, std::min<unsigned>(kMaxThreads, std::thread::hardware_concurrency()));
        if (const char *e = getenv("RTIOW_BUILD_THREADS")) threads = std::max(1, std::min(kMaxThreads, atoi(e)));
        uint32_t deepest = 0;
        const int32_t root = cluster(0, leaves.size(), 0, me, threads, box, safe, deepest);
        max_depth = std::max(max_depth, deepest);
        lap("surface-area clustering");
        return root;
    }
};

int check_desc(const rt_scene_desc *d) {
    if (!d) return fail(RT_E_INVALID, "null scene description");
    if ((d->n_textures && !d->textures) || (d->n_materials && !d->materials) || (d->n_prims && !d->prims) ||
        (d->n_nodes && !d->nodes) || (d->n_items && !d->items))
        return fail(RT_E_INVALID, "scene description has a null table with a non-zero count");
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const rt_texture_desc &t = d->textures[i];
        if (t.kind > RT_TEX_PERLIN) return fail(RT_E_INVALID, "texture %u: unknown kind %u", i, t.kind);
        if (t.kind == RT_TEX_CHEQUER && (t.even >= i || t.odd >= i))
            return fail(RT_E_INVALID, "texture %u: chequer children must index earlier textures", i);
        if (t.kind == RT_TEX_PERLIN && (!d->perlin_ranfloat || !d->perlin_perm_x || !d->perlin_perm_y || !d->perlin_perm_z))
            return fail(RT_E_INVALID, "texture %u: perlin texture without perlin tables", i);
    }
    for (uint32_t i = 0; i < d->n_materials; ++i) {
        const rt_material_desc &m = d->materials[i];
        if (m.kind > RT_MAT_DIELECTRIC) return fail(RT_E_INVALID, "material %u: unknown kind %u", i, m.kind);
        if (m.kind == RT_MAT_LAMBERTIAN && m.texture >= d->n_textures)
            return fail(RT_E_INVALID, "material %u: texture %u out of range", i, m.texture);
    }
    for (uint32_t i = 0; i < d->n_prims; ++i) {
        const rt_prim_desc &p = d->prims[i];
        if (p.kind > RT_PRIM_MOVING_SPHERE) return fail(RT_E_INVALID, "primitive %u: unknown kind %u", i, p.kind);
        if (p.material >= d->n_materials) return fail(RT_E_INVALID, "primitive %u: material %u out of range", i, p.material);
    }
    for (uint32_t i = 0; i < d->n_items; ++i) {
        const rt_item_desc &it = d->items[i];
        if (it.index < 0 || (uint32_t)it.index >= (it.is_bvh ? d->n_nodes : d->n_prims))
            return fail(RT_E_INVALID, "item %u: index %d out of range", i, it.index);
    }
    return RT_OK;
}

}  // namespace

extern "C" {

const char *rt_last_error(void) { return g_err; }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int scene_create(const rt_scene_desc *desc, int device, rt_scene **out);

int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out) {
    if (!out) return fail(RT_E_INVALID, "null output handle");
    *out = nullptr;
    try {
        return scene_create(desc, device, out);
    } catch (const std::bad_alloc &) {   // the flattened tree is built in host vectors
        return fail(RT_E_INVALID, "rt_scene_create: out of host memory");
    } catch (const std::exception &e) {  // e.g. std::system_error when the build's helper threads cannot be started
        return fail(RT_E_DEVICE, "rt_scene_create: %s", e.what());
    } catch (...) {
        return fail(RT_E_DEVICE, "rt_scene_create: unknown failure while laying out the tree");
    }
}

static int scene_create(const rt_scene_desc *desc, int device, rt_scene **out) {
    if (int rc = check_desc(desc)) return rc;
    if (device < -1) return fail(RT_E_INVALID, "device %d: want an index, or -1 for the current device", device);
    const bool verbose = getenv("RTIOW_VERBOSE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto seconds_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };

    flattener fl;
    // RTIOW_TREE=reference keeps the reference's own topology on the device (for A/B and for the
    // test that both layouts give bit-identical results); the default re-clusters its leaves.
    const char *tree_env = getenv("RTIOW_TREE");
    const bool reference_layout = tree_env && strcmp(tree_env, "reference") == 0;
    fl.desc = desc;
    // RTIOW_STATIC_VARIANT=0 treats a scene without moving spheres like any other (tests, A/B)
    const char *static_env = getenv("RTIOW_STATIC_VARIANT");
    fl.static_scene = !(static_env && atoi(static_env) == 0);
    for (uint32_t i = 0; i < desc->n_prims; ++i)
        if (desc->prims[i].kind == RT_PRIM_MOVING_SPHERE) fl.static_scene = false;
    // The world's extent (centre and half-diagonal of the box around all spheres).  In a scene of
    // static spheres every ray starts on a sphere or at the camera: with the camera within three
    // half-diagonals of the centre no origin is farther than four from any sphere's centre, and when
    // the slack of the sphere test at that distance is a small part of every radius it is built
    // into the spheres' boxes, so that such frames need no per-segment slack (rt_render checks the
    // camera; rays from anywhere else -- rt_trace_batch -- always compute theirs).
    double extent_c[3] = {0, 0, 0}, extent_h = 0;
    if (fl.static_scene && desc->n_prims) {
        double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
        for (uint32_t i = 0; i < desc->n_prims; ++i)
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::fmin(lo[a], desc->prims[i].centre0[a] - std::fabs(desc->prims[i].radius));
                hi[a] = std::fmax(hi[a], desc->prims[i].centre0[a] + std::fabs(desc->prims[i].radius));
            }
        double h2 = 0;
        for (int a = 0; a < 3; ++a) { extent_c[a] = 0.5 * (lo[a] + hi[a]); const double e = 0.5 * (hi[a] - lo[a]); h2 += e * e; }
        extent_h = std::sqrt(h2) * (1 + 0x1p-30);
        bool small = std::isfinite(extent_h) && extent_h > 0 && !getenv("RTIOW_FAR_SLACK_ALWAYS");
        fl.far_allowance = 4 * extent_h;
        for (uint32_t i = 0; i < desc->n_prims && small; ++i) {
            const double r = std::fabs(desc->prims[i].radius);
            small = r > 0 && fl.far_pad(r) <= 0x1p-10 * r;
        }
        if (!small) fl.far_allowance = 0;
    }
    // Where the tree over the spheres of a static scene is built: on the GPU for large scenes with a single
    // bvh (a linear BVH in milliseconds: set-up is most of such a scene's first frame), on the host otherwise
    // (the surface-area clustering: the tighter tree).  RTIOW_BUILD=device|host decides for any size.
    {
        const char *build_env = getenv("RTIOW_BUILD");
        const char *wide_switch = getenv("RTIOW_WIDE");
        uint32_t n_bvh = 0;
        for (uint32_t i = 0; i < desc->n_items; ++i) n_bvh += desc->items[i].is_bvh ? 1u : 0u;
        const bool wanted = build_env ? strcmp(build_env, "device") == 0 : desc->n_prims >= (1u << 17);
        fl.device_build = wanted && fl.static_scene && !reference_layout && n_bvh == 1 && !(wide_switch && atoi(wide_switch) == 1) &&
                          rt_device_count() > 0;
    }
    std::vector<DItem> items;
    for (uint32_t i = 0; i < desc->n_items; ++i) {
        const rt_item_desc &it = desc->items[i];
        DItem di;
        memset(&di, 0, sizeof di);
        bool safe;
        if (it.is_bvh) {
            di.has_box = 1;
            di.ref = reference_layout ? fl.build(it.index, 0, safe) : fl.build_clustered(it.index, safe);
        } else {
            const double unbounded[6] = {-HUGE_VAL, HUGE_VAL, -HUGE_VAL, HUGE_VAL, -HUGE_VAL, HUGE_VAL};   // object_list.cpp:18 tries it directly
            di.ref = fl.leaf_of(it.index, it.index, unbounded, safe);
        }
        if (fl.error) return fail(RT_E_UNSUPPORTED, "%s", fl.error);
        di.ref4 = di.ref;
        items.push_back(di);
    }

    const double t_layout = seconds_since(t_start);
    if (verbose) fprintf(stderr, "rtiow: tree laid out on the host in %.3f s (%zu nodes)\n", t_layout, fl.nodes.size());
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (verbose) fprintf(stderr, "rtiow:   %s %.3f s\n", what, seconds_since(t_lap));
        t_lap = std::chrono::steady_clock::now();
    };
    if (rt_device_count() <= 0) return fail(RT_E_DEVICE, "no HIP device available");
    if (int rc = use_device(device, &device)) return rc;

    std::unique_ptr<rt_scene> owner(new rt_scene);   // freed on every early return
    rt_scene *s = owner.get();
    s->device = device;
    // Scenes without a moving sphere run the kernel variant that postpones the leaves' box tests to
    // the winning leaf and skips behind the best hit without consulting flags (k_render's DEFER), on
    // a tree whose leaves are single spheres.  RTIOW_STATIC_VARIANT=0 keeps the general variant and
    // the reference's leaves (tests, A/B).
    const bool on_device = fl.device_build && fl.device_refs.size() >= 2;
    size_t n_nodes = fl.nodes.size();
    uint32_t staged_on_device = 0;
    if (on_device) {
        // the spheres go up first; the tree is built over them where they are.  The table's first entries are
        // copies of the tree's top levels, breadth-first: what the render kernel keeps in LDS beside a
        // twelve-entry stack per lane (a large tree is deeper than that).
        hipError_t e = s->prims.upload(fl.prims.data(), fl.prims.size());
        const size_t tree_nodes = fl.device_refs.size() - 1;
        const size_t beside = kHeadLds + 4 * kWavesMax * 3072 + (size_t)kSpillStackCap * 256 * kWavesMax * sizeof(int);
        size_t room = std::min<size_t>({tree_nodes, (kLdsBudget - beside - 8) / LNODE_STRIDE, (size_t)lbvh::kStageMax});
        if (const char *f = getenv("RTIOW_RENDER_LDS_NODES")) room = std::min(room, (size_t)std::max(0, atoi(f)));
        staged_on_device = (uint32_t)room;
        n_nodes = staged_on_device + tree_nodes;
        if (e == hipSuccess) e = s->nodes.alloc(n_nodes);
        uint32_t depth = 0;
        if (e == hipSuccess)
            e = lbvh::build(s->prims.p, fl.device_refs.data(), (uint32_t)fl.device_refs.size(), fl.far_allowance, fl.device_lo, fl.device_hi, s->nodes.p, staged_on_device, &depth);
        if (e != hipSuccess) return fail(RT_E_DEVICE, "tree build on the device: %s", hipGetErrorString(e));
        fl.max_depth = std::max(fl.max_depth, depth);
        lap("tree built on the device");
    }
    const bool static_scene = fl.static_scene && n_nodes > 0;
    if (static_scene && !on_device) {
        fl.split_pairs(items);   // (what re-clustering left whole: the reference layout's leaves)
        n_nodes = fl.nodes.size();
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->n_cus = prop.multiProcessorCount;
    }
    // RTIOW_WIDE=1: scenes without moving spheres walk a four-wide tree instead of the binary one.  Half
    // the steps per walk, but a step costs more than two of the binary walk's: measured 3-4 % slower on
    // the 485-sphere scene (tree in LDS) and on 10^6 spheres (tree in device memory), so it is not the
    // default; the test suite holds it to the binary walk's frames bit for bit.
    const char *wide_env = getenv("RTIOW_WIDE");
    const bool wide = static_scene && wide_env && atoi(wide_env) == 1;
    if (wide) fl.collapse4(items);
    // shape of the render kernel: everything in LDS when it fits, else the top levels only
    // (a binary walk holds one stack entry per level; a four-wide one up to three, and its pushes write three entries ahead)
    const uint32_t stack_depth = fl.max_depth + 2, stack_depth4 = 3 * fl.depth4 + 4;
    size_t spill_ints = 0;
    {
        // per wave: the sums of its open tiles, 24 bytes a pixel -- four 4x4 tiles or one 8x8 tile beside a
        // tree that fills LDS, twice that when only the top of the tree is there
        constexpr size_t kSumsLdsTree = 1536, kSumsLargeTree = 3072;
        size_t sums = kSumsLdsTree;
        const size_t leaf_bytes = fl.leaf_boxes.size() * (static_scene ? SLEAF_STRIDE : LLEAF_STRIDE);
        auto whole_tree = [&](int waves) {
            return kHeadLds + 4 * waves * sums + (size_t)stack_depth * 256 * waves * sizeof(int) + ((n_nodes * LNODE_STRIDE + 7) & ~(size_t)7) + leaf_bytes;
        };
        // the four-wide tree whole in LDS, with `cap` stack entries per lane there
        auto whole_tree4 = [&](int waves, uint32_t cap) {
            return kHeadLds + 4 * waves * sums + (size_t)cap * 256 * waves * sizeof(int) + ((fl.nodes4.size() * LNODE4_STRIDE + 15) & ~(size_t)15) + leaf_bytes;
        };
        const char *force_waves = getenv("RTIOW_RENDER_WAVES");   // "3" keeps the 768-thread shape (tests)
        const int waves_fit = whole_tree(kWavesMax) <= kLdsBudget && !(force_waves && atoi(force_waves) == kWavesMin) ? kWavesMax : kWavesMin;
        const size_t whole = whole_tree(waves_fit);
        // RTIOW_RENDER_LDS_NODES=k forces the large-tree shape with at most k nodes staged and
        // RTIOW_RENDER_STACK_CAP=c its LDS stack entries, so that small scenes can exercise
        // the staged walk and the spilling stack in tests.
        const char *force_nodes = getenv("RTIOW_RENDER_LDS_NODES"), *force_cap = getenv("RTIOW_RENDER_STACK_CAP");
        // LDS stack entries of a four-wide walk: its worst case is rarely approached, so twelve entries (or
        // what the tree leaves room for, at least eight) stay in LDS and deeper ones spill
        const int waves4 = force_waves && atoi(force_waves) == kWavesMin ? kWavesMin : kWavesMax;
        uint32_t cap4 = std::min<uint32_t>(stack_depth4, 12);
        if (const char *e = getenv("RTIOW_WIDE_STACK_CAP")) cap4 = std::min<uint32_t>(stack_depth4, (uint32_t)std::max(4, atoi(e)));
        if (force_cap) cap4 = std::min<uint32_t>(cap4, (uint32_t)std::max(4, atoi(force_cap)));
        const bool wide_in_lds = wide && !force_nodes && fl.nodes4.size() <= kWideLdsMaxNodes && fl.leaf_boxes.size() <= kWideLdsMaxLeaves &&
                                 whole_tree4(waves4, cap4) <= kLdsBudget;
        if (wide_in_lds) {
            s->render_wide = true;
            s->render_lds_tree = true;
            s->render_waves = waves4;
            s->render_lnodes = (uint32_t)fl.nodes4.size();
            s->render_lleaves = (uint32_t)fl.leaf_boxes.size();
            s->render_stack_cap = cap4;
            s->render_lds_bytes = whole_tree4(waves4, cap4);
            if (stack_depth4 > cap4) spill_ints = (size_t)(stack_depth4 - cap4) * s->n_cus * 256 * waves4;
        } else if (whole <= kLdsBudget && !force_nodes && !on_device) {
            s->render_lds_tree = true;
            s->render_waves = waves_fit;
            s->render_lnodes = (uint32_t)fl.nodes.size();
            s->render_lleaves = (uint32_t)fl.leaf_boxes.size();
            s->render_stack_cap = stack_depth;
            s->render_lds_bytes = whole;
        } else if (wide) {
            // the four-wide tree in device memory: LDS holds the stacks
            sums = kSumsLargeTree;
            uint32_t cap = std::min<uint32_t>(stack_depth4, 16);
            if (const char *e = getenv("RTIOW_WIDE_STACK_CAP")) cap = std::min<uint32_t>(stack_depth4, (uint32_t)std::max(4, atoi(e)));
            if (force_cap) cap = std::min<uint32_t>(cap, (uint32_t)std::max(4, atoi(force_cap)));
            s->render_wide = true;
            s->render_waves = kWavesMax;
            s->render_lnodes = 0;
            s->render_stack_cap = cap;
            s->render_lds_bytes = kHeadLds + 4 * kWavesMax * sums + (size_t)cap * 256 * kWavesMax * sizeof(int);
            if (stack_depth4 > cap) spill_ints = (size_t)(stack_depth4 - cap) * s->n_cus * 256 * kWavesMax;
        } else {
            sums = kSumsLargeTree;
            uint32_t cap = std::min<uint32_t>(stack_depth, kSpillStackCap);
            if (force_cap) cap = std::min<uint32_t>(cap, (uint32_t)std::max(1, atoi(force_cap)));
            s->render_waves = kWavesMax;
            const size_t base = kHeadLds + 4 * kWavesMax * sums + (size_t)cap * 256 * kWavesMax * sizeof(int);
            size_t k = std::min(fl.nodes.size(), (kLdsBudget - base - 8) / LNODE_STRIDE);
            if (force_nodes) k = std::min(k, (size_t)std::max(0, atoi(force_nodes)));
            // (a tree built on the device has its top levels in front already)
            if (!on_device) fl.stage_top_levels(k, items);
            else k = std::min<size_t>(staged_on_device, (kLdsBudget - base - 8) / LNODE_STRIDE);
            s->render_lnodes = (uint32_t)k;
            s->render_stack_cap = cap;
            s->render_lds_bytes = base + ((k * LNODE_STRIDE + 7) & ~(size_t)7);
            spill_ints = (size_t)(stack_depth - cap) * s->n_cus * 256 * kWavesMax;
        }
        s->render_sums_per_wave = (uint32_t)sums;
    }
    lap("kernel shape, top levels staged");
    s->n_prims_desc = desc->n_prims;
    s->n_textures = desc->n_textures;
    s->material_of_prim.resize(desc->n_prims);
    for (uint32_t i = 0; i < desc->n_prims; ++i) s->material_of_prim[i] = desc->prims[i].material;

    const int table_slices = (desc->n_materials >= (1u << 16) || desc->n_textures >= (1u << 16)) ? std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
    std::vector<DMat> mats(desc->n_materials);
    flattener::for_slices(0, desc->n_materials, table_slices, [&](size_t b0, size_t b1, int) {
        for (size_t i = b0; i < b1; ++i) {
            const rt_material_desc &m = desc->materials[i];
            DMat d;
            memset(&d, 0, sizeof d);
            d.kind = m.kind; d.tex = m.texture; d.fuzz = m.fuzziness; d.ri = m.refractive_index;
            d.inv_ri = 1 / d.ri;
            { const double r0 = (1 - d.ri) / (1 + d.ri); d.r0sq = r0 * r0; }
            for (int k = 0; k < 3; ++k) d.albedo[k] = m.albedo[k];
            mats[i] = d;
        }
    });
    std::vector<DTex> texs(desc->n_textures);
    flattener::for_slices(0, desc->n_textures, table_slices, [&](size_t b0, size_t b1, int) {
        for (size_t i = b0; i < b1; ++i) {
            const rt_texture_desc &t = desc->textures[i];
            DTex d;
            memset(&d, 0, sizeof d);
            d.kind = t.kind; d.even = t.even; d.odd = t.odd;
            for (int k = 0; k < 3; ++k) d.c[k] = t.colour[k];
            texs[i] = d;
        }
    });
    std::vector<double> ranfloat(256, 0.0);
    std::vector<uint8_t> perm(768, 0);
    if (desc->perlin_ranfloat && desc->perlin_perm_x && desc->perlin_perm_y && desc->perlin_perm_z) {
        memcpy(ranfloat.data(), desc->perlin_ranfloat, 256 * sizeof(double));
        memcpy(perm.data(), desc->perlin_perm_x, 256);
        memcpy(perm.data() + 256, desc->perlin_perm_y, 256);
        memcpy(perm.data() + 512, desc->perlin_perm_z, 256);
    }

    lap("material and texture tables");
    {   // the address of rt_render_noalloc's working memory on this device, while set-up may still take its time
        unsigned char *unused;
        if (int rc = scratch_on_device(device, &unused)) return rc;
    }
    hipError_t e = hipSuccess;
    if (e == hipSuccess && !on_device) e = s->nodes.upload(fl.nodes.data(), fl.nodes.size());
    if (e == hipSuccess && s->render_wide) e = s->nodes4.upload(fl.nodes4.data(), fl.nodes4.size());
    if (e == hipSuccess) e = s->leaves.upload(fl.leaf_boxes.data(), fl.leaf_boxes.size());
    if (e == hipSuccess && !on_device) e = s->prims.upload(fl.prims.data(), fl.prims.size());
    if (e == hipSuccess) e = s->items.upload(items.data(), items.size());
    if (e == hipSuccess) e = s->mats.upload(mats.data(), mats.size());
    if (e == hipSuccess) e = s->texs.upload(texs.data(), texs.size());
    if (e == hipSuccess) e = s->ranfloat.upload(ranfloat.data(), ranfloat.size());
    if (e == hipSuccess) e = s->perm.upload(perm.data(), perm.size());
    if (e == hipSuccess) e = s->counter.alloc(kCounters);
    if (e == hipSuccess && spill_ints) e = s->spill.alloc(spill_ints);
    s->spill_ints = spill_ints;
    lap("upload");
    // the kernels with per-lane stacks may ask for more dynamic LDS than the default 64 KiB limit
    const void *kernels[28] = {
        reinterpret_cast<const void *>(&k_trace_batch),
#define RT_SHAPES(D, W)                                                                \
    reinterpret_cast<const void *>(&k_render<float, true, kWavesMax, false, D, W>),    \
    reinterpret_cast<const void *>(&k_render<double, true, kWavesMax, false, D, W>),   \
    reinterpret_cast<const void *>(&k_render<float, true, kWavesMin, false, D, W>),    \
    reinterpret_cast<const void *>(&k_render<double, true, kWavesMin, false, D, W>),   \
    reinterpret_cast<const void *>(&k_render<float, false, kWavesMax, false, D, W>),   \
    reinterpret_cast<const void *>(&k_render<double, false, kWavesMax, false, D, W>),  \
    reinterpret_cast<const void *>(&k_render<float, true, kWavesMax, true, D, W>),     \
    reinterpret_cast<const void *>(&k_render<float, true, kWavesMin, true, D, W>),     \
    reinterpret_cast<const void *>(&k_render<float, false, kWavesMax, true, D, W>)
        RT_SHAPES(false, false), RT_SHAPES(true, false), RT_SHAPES(true, true)
#undef RT_SHAPES
    };
    for (const void *k : kernels)
        if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget);
    if (e != hipSuccess) return fail(RT_E_DEVICE, "scene upload: %s", hipGetErrorString(e));
    lap("kernel attributes");
    s->d.nodes = s->nodes.p; s->d.nodes4 = s->render_wide ? s->nodes4.p : nullptr; s->d.leaves = s->leaves.p; s->d.prims = s->prims.p; s->d.items = s->items.p;
    s->d.mats = s->mats.p; s->d.texs = s->texs.p; s->d.ranfloat = s->ranfloat.p; s->d.perm = s->perm.p;
    s->d.n_items = (uint32_t)items.size();
    s->d.stack_depth = stack_depth;
    // can a sample's radiance leave [0, 1]?  (the sky's never does; attenuations are texture colours and metal albedos)
    s->d.wide_radiance = 0;
    for (uint32_t i = 0; i < desc->n_textures; ++i)
        for (int k = 0; k < 3; ++k)
            if (desc->textures[i].kind == RT_TEX_CONSTANT && !(desc->textures[i].colour[k] >= 0 && desc->textures[i].colour[k] <= 1)) s->d.wide_radiance = 1;
    for (uint32_t i = 0; i < desc->n_materials; ++i)
        for (int k = 0; k < 3; ++k)
            if (desc->materials[i].kind == RT_MAT_METAL && !(desc->materials[i].albedo[k] >= 0 && desc->materials[i].albedo[k] <= 1)) s->d.wide_radiance = 1;
    // (a Perlin texture's colour is a blend, with weights in [0, 1], of the caller's table values: perlin.cpp:33-47)
    for (uint32_t i = 0; i < desc->n_textures; ++i)
        if (desc->textures[i].kind == RT_TEX_PERLIN)
            for (int k = 0; k < 256; ++k)
                if (!(ranfloat[(size_t)k] >= 0 && ranfloat[(size_t)k] <= 1)) s->d.wide_radiance = 1;
    // RTIOW_RECHECK_ALL_LEAVES makes every hit of the static-scene variant take its second, exact
    // walk (tests).
    {
        s->render_defer = static_scene;
        s->d.recheck_all = getenv("RTIOW_RECHECK_ALL_LEAVES") ? 1u : 0u;
    }
    {   // The reach of the moving spheres: how far, per axis, a sphere at ray time t can stick out of the box
        // the walk tests before it is tried -- the box of its leaf, whatever the caller put there (the
        // reference's covers the positions at the bvh's own times, moving_sphere.cpp:51-60, which need not
        // be the sphere's time0 and time1).  Along axis k the centre is m + v t with v = dc / (time1 -
        // time0) and m = c0 - v time0, so the sphere's high side exceeds the box's by (m + r - hi) + v t
        // <= max(0, m + r - hi) + |v| |t|, the low side likewise: reach_a = max |v|, reach_b = the largest
        // excess at t = 0.  Every box above a leaf contains the leaf's, so the bound holds for them too.
        auto up = [](double v) { const float f = (float)v; return (double)f < v ? nextafterf(f, HUGE_VALF) : f; };
        auto worst = [](double now, double v) { return (v >= now) ? v : (std::isnan(v) ? HUGE_VAL : now); };   // (a zero time span: infinite or NaN)
        for (int k = 0; k < 3; ++k) {
            double a = 0, b = 0;
            for (size_t i = 0; i < fl.prims.size(); ++i) {
                const DPrim &p = fl.prims[i];
                if (p.orig < 0 || p.kind == PK_STATIC) continue;
                const double *box = fl.leaf_boxes[i / 2].box;
                const double v = p.dc[k] / (p.t1 - p.t0), m = p.c0[k] - v * p.t0, r = std::fabs(p.r);
                // (these sums are rounded, as are the reference's when it forms the box: 2^-40 of the magnitudes covers both)
                const double pad = 0x1p-40 * (std::fabs(m) + r);
                a = worst(a, std::fabs(v));
                b = worst(b, (m + r) - box[2 * k + 1] + pad);
                b = worst(b, box[2 * k] - (m - r) + pad);
            }
            s->d.reach_a[k] = up(a * (1 + 0x1p-20));
            s->d.reach_b[k] = up(b * (1 + 0x1p-20));
        }
    }
    if (static_scene && fl.far_allowance > 0) {
        for (int a = 0; a < 3; ++a) s->far_free_centre[a] = extent_c[a];
        s->far_free_radius = 3 * extent_h * (1 - 0x1p-20);
    }
    bool any_static = false;
    for (const DPrim &p : fl.prims) any_static = any_static || (p.orig >= 0 && p.kind == PK_STATIC);
    if (any_static) {   // see FarSlack
        double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL}, r_min = HUGE_VAL;
        for (const DPrim &p : fl.prims) {
            if (p.orig < 0 || p.kind != PK_STATIC) continue;
            for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], p.c0[a]); hi[a] = std::fmax(hi[a], p.c0[a]); }
            r_min = std::fmin(r_min, std::fabs(p.r));
        }
        double h2 = 0;
        for (int a = 0; a < 3; ++a) { const double e = 0.5 * (hi[a] - lo[a]); h2 += e * e; }
        FarSlack &f = s->d.far;
        f.cx = (float)(0.5 * (lo[0] + hi[0])); f.cy = (float)(0.5 * (lo[1] + hi[1])); f.cz = (float)(0.5 * (lo[2] + hi[2]));
        // the centre is rounded to single precision: the half-diagonal makes up for it (2^-22 of the coordinates)
        const double coarse = 0x1p-22 * (std::fabs(lo[0]) + std::fabs(hi[0]) + std::fabs(lo[1]) + std::fabs(hi[1]) + std::fabs(lo[2]) + std::fabs(hi[2]));
        f.h = nextafterf((float)((std::sqrt(h2) + coarse) * (1 + 0x1p-20)), HUGE_VALF);
        f.k = r_min > 0 ? nextafterf((float)(0x1p-49 / r_min * (1 + 0x1p-20)), HUGE_VALF) : HUGE_VALF;
        if (!(f.k > 0)) f.k = 0x1p-126f;   // (a radius beyond 2^77: the slack is below every other margin, but stay in the widening branch)
        if (!std::isfinite(f.h) || !std::isfinite(f.cx) || !std::isfinite(f.cy) || !std::isfinite(f.cz)) { f.cx = f.cy = f.cz = 0; f.h = HUGE_VALF; }
    }
    s->lds_bytes = (size_t)s->d.stack_depth * kBlock * sizeof(int);
    s->n_nodes_dev = (uint32_t)n_nodes;
    s->n_prims_dev = (uint32_t)fl.prims.size();
    if (getenv("RTIOW_VERBOSE"))
        fprintf(stderr, "rtiow: %zu nodes, %zu spheres, tree depth %u, %s layout; render kernel: %s, %u nodes and %u leaves in LDS, "
                        "%u stack entries in LDS, %zu B of LDS (budget %zu), %zu B of spill, leaf boxes tested %s\n",
                n_nodes, (size_t)desc->n_prims, fl.max_depth, reference_layout ? "reference" : (on_device ? "built on the device" : "re-clustered"),
                s->render_wide ? (s->render_lds_tree ? (s->render_waves == kWavesMax ? "four-wide tree whole in LDS, 1024 threads" : "four-wide tree whole in LDS, 768 threads") : "four-wide tree in device memory, 1024 threads")
                               : s->render_lds_tree ? (s->render_waves == kWavesMax ? "whole tree in LDS, 1024 threads" : "whole tree in LDS, 768 threads") : "top levels in LDS, 1024 threads",
                s->render_lnodes, s->render_lleaves,
                s->render_stack_cap, s->render_lds_bytes, kLdsBudget, spill_ints * sizeof(int), s->render_defer ? "for the winning leaf" : "up front");
    if (verbose) fprintf(stderr, "rtiow: scene created in %.3f s (%.3f s of it after the layout: kernel shape, top levels, upload)\n", seconds_since(t_start), seconds_since(t_start) - t_layout);
    *out = owner.release();
    return RT_OK;
}

void rt_scene_destroy(rt_scene *scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    delete scene;
}

int rt_trace_batch(rt_scene *scene, const rt_ray *rays, size_t n, double t_min, double t_max, rt_hit *out) {
    if (!scene || (n && (!rays || !out))) return fail(RT_E_INVALID, "rt_trace_batch: null argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(scene->device));
    dev_buf<rt_ray> d_rays;
    dev_buf<rt_hit> d_out;
    HIP_TRY(d_rays.upload(rays, n));
    HIP_TRY(d_out.alloc(n));
    hipLaunchKernelGGL(k_trace_batch, dim3(grid_for(n)), dim3(kBlock), scene->lds_bytes, 0, scene->d, d_rays.p, n, t_min,
                       t_max, d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, n * sizeof(rt_hit), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_aabb_hit_batch(int device, const double *boxes, const rt_ray *rays, const double *t_min, const double *t_max,
                      size_t n, uint8_t *out) {
    if (n && (!boxes || !rays || !t_min || !t_max || !out)) return fail(RT_E_INVALID, "rt_aabb_hit_batch: null argument");
    if (n == 0) return RT_OK;
    if (int rc = use_device(device, nullptr)) return rc;
    dev_buf<double> d_boxes, d_tmin, d_tmax;
    dev_buf<rt_ray> d_rays;
    dev_buf<uint8_t> d_out;
    HIP_TRY(d_boxes.upload(boxes, n * 6));
    HIP_TRY(d_rays.upload(rays, n));
    HIP_TRY(d_tmin.upload(t_min, n));
    HIP_TRY(d_tmax.upload(t_max, n));
    HIP_TRY(d_out.alloc(n));
    hipLaunchKernelGGL(k_aabb_batch, dim3(grid_for(n)), dim3(kBlock), 0, 0, d_boxes.p, d_rays.p, d_tmin.p, d_tmax.p, n, d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, n, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_texture_batch(rt_scene *scene, uint32_t texture, const double *p, size_t n, double *rgb) {
    if (!scene || (n && (!p || !rgb))) return fail(RT_E_INVALID, "rt_texture_batch: null argument");
    if (texture >= scene->n_textures) return fail(RT_E_INVALID, "rt_texture_batch: texture %u out of range", texture);
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(scene->device));
    dev_buf<double> d_p, d_rgb;
    HIP_TRY(d_p.upload(p, n * 3));
    HIP_TRY(d_rgb.alloc(n * 3));
    hipLaunchKernelGGL(k_texture_batch, dim3(grid_for(n)), dim3(kBlock), 0, 0, scene->d, texture, d_p.p, n, d_rgb.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rgb, d_rgb.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scatter_batch(rt_scene *scene, const int32_t *prim, const rt_ray *rays, const rt_hit *hits, const double *tape,
                     uint32_t k, size_t n, uint8_t *scattered, double *attenuation, rt_ray *out, int32_t *used) {
    if (!scene || (n && (!prim || !rays || !hits || !scattered || !attenuation || !out || !used)) || (n && k && !tape))
        return fail(RT_E_INVALID, "rt_scatter_batch: null argument");
    if (n == 0) return RT_OK;
    std::vector<uint32_t> mat;
    try {
        mat.resize(n);
    } catch (const std::bad_alloc &) {
        return fail(RT_E_INVALID, "rt_scatter_batch: out of host memory");
    }
    for (size_t i = 0; i < n; ++i) {
        if (prim[i] < 0 || (uint32_t)prim[i] >= scene->n_prims_desc)
            return fail(RT_E_INVALID, "rt_scatter_batch: item %zu: primitive %d out of range", i, prim[i]);
        mat[i] = scene->material_of_prim[prim[i]];
    }
    HIP_TRY(hipSetDevice(scene->device));
    dev_buf<uint32_t> d_mat;
    dev_buf<rt_ray> d_rays, d_out;
    dev_buf<rt_hit> d_hits;
    dev_buf<double> d_tape, d_att;
    dev_buf<uint8_t> d_sc;
    dev_buf<int32_t> d_used;
    HIP_TRY(d_mat.upload(mat.data(), n));
    HIP_TRY(d_rays.upload(rays, n));
    HIP_TRY(d_hits.upload(hits, n));
    HIP_TRY(d_tape.upload(tape, n * (size_t)k));
    HIP_TRY(d_out.alloc(n));
    HIP_TRY(d_att.alloc(n * 3));
    HIP_TRY(d_sc.alloc(n));
    HIP_TRY(d_used.alloc(n));
    hipLaunchKernelGGL(k_scatter_batch, dim3(grid_for(n)), dim3(kBlock), 0, 0, scene->d, d_mat.p, d_rays.p, d_hits.p,
                       d_tape.p, k, n, d_sc.p, d_att.p, d_out.p, d_used.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(scattered, d_sc.p, n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(attenuation, d_att.p, n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out, d_out.p, n * sizeof(rt_ray), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(used, d_used.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i)
        if (used[i] < 0) return fail(RT_E_TAPE, "rt_scatter_batch: item %zu needs more than %u uniforms", i, k);
    return RT_OK;
}

static DCamera to_dcamera(const rt_camera *c) {
    DCamera d;
    for (int k = 0; k < 3; ++k) {
        d.origin[k] = c->origin[k]; d.u[k] = c->u[k]; d.v[k] = c->v[k]; d.w[k] = c->w[k];
        d.llc[k] = c->lower_left_corner[k]; d.horizontal[k] = c->horizontal[k]; d.vertical[k] = c->vertical[k];
    }
    d.lens_radius = c->lens_radius; d.t0 = c->time0; d.t1 = c->time1;
    return d;
}

int rt_camera_batch(int device, const rt_camera *camera, const double *st, const double *tape, uint32_t k, size_t n,
                    rt_ray *out, int32_t *used) {
    if (!camera || (n && (!st || !out || !used)) || (n && k && !tape)) return fail(RT_E_INVALID, "rt_camera_batch: null argument");
    if (n == 0) return RT_OK;
    if (int rc = use_device(device, nullptr)) return rc;
    dev_buf<double> d_st, d_tape;
    dev_buf<rt_ray> d_out;
    dev_buf<int32_t> d_used;
    HIP_TRY(d_st.upload(st, n * 2));
    HIP_TRY(d_tape.upload(tape, n * (size_t)k));
    HIP_TRY(d_out.alloc(n));
    HIP_TRY(d_used.alloc(n));
    hipLaunchKernelGGL(k_camera_batch, dim3(grid_for(n)), dim3(kBlock), 0, 0, to_dcamera(camera), d_st.p, d_tape.p, k, n,
                       d_out.p, d_used.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d_out.p, n * sizeof(rt_ray), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(used, d_used.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i)
        if (used[i] < 0) return fail(RT_E_TAPE, "rt_camera_batch: item %zu needs more than %u uniforms", i, k);
    return RT_OK;
}

// Where one frame keeps its working memory: the scene's own buffers (rt_render) or the
// per-device scratch array (rt_render_noalloc).
struct frame_memory {
    unsigned long long *counters;   // kCounters entries
    uint32_t *tiles;                // tile costs, then the tile order; null = none available
    size_t tiles_capacity;          // entries
    int *spill;                     // deep stack entries of trees that are not in LDS whole (null when unused)
};

static int check_render_args(const char *who, rt_scene *scene, const rt_camera *camera, const rt_render_params *pr, void *accum_dev) {
    if (!scene || !camera || !pr || !accum_dev) return fail(RT_E_INVALID, "%s: null argument", who);
    if (pr->nx == 0 || pr->ny == 0) return fail(RT_E_INVALID, "%s: empty image", who);
    if ((uint64_t)pr->nx * pr->ny > 0xFFFFFFFFull / 3) return fail(RT_E_INVALID, "%s: image too large", who);
    if (pr->sample_end < pr->sample_begin) return fail(RT_E_INVALID, "%s: sample_end < sample_begin", who);
    if (pr->sample_end - pr->sample_begin > (1u << 24)) return fail(RT_E_INVALID, "%s: more than 2^24 samples per pixel in one call", who);
    if (pr->accum_type > RT_ACCUM_F64) return fail(RT_E_INVALID, "%s: unknown accum_type %u", who, pr->accum_type);
    return RT_OK;
}

// how a frame is cut into tiles and handed to the persistent blocks
struct frame_plan {
    uint32_t tile_shift, tiles_x, n_tiles, waves, blocks;
    bool ordered;
    uint32_t regions;   // > 1: a strip of the frame and a tile queue per XCD
};

static frame_plan plan_frame(const rt_scene *scene, const rt_render_params *pr) {
    frame_plan f;
    const uint32_t spp = pr->sample_end - pr->sample_begin;
    // 8x8-pixel tiles fill the 64 lanes with a single sample per pixel; from 4 samples per pixel a
    // 4x4 tile does, and tiles a quarter the size leave a quarter of the imbalance at the end of
    // the frame (measured faster from 8 to 1024 samples, within 2 % at 32).
    // (with several tiles open per wave no lane waits at a tile's end, and the small tiles win at every
    // depth from 4 samples on: 10^6 spheres at 8 spp 8.1 against 9.0 ms, the 485 spheres 3.3 against 4.8)
    f.tile_shift = spp >= 4 ? 2u : 3u;
    if (const char *e = getenv("RTIOW_TILE_SHIFT")) f.tile_shift = std::min(3u, std::max(1u, (uint32_t)atoi(e)));   // 1: 2x2 ... 3: 8x8 (experiments)
    const uint32_t tile_side = 1u << f.tile_shift;
    f.tiles_x = (pr->nx + tile_side - 1) >> f.tile_shift;
    f.n_tiles = f.tiles_x * ((pr->ny + tile_side - 1) >> f.tile_shift);
    f.waves = 4 * (uint32_t)scene->render_waves;
    f.blocks = std::min<uint32_t>((uint32_t)scene->n_cus, (f.n_tiles + f.waves - 1) / f.waves);   // the spill area is sized for n_cus blocks
    // Worth a costing pass when every wave gets several tiles and the frame is many samples deep:
    // the pass traces 1 to 4 samples per pixel without touching the image, under 2 % of the work.
    f.ordered = spp >= kOrderMinSamples && f.n_tiles >= 2 * f.blocks * f.waves && !getenv("RTIOW_TILE_ORDER_OFF");
    // a tree in device memory: the tiles of each of eight vertical strips go to the blocks of one XCD first
    // (at least 64 tile columns, so that the costing pass's 8x8 tiles make strips too)
    f.regions = !scene->render_lds_tree && f.tiles_x >= 8 * kRegions && f.blocks >= kRegions ? kRegions : 1u;
    if (const char *e = getenv("RTIOW_TILE_REGIONS")) f.regions = atoi(e) > 1 && f.tiles_x >= 2 * kRegions ? kRegions : 1u;   // (A/B, tests)
    return f;
}

// Queues one frame on `stream`: counters cleared, costing pass and tile sort when the plan says so,
// then the main launch.  Allocates nothing.
static int launch_frame(rt_scene *scene, const rt_camera *camera, const rt_render_params *pr, void *accum_dev, hipStream_t stream,
                        const frame_plan &f, const frame_memory &mem, bool order_on_file) {
    const uint32_t spp = pr->sample_end - pr->sample_begin;
    // counters[0] = rays traced, [1] = next tile of the frame, [4..11] = next tile of each XCD's strip; from kPassCounters on: the same for the costing pass
    HIP_TRY(hipMemsetAsync(mem.counters, 0, kCounters * sizeof(unsigned long long), stream));
    dim3 grid(f.blocks), block(64 * f.waves);
    DCamera cam = to_dcamera(camera);
    DScene dsc = scene->d;
    if (scene->far_free_radius > 0) {   // the sphere boxes' padding covers every origin of this frame: no per-segment slack
        double d2 = 0;
        for (int a = 0; a < 3; ++a) { const double e = camera->origin[a] - scene->far_free_centre[a]; d2 += e * e; }
        if (std::sqrt(d2) + std::fabs(camera->lens_radius) * 2 <= scene->far_free_radius) dsc.far.k = 0;
    }
#define RT_LAUNCH(T, L, W, C, D, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT)                                                        \
    RT_LAUNCH_WIDTH(T, L, W, C, D, false, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT)
#define RT_LAUNCH_WIDTH(T, L, W, C, D, X, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT)                                                        \
    hipLaunchKernelGGL((k_render<T, L, W, C, D, X>), grid, block, scene->render_lds_bytes, stream, dsc, cam, pr->nx, pr->ny,   \
                       s_from, S_END, pr->max_depth, pr->t_min, pr->seed, (T *)(ACCUM), COUNTERS,                           \
                       scene->render_lnodes, scene->render_lleaves, scene->render_stack_cap, mem.spill, ORDER, COST, SHIFT, \
                       scene->render_sums_per_wave, regions)
#define RT_LAUNCH_TREE(T, C, D, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT)                                                      \
    do {                                                                                                                  \
        if (!scene->render_lds_tree) RT_LAUNCH(T, false, kWavesMax, C, D, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT);               \
        else if (scene->render_waves == kWavesMax) RT_LAUNCH(T, true, kWavesMax, C, D, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT);  \
        else RT_LAUNCH(T, true, kWavesMin, C, D, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT);                                       \
    } while (0)
#define RT_LAUNCH_SHAPE(T, C, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT)                                                        \
    do {                                                                                                                  \
        if (scene->render_wide) {                                                                                         \
            if (!scene->render_lds_tree) RT_LAUNCH_WIDTH(T, false, kWavesMax, C, true, true, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT);               \
            else if (scene->render_waves == kWavesMax) RT_LAUNCH_WIDTH(T, true, kWavesMax, C, true, true, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT);  \
            else RT_LAUNCH_WIDTH(T, true, kWavesMin, C, true, true, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT);                                       \
        } else if (scene->render_defer) RT_LAUNCH_TREE(T, C, true, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT);                             \
        else RT_LAUNCH_TREE(T, C, false, S_END, ACCUM, COUNTERS, ORDER, COST, SHIFT);                                              \
    } while (0)
    uint32_t s_from = pr->sample_begin;
    const uint32_t *order = nullptr;
    // trees in device memory: a strip of the frame and a tile queue per XCD (see k_render)
    const uint32_t regions = f.regions;
    if (f.ordered) {
        const uint32_t cost_tiles_x = (pr->nx + 7) / 8, n_cost = cost_tiles_x * ((pr->ny + 7) / 8);
        uint32_t *cost = mem.tiles, *sorted = mem.tiles + n_cost;
        uint32_t probe = std::min(4u, std::max(1u, spp / 64u));   // measured: 1 sample ranks noisily, more than 4 costs more than it saves
        if (const char *e = getenv("RTIOW_TILE_PROBE")) probe = std::min<uint32_t>(spp, (uint32_t)std::max(1, atoi(e)));
        if (!order_on_file) {
            RT_LAUNCH_SHAPE(float, true, pr->sample_begin + probe, nullptr, mem.counters + kPassCounters, (const uint32_t *)nullptr, cost, 3u);
            hipLaunchKernelGGL(k_order_tiles, dim3(1), dim3(kOrderBlock), 0, stream, cost, cost_tiles_x, n_cost, f.tiles_x, f.n_tiles, 3u - f.tile_shift, sorted, regions);
        }
        order = sorted;
    }
    // (a launch adds at most kSamplesPerLaunch samples to a pixel's integer sums; deeper frames take several)
    for (uint64_t from = pr->sample_begin; from < pr->sample_end; from += kSamplesPerLaunch) {
        s_from = (uint32_t)from;
        const uint32_t s_to = pr->sample_end - s_from > kSamplesPerLaunch ? s_from + kSamplesPerLaunch : pr->sample_end;
        if (s_from != pr->sample_begin) HIP_TRY(hipMemsetAsync(mem.counters + 1, 0, (kPassCounters - 1) * sizeof(unsigned long long), stream));   // the tile queues start over
        if (pr->accum_type == RT_ACCUM_F32) RT_LAUNCH_SHAPE(float, false, s_to, accum_dev, mem.counters, order, (uint32_t *)nullptr, f.tile_shift);
        else RT_LAUNCH_SHAPE(double, false, s_to, accum_dev, mem.counters, order, (uint32_t *)nullptr, f.tile_shift);
    }
#undef RT_LAUNCH_SHAPE
#undef RT_LAUNCH_TREE
#undef RT_LAUNCH_WIDTH
#undef RT_LAUNCH
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_render(rt_scene *scene, const rt_camera *camera, const rt_render_params *pr, void *accum_dev, void *stream_v,
              rt_stats *stats) {
    if (int rc = check_render_args("rt_render", scene, camera, pr, accum_dev)) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t stream = (hipStream_t)stream_v;
    struct event_pair {   // destroyed on every return
        hipEvent_t a = nullptr, b = nullptr;
        ~event_pair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    const frame_plan f = plan_frame(scene, pr);
    if (f.ordered && scene->tiles_capacity < 2 * (size_t)f.n_tiles + 16) {   // costs of the 8x8 tiles (at most n_tiles of them), then the order and its strips' starts
        if (scene->tiles.p) { (void)hipFree(scene->tiles.p); scene->tiles.p = nullptr; scene->tiles_capacity = 0; }
        scene->order_valid = false;
        HIP_TRY(scene->tiles.alloc(2 * (size_t)f.n_tiles + 16));
        scene->tiles_capacity = 2 * (size_t)f.n_tiles + 16;
    }
    if (stats) {
        HIP_TRY(hipEventCreate(&ev.a));
        HIP_TRY(hipEventCreate(&ev.b));
        HIP_TRY(hipEventRecord(ev.a, stream));
    }
    // A tile's cost is a property of the view, not of the samples: a frame of the same camera, size
    // and depth limit as the last ordered frame of this scene (the next frame of an animation's
    // still shot, the next batch of samples of a progressive render, the next step of a benchmark)
    // takes over its order and skips the costing pass.  RTIOW_TILE_COST_REUSE=0 measures every frame.
    bool order_on_file = false;
    if (f.ordered) {
        rt_scene::order_key key{};
        key.camera = *camera; key.nx = pr->nx; key.ny = pr->ny; key.max_depth = pr->max_depth; key.tile_shift = f.tile_shift | (f.regions << 8); key.t_min = pr->t_min; key.stream = stream_v;
        const char *reuse = getenv("RTIOW_TILE_COST_REUSE");
        order_on_file = scene->order_valid && memcmp(&key, &scene->order_of, sizeof key) == 0 && !(reuse && atoi(reuse) == 0) && !getenv("RTIOW_TILE_PROBE");
        scene->order_of = key;
        scene->order_valid = true;
    }
    const frame_memory mem = {scene->counter.p, scene->tiles.p, scene->tiles_capacity, scene->spill.p};
    if (int rc = launch_frame(scene, camera, pr, accum_dev, stream, f, mem, order_on_file)) { scene->order_valid = false; return rc; }
    if (stats) {
        HIP_TRY(hipEventRecord(ev.b, stream));
        HIP_TRY(hipEventSynchronize(ev.b));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        unsigned long long rays = 0;
        HIP_TRY(hipMemcpy(&rays, scene->counter.p, sizeof rays, hipMemcpyDeviceToHost));
#ifdef RT_LANE_STATS
        std::vector<unsigned long long> all(kCounters);
        HIP_TRY(hipMemcpy(all.data(), scene->counter.p, kCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long c[16] = {0};
        unsigned long long clk[3] = {0, 0, 0};
        for (size_t w = 0; w < (kCounters - kQueueCounters) / 16; ++w) {
            for (int q = 0; q < 12; ++q) c[4 + q] += all[kQueueCounters + 16 * w + q];
            for (int q = 0; q < 3; ++q) clk[q] += all[kQueueCounters + 16 * w + 12 + q];
        }
        {
            const double sum = (double)(clk[0] + clk[1] + clk[2]);
            fprintf(stderr, "clocks: claiming %.1f %%, set-up and walk %.1f %%, shading %.1f %% of the waves' round time\n", 100 * clk[0] / sum, 100 * clk[1] / sum, 100 * clk[2] / sum);
        }
        const char *phase[6] = {"node step", "leaf step", "round", "hit shading", "sky", "claim"};
        for (int q = 0; q < 6; ++q)
            fprintf(stderr, "lanes: %-11s executed %llu times by a wave, %.1f of 64 lanes, %.2f lane-visits per segment\n", phase[q], c[4 + 2 * q],
                    c[4 + 2 * q] ? (double)c[5 + 2 * q] / (double)c[4 + 2 * q] : 0.0, rays ? (double)c[5 + 2 * q] / (double)rays : 0.0);
#endif
        stats->kernel_ms = ms;
        stats->rays = rays;
        stats->paths = (uint64_t)pr->nx * pr->ny * (pr->sample_end - pr->sample_begin);
    }
    return RT_OK;
}

int rt_render_noalloc(rt_scene *scene, const rt_camera *camera, const rt_render_params *pr, void *accum_dev, void *stream_v) {
    if (int rc = check_render_args("rt_render_noalloc", scene, camera, pr, accum_dev)) return rc;
#ifdef RT_LANE_STATS
    return fail(RT_E_UNSUPPORTED, "rt_render_noalloc: not available in the lane-statistics build");
#endif
    HIP_TRY(hipSetDevice(scene->device));
    unsigned char *base = nullptr;
    if (int rc = scratch_on_device(scene->device, &base)) return rc;   // (looked up when the device's first scene was created)
    const frame_plan f = plan_frame(scene, pr);
    if (f.ordered && 2 * (size_t)f.n_tiles + 16 > kScratchTiles)
        return fail(RT_E_UNSUPPORTED, "rt_render_noalloc: %u tiles of %ux%u pixels; the scratch area orders at most %zu (images up to about %zu pixels)",
                    f.n_tiles, 1u << f.tile_shift, 1u << f.tile_shift, kScratchTiles / 2 - 8, (kScratchTiles / 2 - 8) * 16);
    if (scene->spill_ints > kScratchSpillInts)
        return fail(RT_E_UNSUPPORTED, "rt_render_noalloc: the scene's tree needs %zu bytes of stack spill, the scratch area holds %zu",
                    scene->spill_ints * sizeof(int), kScratchSpillInts * sizeof(int));
    const frame_memory mem = {reinterpret_cast<unsigned long long *>(base), reinterpret_cast<uint32_t *>(base + kScratchCounterBytes), kScratchTiles,
                              scene->spill_ints ? reinterpret_cast<int *>(base + kScratchCounterBytes + kScratchTiles * sizeof(uint32_t)) : nullptr};
    return launch_frame(scene, camera, pr, accum_dev, (hipStream_t)stream_v, f, mem, false);   // (the array is shared: every frame is costed)
}

int rt_render_host(rt_scene *scene, const rt_camera *camera, const rt_render_params *pr, void *accum_host, rt_stats *stats) {
    if (!scene || !camera || !pr || !accum_host) return fail(RT_E_INVALID, "rt_render_host: null argument");
    if (pr->accum_type > RT_ACCUM_F64) return fail(RT_E_INVALID, "rt_render_host: unknown accum_type %u", pr->accum_type);
    HIP_TRY(hipSetDevice(scene->device));
    size_t bytes = (size_t)pr->nx * pr->ny * 3 * (pr->accum_type == RT_ACCUM_F32 ? sizeof(float) : sizeof(double));
    dev_buf<uint8_t> d;
    HIP_TRY(d.alloc(bytes));
    HIP_TRY(hipMemset(d.p, 0, bytes));
    if (int rc = rt_render(scene, camera, pr, d.p, nullptr, stats)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(accum_host, d.p, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---- several GPUs ------------------------------------------------------------------------
}  // extern "C"

namespace {

// RCCL, opened when a frame first needs a collective: single-GPU callers never load it.
struct rccl_api {
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
    std::string why;
    rccl_api() {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { why = dlerror() ? dlerror() : "librccl.so.1 not found"; return; }
        comm_init_all = reinterpret_cast<decltype(comm_init_all)>(dlsym(h, "ncclCommInitAll"));
        reduce = reinterpret_cast<decltype(reduce)>(dlsym(h, "ncclReduce"));
        error_string = reinterpret_cast<decltype(error_string)>(dlsym(h, "ncclGetErrorString"));
        ok = comm_init_all && reduce && error_string;
        if (!ok) why = "librccl lacks ncclCommInitAll / ncclReduce / ncclGetErrorString";
    }
};
rccl_api &rccl() { static rccl_api api; return api; }

// One communicator per rank for a list of devices, made once per list and kept until the process ends
// (ncclCommInitAll takes seconds; a renderer calls with the same list frame after frame).
std::mutex g_comm_mutex;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;
int communicators(const std::vector<int> &devices, std::vector<ncclComm_t> &out) {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    auto it = g_comms.find(devices);
    if (it == g_comms.end()) {
        rccl_api &api = rccl();
        if (!api.ok) return fail(RT_E_UNSUPPORTED, "rt_render_multi: RCCL is not available (%s)", api.why.c_str());
        std::vector<ncclComm_t> comms(devices.size());
        const ncclResult_t r = api.comm_init_all(comms.data(), (int)devices.size(), devices.data());
        if (r != ncclSuccess) return fail(RT_E_DEVICE, "ncclCommInitAll over %zu devices: %s", devices.size(), api.error_string(r));
        it = g_comms.emplace(devices, std::move(comms)).first;
    }
    out = it->second;
    return RT_OK;
}

// what one GPU of a multi-GPU frame holds between the render and the reduction
struct gpu_share {
    int device = 0;
    std::vector<uint32_t> scenes;   // indices into the caller's list, in its order
    hipStream_t stream = nullptr;
    void *accum = nullptr;
    double kernel_ms = 0;
    uint64_t rays = 0;
    int rc = RT_OK;
    std::string error;
};

}  // namespace

extern "C" {

int rt_render_multi(rt_scene *const *scenes, uint32_t n_scenes, const rt_camera *camera, const rt_render_params *pr, void *accum_host,
                    rt_stats *stats) {
    if (!scenes || n_scenes == 0) return fail(RT_E_INVALID, "rt_render_multi: no scenes");
    for (uint32_t g = 0; g < n_scenes; ++g)
        if (int rc = check_render_args("rt_render_multi", scenes[g], camera, pr, accum_host)) return rc;
    const size_t n_values = (size_t)pr->nx * pr->ny * 3;
    const size_t bytes = n_values * (pr->accum_type == RT_ACCUM_F32 ? sizeof(float) : sizeof(double));
    try {
        // the GPUs of the list, in order of first appearance; scenes on the same GPU share its buffer
        std::vector<gpu_share> gpus;
        for (uint32_t g = 0; g < n_scenes; ++g) {
            size_t k = 0;
            while (k < gpus.size() && gpus[k].device != scenes[g]->device) ++k;
            if (k == gpus.size()) { gpus.emplace_back(); gpus[k].device = scenes[g]->device; }
            gpus[k].scenes.push_back(g);
        }
        // RTIOW_FORCE_RCCL=1 runs the reduction over a single GPU too (a one-rank communicator): the
        // whole RCCL call sequence on a machine with one GPU
        const bool collective = gpus.size() > 1 || getenv("RTIOW_FORCE_RCCL");
        std::vector<ncclComm_t> comms;
        if (collective) {
            std::vector<int> devices;
            for (const gpu_share &s : gpus) devices.push_back(s.device);
            if (int rc = communicators(devices, comms)) return rc;
        }
        const uint32_t b = pr->sample_begin, spp = pr->sample_end - pr->sample_begin;
        auto each_gpu = [&](auto work) {   // work(share, rank) on one host thread per GPU; false if any failed
            std::vector<std::thread> threads;
            try {
                for (size_t k = 1; k < gpus.size(); ++k) threads.emplace_back([&, k] { work(gpus[k], (int)k); });
            } catch (...) {   // no more threads to be had: the ones that started finish, then the call fails
                for (std::thread &t : threads) t.join();
                throw;
            }
            work(gpus[0], 0);
            for (std::thread &t : threads) t.join();
            for (const gpu_share &s : gpus)
                if (s.rc != RT_OK) return false;
            return true;
        };
        auto failed = [](gpu_share &s, int rc, const char *what, const char *detail) {
            s.rc = rc;
            s.error = std::string(what) + ": " + detail;
        };
        // every GPU renders its scenes' sample ranges into its own sums
        bool ok = each_gpu([&](gpu_share &s, int) {
            hipError_t e = hipSetDevice(s.device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipMalloc(&s.accum, bytes);
            if (e == hipSuccess) e = hipMemsetAsync(s.accum, 0, bytes, s.stream);
            if (e != hipSuccess) return failed(s, RT_E_DEVICE, "rt_render_multi: set-up", hipGetErrorString(e));
            for (uint32_t g : s.scenes) {
                rt_render_params p = *pr;
                p.sample_begin = b + (uint32_t)((uint64_t)spp * g / n_scenes);
                p.sample_end = b + (uint32_t)((uint64_t)spp * (g + 1) / n_scenes);
                if (p.sample_end == p.sample_begin) continue;
                rt_stats st;
                if (int rc = rt_render(scenes[g], camera, &p, s.accum, s.stream, &st)) return failed(s, rc, "rt_render_multi", g_err);
                s.kernel_ms += st.kernel_ms;
                s.rays += st.rays;
            }
        });
        // ... then the sums meet on the first GPU (only when every render went through: a rank that
        // stayed away would leave the others waiting in the collective)
        if (ok && collective)
            ok = each_gpu([&](gpu_share &s, int rank) {
                hipError_t e = hipSetDevice(s.device);
                if (e != hipSuccess) return failed(s, RT_E_DEVICE, "rt_render_multi: hipSetDevice", hipGetErrorString(e));
                const ncclResult_t r = rccl().reduce(s.accum, s.accum, n_values, pr->accum_type == RT_ACCUM_F32 ? ncclFloat : ncclDouble, ncclSum, 0,
                                                     comms[(size_t)rank], s.stream);
                if (r != ncclSuccess) return failed(s, RT_E_DEVICE, "ncclReduce", rccl().error_string(r));
                e = hipStreamSynchronize(s.stream);
                if (e != hipSuccess) return failed(s, RT_E_DEVICE, "rt_render_multi: reduction", hipGetErrorString(e));
            });
        if (ok) {
            gpu_share &s = gpus[0];
            hipError_t e = hipSetDevice(s.device);
            if (e == hipSuccess) e = hipStreamSynchronize(s.stream);
            if (e == hipSuccess) e = hipMemcpy(accum_host, s.accum, bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { failed(s, RT_E_DEVICE, "rt_render_multi: copy to host", hipGetErrorString(e)); ok = false; }
        }
        int rc = RT_OK, where = 0;
        std::string why;
        for (gpu_share &s : gpus) {
            (void)hipSetDevice(s.device);
            if (s.stream) { (void)hipStreamSynchronize(s.stream); (void)hipStreamDestroy(s.stream); }
            if (s.accum) (void)hipFree(s.accum);
            if (s.rc != RT_OK && rc == RT_OK) { rc = s.rc; why = s.error; where = s.device; }
        }
        if (rc != RT_OK) return fail(rc, "%s (device %d, %zu GPUs in the call)", why.c_str(), where, gpus.size());
        if (stats) {
            stats->paths = (uint64_t)pr->nx * pr->ny * spp;
            stats->rays = 0;
            stats->kernel_ms = 0;
            for (const gpu_share &s : gpus) { stats->rays += s.rays; stats->kernel_ms = std::max(stats->kernel_ms, s.kernel_ms); }
        }
        return RT_OK;
    } catch (const std::exception &e) {   // thread or memory exhaustion
        return fail(RT_E_DEVICE, "rt_render_multi: %s", e.what());
    }
}

int rt_resolve(int device, const float *accum_dev, size_t npix, uint32_t spp, uint8_t *rgb8_dev, void *stream) {
    if (npix && (!accum_dev || !rgb8_dev)) return fail(RT_E_INVALID, "rt_resolve: null argument");
    if (npix == 0) return RT_OK;
    if (int rc = use_device(device, nullptr)) return rc;
    hipLaunchKernelGGL(k_resolve, dim3(grid_for(npix * 3)), dim3(kBlock), 0, (hipStream_t)stream, accum_dev, npix * 3, spp, rgb8_dev);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // extern "C"
