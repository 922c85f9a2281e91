// bvh_device.hpp -- the tree over a large scene's spheres, built on the GPU.
//
// What the reference fixes about a bvh_node tree (bvh_node.cpp:8-50) are its leaves and their order:
// they come from the host's bvh_node constructor, as before.  The inner nodes above them only have to
// contain the leaf boxes (see the flattener in rtiow_hip.hip), and for scenes of static spheres the
// device tree is built over the single spheres, each behind a box of its own.  On the host that is a
// binned surface-area clustering; here the same job is done on the device as a linear BVH:
//   1. the box of every sphere (sphere_pad_box, the host's formula) and the 63-bit Morton code of its centre,
//   2. a radix sort of (code, sphere) pairs (rocPRIM through hipCUB),
//   3. the hierarchy over the sorted codes, every inner node in parallel (Karras, "Maximizing
//      parallelism in the construction of BVHs, octrees, and k-d trees", 2012),
//   4. boxes bottom-up: exact unions in double precision (fmin / fmax, as aabb.cpp:59-64), from which each
//      node takes its two child boxes in single precision, moved and rounded outward (plane_down / plane_up),
//   5. the depth of the deepest leaf (stack sizes).
// The tree never visits the host.  A linear BVH is a looser tree than the surface-area one (more node
// visits per ray) and takes milliseconds instead of a tenth of a second for 10^6 spheres: rt_scene_create
// uses it for large scenes, where set-up dominates a frame, and RTIOW_BUILD=host|device overrides.
#pragma once

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "rt_device.hpp"

namespace lbvh {

using namespace rtd;

constexpr int kBuildBlock = 256;

// spreads the low 21 bits of v to every third bit
__device__ __forceinline__ uint64_t spread21(uint32_t v) {
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

struct Grid { double lo[3], scale[3]; };   // centre -> cell: (c - lo) * scale, 2^21 cells per axis

// boxes[i] and the Morton code of sphere refs[i] (a one-sphere leaf reference: leaf number, LEAF_SECOND)
__global__ __launch_bounds__(kBuildBlock) void k_sphere_boxes(const DPrim *prims, const int32_t *refs, uint32_t n, double far_allowance, Grid g,
                                                             double *boxes, uint64_t *codes, uint32_t *order) {
    const uint32_t i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t ref = refs[i];
    const DPrim *p = prims + 2 * (size_t)(ref & LEAF_INDEX_MASK) + ((ref & LEAF_SECOND) ? 1 : 0);
    double b[6];
    sphere_pad_box(p->c0, p->r, far_allowance, b);
    uint64_t code = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        boxes[(size_t)i * 6 + 2 * a] = b[2 * a];
        boxes[(size_t)i * 6 + 2 * a + 1] = b[2 * a + 1];
        double q = (p->c0[a] - g.lo[a]) * g.scale[a];
        q = q >= 0 ? q : 0;                       // (NaN: cell 0)
        const uint32_t cell = q < 2097151.0 ? (uint32_t)q : 2097151u;
        code |= spread21(cell) << a;
    }
    codes[i] = code;
    order[i] = i;
}

// length of the common prefix of the codes at sorted positions i and j (equal codes: told apart by position); -1 outside
__device__ __forceinline__ int common_prefix(const uint64_t *codes, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = codes[i], b = codes[j];
    if (a == b) return 64 + __clz(i ^ j);
    return __clzll((long long)(a ^ b));
}

// Inner node i of the n - 1: its range of sorted leaves and the split.  A child is written as an inner
// node's number (>= 0) or as ~(sorted position) of a leaf; parents go to parent_inner / parent_leaf.
__global__ __launch_bounds__(kBuildBlock) void k_hierarchy(const uint64_t *codes, int n, int2 *children, int *parent_inner, int *parent_leaf) {
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= n - 1) return;
    const int d = common_prefix(codes, n, i, i + 1) - common_prefix(codes, n, i, i - 1) >= 0 ? 1 : -1;
    const int least = common_prefix(codes, n, i, i - d);
    int reach = 2;
    while (common_prefix(codes, n, i, i + reach * d) > least) reach *= 2;
    int len = 0;
    for (int t = reach / 2; t >= 1; t /= 2)
        if (common_prefix(codes, n, i, i + (len + t) * d) > least) len += t;
    const int j = i + len * d;
    const int shared = common_prefix(codes, n, i, j);
    int s = 0, t = len;
    do {
        t = (t + 1) / 2;
        if (common_prefix(codes, n, i, i + (s + t) * d) > shared) s += t;
    } while (t > 1);
    const int split = i + s * d + (d < 0 ? d : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const int left = first == split ? ~split : split, right = last == split + 1 ? ~(split + 1) : split + 1;
    children[i] = make_int2(left, right);
    if (left >= 0) parent_inner[left] = i; else parent_leaf[~left] = i;
    if (right >= 0) parent_inner[right] = i; else parent_leaf[~right] = i;
    if (i == 0) parent_inner[0] = -1;
}

__device__ __forceinline__ double load_shared_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_shared_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Boxes from the leaves up: a thread per leaf climbs; at every inner node the first thread to arrive
// stops, the second -- which finds both subtrees finished -- forms the node and climbs on.  `boxes` are
// the spheres' in their original order, `order` maps sorted positions to it.
__global__ __launch_bounds__(kBuildBlock) void k_fit(const double *boxes, const uint32_t *order, const int32_t *refs, const int2 *children,
                                                    const int *parent_inner, const int *parent_leaf, int n, unsigned int *arrived,
                                                    double *node_boxes, DNode *nodes, int32_t first_index) {
    const int leaf = blockIdx.x * kBuildBlock + threadIdx.x;
    if (leaf >= n) return;
    int node = parent_leaf[leaf];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(arrived + node, 1u) == 0u) return;
        __threadfence();
        const int2 ch = children[node];
        const int c[2] = {ch.x, ch.y};
        DNode out;
        double whole[6];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            double b[6];
            if (c[k] >= 0) {
#pragma unroll
                for (int q = 0; q < 6; ++q) b[q] = load_shared_f64(node_boxes + (size_t)c[k] * 6 + q);
                out.child[k] = first_index + c[k];
            } else {
                const uint32_t at = order[~c[k]];
#pragma unroll
                for (int q = 0; q < 6; ++q) b[q] = boxes[(size_t)at * 6 + q];
                out.child[k] = refs[at];
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                out.bx[k][q] = (q & 1) ? plane_up(b[q]) : plane_down(b[q]);
                whole[q] = k == 0 ? b[q] : ((q & 1) ? fmax(whole[q], b[q]) : fmin(whole[q], b[q]));
            }
        }
        out.flags = NF_SAFE0 | NF_SAFE1;   // static spheres stay inside their boxes
        out.pad = 0;
        nodes[node] = out;
#pragma unroll
        for (int q = 0; q < 6; ++q) store_shared_f64(node_boxes + (size_t)node * 6 + q, whole[q]);
        node = parent_inner[node];
    }
}

// the number of inner nodes above the deepest leaf
__global__ __launch_bounds__(kBuildBlock) void k_depth(const int *parent_inner, const int *parent_leaf, int n, unsigned int *deepest) {
    const int leaf = blockIdx.x * kBuildBlock + threadIdx.x;
    if (leaf >= n) return;
    unsigned int d = 0;
    for (int node = parent_leaf[leaf]; node >= 0; node = parent_inner[node]) ++d;
    atomicMax(deepest, d);
}

// Copies the top of the tree, breadth-first from `root`, into nodes[0, k): the part the render kernel keeps
// in LDS (its nodes are the first k of the table).  A copy's children that were copied too are referred to
// by their new numbers, the others by their old ones; the originals stay where they are, unreferenced.
// One lane: a thousand nodes, each a dependent 64-byte read.
constexpr uint32_t kStageMax = 4096;
__global__ void k_stage_top(DNode *nodes, uint32_t k, int32_t root) {
    __shared__ int32_t queue[kStageMax];
    if (threadIdx.x != 0 || blockIdx.x != 0 || k == 0) return;
    uint32_t count = 1;
    queue[0] = root;
    for (uint32_t head = 0; head < count; ++head) {
        DNode nd = nodes[queue[head]];
        for (int c = 0; c < 2; ++c) {
            if (nd.child[c] >= 0 && count < k) {
                queue[count] = nd.child[c];
                nd.child[c] = (int32_t)count++;
            }
        }
        nodes[head] = nd;
    }
}

struct scratch {   // freed when the build ends, whichever way
    void *p[16] = {};
    int used = 0;
    ~scratch() { for (int i = 0; i < used; ++i) (void)hipFree(p[i]); }
    template <class T> hipError_t get(T **out, size_t count) {
        void *q = nullptr;
        const hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) p[used++] = q;
        *out = static_cast<T *>(q);
        return e;
    }
};

// Builds the tree over the n >= 2 single spheres `refs` (host array; leaf references into d_prims, which
// is already on the current device) into d_nodes[staged, staged + n - 1) (device memory of the caller's:
// staged + n - 1 nodes), node j of the build being number staged + j of the table, and copies the top
// `staged` nodes (as many as the tree has, at most) to its front, breadth-first: the walk starts at node 0.
// *staged_out is the number copied.  Returns the HIP error of the first call that failed.
inline hipError_t build(const DPrim *d_prims, const int32_t *refs, uint32_t n, double far_allowance, const double centre_lo[3],
                        const double centre_hi[3], DNode *d_table, uint32_t staged, uint32_t *depth_out) {
    if (staged > kStageMax) staged = kStageMax;
    DNode *const d_nodes = d_table + staged;
    const int32_t first_index = (int32_t)staged;
    scratch mem;
    int32_t *d_refs;
    double *d_boxes, *d_node_boxes;
    uint64_t *d_codes, *d_codes_sorted;
    uint32_t *d_order, *d_order_sorted;
    int2 *d_children;
    int *d_parent_inner, *d_parent_leaf;
    unsigned int *d_counters;   // arrival counts of the n - 1 nodes, then the depth
    hipError_t e = hipSuccess;
#define LBVH_TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
    LBVH_TRY(mem.get(&d_refs, n));
    LBVH_TRY(mem.get(&d_boxes, (size_t)n * 6));
    LBVH_TRY(mem.get(&d_node_boxes, (size_t)n * 6));
    LBVH_TRY(mem.get(&d_codes, n));
    LBVH_TRY(mem.get(&d_codes_sorted, n));
    LBVH_TRY(mem.get(&d_order, n));
    LBVH_TRY(mem.get(&d_order_sorted, n));
    LBVH_TRY(mem.get(&d_children, n));
    LBVH_TRY(mem.get(&d_parent_inner, n));
    LBVH_TRY(mem.get(&d_parent_leaf, n));
    LBVH_TRY(mem.get(&d_counters, (size_t)n + 1));
    LBVH_TRY(hipMemcpyAsync(d_refs, refs, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, 0));
    LBVH_TRY(hipMemsetAsync(d_counters, 0, ((size_t)n + 1) * sizeof(unsigned int), 0));
    Grid g;
    for (int a = 0; a < 3; ++a) {
        const double ext = centre_hi[a] - centre_lo[a];
        g.lo[a] = centre_lo[a];
        g.scale[a] = (ext > 0 && ext < HUGE_VAL) ? 2097152.0 / ext : 0.0;
    }
    const unsigned blocks = (n + kBuildBlock - 1) / kBuildBlock;
    hipLaunchKernelGGL(k_sphere_boxes, dim3(blocks), dim3(kBuildBlock), 0, 0, d_prims, d_refs, n, far_allowance, g, d_boxes, d_codes, d_order);
    size_t temp_bytes = 0;
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_codes, d_codes_sorted, d_order, d_order_sorted, (int)n, 0, 63, 0));
    unsigned char *d_temp;
    LBVH_TRY(mem.get(&d_temp, temp_bytes));
    LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_codes, d_codes_sorted, d_order, d_order_sorted, (int)n, 0, 63, 0));
    hipLaunchKernelGGL(k_hierarchy, dim3(blocks), dim3(kBuildBlock), 0, 0, d_codes_sorted, (int)n, d_children, d_parent_inner, d_parent_leaf);
    hipLaunchKernelGGL(k_fit, dim3(blocks), dim3(kBuildBlock), 0, 0, d_boxes, d_order_sorted, d_refs, d_children, d_parent_inner, d_parent_leaf, (int)n,
                       d_counters, d_node_boxes, d_nodes, first_index);
    hipLaunchKernelGGL(k_depth, dim3(blocks), dim3(kBuildBlock), 0, 0, d_parent_inner, d_parent_leaf, (int)n, d_counters + n);
    // (a tree of fewer nodes than places: every node is copied, the places left over stay unused)
    if (staged) hipLaunchKernelGGL(k_stage_top, dim3(1), dim3(64), 0, 0, d_table, staged < n - 1 ? staged : n - 1, first_index);
    LBVH_TRY(hipGetLastError());
    unsigned int deepest = 0;
    LBVH_TRY(hipMemcpy(&deepest, d_counters + n, sizeof deepest, hipMemcpyDeviceToHost));   // (waits for the kernels)
    *depth_out = deepest;
#undef LBVH_TRY
    return hipSuccess;
}

}  // namespace lbvh
