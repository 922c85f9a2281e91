// conservative_check.hip -- brute-force check of the one property the single-precision inner
// nodes rest on: whenever the reference's double-precision slab test (rtd::box_test, aabb.cpp:8-57)
// accepts a box for a ray and interval, the conservative single-precision test (rtd::node_test on
// the outward-rounded box) accepts it too, and its entry bound does not exceed the exact entry.
// Rays are aimed at the faces, edges and corners of the box and nudged by a few ulps (double and
// single), run parallel to axes (zero and negative-zero components), start on planes, and use
// intervals that end at the entry or exit distance; coordinates span scene-like and extreme scales.
// Prints the number of violations (must be 0) and how often the conservative test passes a box
// the exact one rejects (harmless, informative).   usage: conservative_check [pairs]
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include "rt_device.hpp"

using namespace rtd;

__device__ unsigned long long mix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
struct Gen {
    unsigned long long s;
    __device__ unsigned long long next() { s = mix(s); return s; }
    __device__ double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    __device__ double sym() { return unit() * 2 - 1; }
};
__device__ double nudge(double v, int ulps64, int ulps32) {
    for (int i = 0; i < (ulps64 < 0 ? -ulps64 : ulps64); ++i) v = nextafter(v, ulps64 < 0 ? -HUGE_VAL : HUGE_VAL);
    float f = (float)v;
    if (ulps32) { for (int i = 0; i < (ulps32 < 0 ? -ulps32 : ulps32); ++i) f = nextafterf(f, ulps32 < 0 ? -HUGE_VALF : HUGE_VALF); v = (double)f + (v - (double)(float)v); }
    return v;
}

__global__ void k_check(unsigned long long base, unsigned long long n, unsigned long long *counts, double *example) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Gen g = {mix(base + i)};
    const unsigned mode = (unsigned)(g.next() % 8);
    // scale of the scene: mostly like the tracer's scenes, sometimes extreme
    const double scales[8] = {1, 1, 10, 1000, 1e-3, 1e6, 1e-20, 1e20};
    const unsigned si = (unsigned)(g.next() % 8);
    const double S = scales[si];
    double box[6];
    for (int a = 0; a < 3; ++a) {
        const double c = g.sym() * S, h = g.unit() * g.unit() * S;
        box[2 * a] = c - h; box[2 * a + 1] = c + h;
    }
    if (g.next() % 16 == 0) { const int a = (int)(g.next() % 3); box[2 * a] = -HUGE_VAL; box[2 * a + 1] = HUGE_VAL; }   // the unbounded box of a bare primitive
    if (g.next() % 16 == 0) { const int a = (int)(g.next() % 3); box[2 * a + 1] = box[2 * a]; }                          // a flat box
    Ray r;
    double o[3], d[3], target[3];
    for (int a = 0; a < 3; ++a) {
        o[a] = g.sym() * S * (g.next() % 4 == 0 ? 100.0 : 3.0);
        // aim at a face, an edge or a corner: each coordinate is a plane or a point between
        const unsigned pick = (unsigned)(g.next() % 4);
        const double lo = box[2 * a], hi = box[2 * a + 1];
        double t = pick == 0 ? lo : (pick == 1 ? hi : lo + g.unit() * (hi - lo));
        if (!(t == t) || t - t != 0) t = g.sym() * S;    // infinite planes: any point
        target[a] = nudge(t, (int)(g.next() % 7) - 3, (int)(g.next() % 7) - 3);
    }
    if (mode == 1) { const int a = (int)(g.next() % 3); o[a] = box[2 * a + (g.next() & 1)]; }     // starts on a plane
    for (int a = 0; a < 3; ++a) d[a] = target[a] - o[a];
    if (mode == 2) { const int a = (int)(g.next() % 3); d[a] = (g.next() & 1) ? 0.0 : -0.0; }     // parallel to a slab
    if (mode == 3) { const int a = (int)(g.next() % 3); d[a] *= 1e-18; }                          // nearly parallel
    if (mode == 4) { const double k = ldexp(1.0, (int)(g.next() % 400) - 200); for (int a = 0; a < 3; ++a) d[a] *= k; }
    if (mode == 5) for (int a = 0; a < 3; ++a) d[a] = g.sym();                                   // anywhere
    r.o = mk(o[0], o[1], o[2]); r.d = mk(d[0], d[1], d[2]); r.time = 0;
    const FarSlack none = {0, 0, 0, 0, 0};
    const RayCtx c = make_ctx(r, none);
    // interval: the tracer's (0.001, huge), or ending / starting at the box
    double tmin = 0.001, tmax = DBL_MAX, e0;
    (void)box_test(box, c, -DBL_MAX, DBL_MAX, e0);
    const unsigned iv = (unsigned)(g.next() % 6);
    if (iv == 1) tmax = nudge(e0, (int)(g.next() % 5) - 2, 0);
    if (iv == 2) tmin = nudge(e0, (int)(g.next() % 5) - 2, 0);
    if (iv == 3) { tmin = g.sym() * 10; tmax = tmin + g.unit() * 100; }
    if (iv == 4) tmax = g.unit() * 50;
    double entry;
    const bool exact = box_test(box, c, tmin, tmax, entry);
    float bx[6];
    for (int k = 0; k < 6; ++k) bx[k] = (k & 1) ? plane_up(box[k]) : plane_down(box[k]);
    float e32;
    const bool cons = node_test<true>(c.nx8 ? bx[1] : bx[0], c.nx8 ? bx[0] : bx[1], c.ny8 ? bx[3] : bx[2], c.ny8 ? bx[2] : bx[3],
                                c.nz8 ? bx[5] : bx[4], c.nz8 ? bx[4] : bx[5], c, below(tmin), above(tmax), HUGE_VALF, e32);
    atomicAdd(counts + 0, 1ull);
    if (exact) atomicAdd(counts + 1, 1ull);
    if (cons && !exact) atomicAdd(counts + 2, 1ull);
    // the entry bound feeds "skip when the box starts beyond the best hit", a comparison with another
    // float: no float below it may reach the exact entry (+inf is a sound bound for an entry beyond
    // the largest float)
    const bool bad = exact && (!cons || ((double)e32 > entry && !(e32 == HUGE_VALF && entry > (double)FLT_MAX)));
    if (bad && si < 5 && mode != 4) atomicAdd(counts + 4, 1ull);
    if (bad && atomicAdd(counts + 3, 1ull) == 0) {
        for (int k = 0; k < 6; ++k) example[k] = box[k];
        for (int a = 0; a < 3; ++a) { example[6 + a] = o[a]; example[9 + a] = d[a]; }
        example[12] = tmin; example[13] = tmax; example[14] = entry; example[15] = e32; example[16] = cons;
    }
}

int main(int argc, char **argv) {
    const unsigned long long total = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2000000000ull;
    unsigned long long *counts;
    double *ex;
    if (hipMalloc(&counts, 64) != hipSuccess || hipMalloc(&ex, 17 * 8) != hipSuccess || hipMemset(counts, 0, 64) != hipSuccess) return 2;
    const unsigned long long chunk = 1ull << 27;
    for (unsigned long long base = 0; base < total; base += chunk) {
        const unsigned long long n = total - base < chunk ? total - base : chunk;
        hipLaunchKernelGGL(k_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, base, n, counts, ex);
    }
    unsigned long long h[5];
    double e[17];
    if (hipMemcpy(h, counts, 40, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(e, ex, sizeof e, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    std::printf("%llu pairs, %llu accepted by the exact test, %llu accepted only by the conservative test, %llu violations (%llu at scene-like scales)\n", h[0], h[1], h[2], h[3], h[4]);
    if (h[3]) {
        std::printf("box %a %a %a %a %a %a\norigin %a %a %a\ndirection %a %a %a\ninterval %a %a exact entry %a conservative entry %a passed %g\n",
                    e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], e[9], e[10], e[11], e[12], e[13], e[14], e[15], e[16]);
    }
    return h[3] ? 1 : 0;
}
