"""ctypes wrapper over oracle/liboracle.so, the CPU restatement of the reference.

Test infrastructure: imported only by tests, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

ORC_RNG_MT, ORC_RNG_STREAM, ORC_RNG_TAPE = 0, 1, 2


class Counters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("box_tests", "sphere_tests", "moving_tests", "rays", "paths")]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, capture_output=True)
        L = ctypes.CDLL(LIB)
        vp, sz, dbl, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_int
        L.orc_scene_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, sz]
        L.orc_scene_load.restype = vp
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_counts.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
        L.orc_scene_nodes.argtypes = [vp, vp, vp]
        L.orc_trace_batch.argtypes = [vp, vp, sz, dbl, dbl, vp, vp, vp]
        L.orc_aabb_batch.argtypes = [vp, sz, vp]
        L.orc_texture_batch.argtypes = [vp, i32, vp, sz, vp]
        L.orc_scatter_batch.argtypes = [vp, vp, vp, vp, vp, i32, sz, vp, vp, vp, vp]
        L.orc_camera_batch.argtypes = [vp, vp, vp, i32, sz, vp, vp]
        L.orc_render.argtypes = [vp, vp, i32, i32, i32, i32, ctypes.c_uint64, i32, vp]
        L.orc_set_max_depth.argtypes = [i32]
        L.orc_counters_get.argtypes = [ctypes.POINTER(Counters)]
        L.orc_perlin_tables.argtypes = [vp, vp, vp, vp]
        L.orc_mt_seed.argtypes = [vp, ctypes.c_uint32]
        L.orc_mt_next.argtypes = [vp]
        L.orc_mt_next.restype = ctypes.c_uint32
        L.orc_mt_canonical.argtypes = [vp]
        L.orc_mt_canonical.restype = dbl
        L.orc_mt_uniform_int_0_2.argtypes = [vp]
        L.orc_philox4x32_10.argtypes = [vp, vp, vp]
        L.orc_stream_init.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]
        L.orc_stream_next.argtypes = [vp]
        L.orc_stream_next.restype = ctypes.c_uint32
        L.orc_stream_canonical.argtypes = [vp]
        L.orc_stream_unit_sphere.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, sz, vp]
        L.orc_stream_canonical.restype = dbl
        _lib = L
    return _lib


def aabb_batch(cases):
    c = np.ascontiguousarray(cases, dtype=np.float64)
    out = np.zeros(len(c), np.uint8)
    lib().orc_aabb_batch(_ptr(c), len(c), _ptr(out))
    return out


def camera_batch(cam15, st, tape):
    cam = np.ascontiguousarray(cam15, dtype=np.float64)
    st = np.ascontiguousarray(st, dtype=np.float64)
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    out = np.zeros((len(st), 7), np.float64)
    used = np.zeros(len(st), np.int32)
    lib().orc_camera_batch(_ptr(cam), _ptr(st), _ptr(tape), tape.shape[1], len(st), _ptr(out), _ptr(used))
    return out, used


def stream_unit_sphere(seed, pixel, sample, n):
    """n points of random_in_unit_sphere drawn from the per-path stream (seed, pixel, sample)."""
    out = np.zeros((n, 3), np.float64)
    lib().orc_stream_unit_sphere(seed, pixel, sample, n, _ptr(out))
    return out


def perlin_tables():
    ran = np.zeros(256, np.float64)
    px, py, pz = (np.zeros(256, np.uint8) for _ in range(3))
    lib().orc_perlin_tables(_ptr(ran), _ptr(px), _ptr(py), _ptr(pz))
    return ran, px, py, pz


class Scene:
    def __init__(self, recipe_path):
        err = ctypes.create_string_buffer(512)
        self.h = lib().orc_scene_load(recipe_path.encode(), err, 512)
        if not self.h:
            raise RuntimeError(err.value.decode())

    def __del__(self):
        try:
            if self.h:
                lib().orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    def counts(self):
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        lib().orc_scene_counts(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    def nodes(self):
        _, n, _ = self.counts()
        boxes = np.zeros((n, 6), np.float64)
        children = np.zeros((n, 2), np.int32)
        lib().orc_scene_nodes(self.h, _ptr(boxes), _ptr(children))
        return np.hstack([boxes, children.astype(np.float64)])

    def trace(self, rays, tmin, tmax):
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        n = len(rays)
        hit = np.zeros(n, np.uint8)
        prim = np.zeros(n, np.int32)
        tpn = np.zeros((n, 7), np.float64)
        lib().orc_trace_batch(self.h, _ptr(rays), n, tmin, tmax, _ptr(hit), _ptr(prim), _ptr(tpn))
        return hit, prim, tpn

    def texture(self, tex, points):
        p = np.ascontiguousarray(points, dtype=np.float64)
        out = np.zeros((len(p), 3), np.float64)
        lib().orc_texture_batch(self.h, tex, _ptr(p), len(p), _ptr(out))
        return out

    def scatter(self, prim, rays, hits, tape):
        prim = np.ascontiguousarray(prim, dtype=np.int32)
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        hits = np.ascontiguousarray(hits, dtype=np.float64)
        tape = np.ascontiguousarray(tape, dtype=np.float64)
        n = len(prim)
        sc = np.zeros(n, np.uint8)
        att = np.zeros((n, 3), np.float64)
        out = np.zeros((n, 7), np.float64)
        used = np.zeros(n, np.int32)
        lib().orc_scatter_batch(self.h, _ptr(prim), _ptr(rays), _ptr(hits), _ptr(tape), tape.shape[1], n, _ptr(sc), _ptr(att), _ptr(out), _ptr(used))
        return sc, att, out, used

    def render(self, cam15, nx, ny, s_begin, s_end, seed, mode, max_depth=50):
        """Per-pixel sums over the samples, (ny, nx, 3), top row first; also the work counters."""
        cam = np.ascontiguousarray(cam15, dtype=np.float64)
        out = np.zeros((ny, nx, 3), np.float64)
        lib().orc_counters_reset()
        lib().orc_set_max_depth(max_depth)
        try:
            lib().orc_render(self.h, _ptr(cam), nx, ny, s_begin, s_end, seed, mode, _ptr(out))
        finally:
            lib().orc_set_max_depth(50)
        c = Counters()
        lib().orc_counters_get(ctypes.byref(c))
        return out, c
