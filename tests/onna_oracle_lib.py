"""ctypes wrapper over the part of oracle/liboracle.so that restates the reference's Lisp tracer
(oracle/onna_oracle.c).  Test infrastructure: imported only by tests, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.  It reads the flat scene description that
the product's HOST library builds (the boundary's tables): the per-ray arithmetic is its own.
"""
import ctypes

import numpy as np

import oracle_lib

HIT_DTYPE = np.dtype([("hit", "<i4"), ("object", "<i4"), ("material", "<i4"), ("front_face", "<i4"), ("t", "<f8"), ("p", "<f8", 3),
                      ("normal", "<f8", 3), ("u", "<f8"), ("v", "<f8")])


class Counters(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("paths", "rays", "box_tests", "sphere_tests", "quad_tests")]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_ready = False


def lib():
    global _ready
    L = oracle_lib.lib()
    if not _ready:
        vp, sz, dbl, u32, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_uint32, ctypes.c_int32
        L.onc_hit_batch.argtypes = [vp, i32, vp, sz, dbl, dbl, vp, u32, vp]
        L.onc_texture_batch.argtypes = [vp, i32, vp, sz, vp]
        L.onc_pdf_value_batch.argtypes = [vp, i32, vp, sz, vp]
        L.onc_random_direction_batch.argtypes = [vp, i32, vp, vp, u32, sz, vp]
        L.onc_render.argtypes = [vp, vp, vp, vp, ctypes.POINTER(Counters)]
        L.onc_camera_ray.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, u32, u32, vp]
        L.onc_resolve.argtypes = [vp, sz, u32, vp]
        _ready = True
    return L


def _desc(tables):
    return ctypes.cast(tables.desc_ptr, ctypes.c_void_p)


def hit(tables, obj, rays, t_min, t_max, uniforms=None):
    rays = np.ascontiguousarray(rays, dtype=np.float64)
    out = np.zeros(len(rays), HIT_DTYPE)
    k = 0
    if uniforms is not None:
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        k = uniforms.shape[1]
    lib().onc_hit_batch(_desc(tables), obj, _ptr(rays), len(rays), t_min, t_max, _ptr(uniforms) if uniforms is not None else None, k, _ptr(out))
    return out


def texture(tables, tex, uvp):
    uvp = np.ascontiguousarray(uvp, dtype=np.float64)
    out = np.zeros((len(uvp), 3), np.float64)
    lib().onc_texture_batch(_desc(tables), tex, _ptr(uvp), len(uvp), _ptr(out))
    return out


def pdf_value(tables, obj, od):
    od = np.ascontiguousarray(od, dtype=np.float64)
    out = np.zeros(len(od), np.float64)
    lib().onc_pdf_value_batch(_desc(tables), obj, _ptr(od), len(od), _ptr(out))
    return out


def random_direction(tables, obj, origins, tape):
    origins = np.ascontiguousarray(origins, dtype=np.float64)
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    out = np.zeros((len(origins), 3), np.float64)
    lib().onc_random_direction_batch(_desc(tables), obj, _ptr(origins), _ptr(tape), tape.shape[1], len(origins), _ptr(out))
    return out


def render(tables, cam, seed, s_begin=0, s_end=None):
    """Per-pixel sums over the samples [s_begin, s_end), shape (height, width, 3), and the work counters."""
    if s_end is None:
        s_end = cam.sqrt_spp * cam.sqrt_spp
    params = (ctypes.c_uint32 * 4)()
    params[0], params[1] = s_begin, s_end
    ctypes.memmove(ctypes.addressof(params) + 8, ctypes.byref(ctypes.c_uint64(seed)), 8)
    out = np.zeros((cam.image_height, cam.image_width, 3), np.float64)
    c = Counters()
    lib().onc_render(_desc(tables), ctypes.cast(ctypes.byref(cam), ctypes.c_void_p), ctypes.cast(params, ctypes.c_void_p), _ptr(out), ctypes.byref(c))
    return out, c


def camera_ray(cam, i, j, si, sj, seed, pixel, sample):
    out = np.zeros(7, np.float64)
    lib().onc_camera_ray(ctypes.cast(ctypes.byref(cam), ctypes.c_void_p), i, j, si, sj, seed, pixel, sample, _ptr(out))
    return out


def resolve(sums, spp):
    a = np.ascontiguousarray(sums, dtype=np.float64)
    out = np.zeros(a.shape, np.uint8)
    lib().onc_resolve(_ptr(a), a.size, spp, _ptr(out))
    return out
