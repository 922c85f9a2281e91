"""ctypes bindings over the libraries for the Lisp tracer's features (include/onna_hip.h).

* ``libonna_hip.so``  -- kernels and C ABI (quads, boxes, instances, media, image and noise textures,
  lights, the importance-sampling integrator), gfx950;
* ``libonna_host.so`` -- the host scene API (the constructors and scene builders of the reference's
  onna/*.lisp) behind a few C functions.

Plumbing for tests, the benchmark and tools: numpy arrays in and out, no arithmetic of its own and no
CPU fallback -- a missing library is an ImportError, a missing GPU an OnnaError from every compute call.
"""
import ctypes
import os

import numpy as np

from . import _preload_hip_runtime, CSRC

HIP_LIB = os.environ.get("ONNA_HIP_LIB", os.path.join(CSRC, "libonna_hip.so"))   # override: A/B builds during tuning
HOST_LIB = os.path.join(CSRC, "libonna_host.so")

ON_OBJ_SPHERE, ON_OBJ_QUAD, ON_OBJ_LIST, ON_OBJ_BVH, ON_OBJ_TRANSLATE, ON_OBJ_ROTATE_Y, ON_OBJ_MEDIUM = range(7)
ON_MAT_LAMBERTIAN, ON_MAT_METAL, ON_MAT_DIELECTRIC, ON_MAT_DIFFUSE_LIGHT, ON_MAT_ISOTROPIC = range(5)
ON_TEX_SOLID, ON_TEX_CHEQUER, ON_TEX_IMAGE, ON_TEX_PERLIN = range(4)

# include/onna_hip.h
HIT_DTYPE = np.dtype([("hit", "<i4"), ("object", "<i4"), ("material", "<i4"), ("front_face", "<i4"), ("t", "<f8"), ("p", "<f8", 3),
                      ("normal", "<f8", 3), ("u", "<f8"), ("v", "<f8")])
assert HIT_DTYPE.itemsize == 88
OBJECT_DTYPE = np.dtype([("kind", "<u4"), ("material", "<i4"), ("a", "<i4"), ("b", "<i4"), ("v", "<f8", 18), ("box", "<f8", 6)])
assert OBJECT_DTYPE.itemsize == 208
MATERIAL_DTYPE = np.dtype([("kind", "<u4"), ("texture", "<i4"), ("albedo", "<f8", 3), ("fuzz", "<f8"), ("refraction_index", "<f8")])
TEXTURE_DTYPE = np.dtype([("kind", "<u4"), ("even", "<i4"), ("odd", "<i4"), ("table", "<i4"), ("colour", "<f8", 3), ("scale", "<f8")])

# every symbol include/onna_hip.h declares
HIP_SYMBOLS = ["on_last_error", "on_scene_create", "on_scene_destroy", "on_hit_batch", "on_texture_batch", "on_pdf_value_batch",
               "on_render", "on_render_host", "on_resolve"]


class OnnaError(RuntimeError):
    pass


class on_camera(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double * 3) for n in ("centre", "pixel_delta_u", "pixel_delta_v", "pixel00", "defocus_u", "defocus_v", "background")] + [
        ("recip_sqrt_spp", ctypes.c_double), ("image_width", ctypes.c_uint32), ("image_height", ctypes.c_uint32),
        ("sqrt_spp", ctypes.c_uint32), ("max_depth", ctypes.c_uint32)]


class on_render_params(ctypes.Structure):
    _fields_ = [("sample_begin", ctypes.c_uint32), ("sample_end", ctypes.c_uint32), ("seed", ctypes.c_uint64)]


class rt_stats(ctypes.Structure):
    _fields_ = [("paths", ctypes.c_uint64), ("rays", ctypes.c_uint64), ("kernel_ms", ctypes.c_double)]


class on_scene_desc(ctypes.Structure):
    _fields_ = [("objects", ctypes.c_void_p), ("n_objects", ctypes.c_uint32),
                ("children", ctypes.POINTER(ctypes.c_int32)), ("n_children", ctypes.c_uint32),
                ("materials", ctypes.c_void_p), ("n_materials", ctypes.c_uint32),
                ("textures", ctypes.c_void_p), ("n_textures", ctypes.c_uint32),
                ("images", ctypes.c_void_p), ("n_images", ctypes.c_uint32),
                ("pixels", ctypes.c_void_p), ("n_pixel_bytes", ctypes.c_uint64),
                ("perlins", ctypes.c_void_p), ("n_perlins", ctypes.c_uint32),
                ("world", ctypes.c_int32), ("lights", ctypes.c_int32)]


_hip = None
_host = None


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def hip():
    """The GPU library.  Raises ImportError when it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise ImportError(f"{HIP_LIB} is missing: run `make -C {CSRC}` (or __graft_entry__.build())")
        _preload_hip_runtime()
        lib = ctypes.CDLL(HIP_LIB, mode=ctypes.RTLD_GLOBAL)
        vp, sz, dbl, u32, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_uint32, ctypes.c_int32
        lib.on_last_error.restype = ctypes.c_char_p
        lib.on_scene_create.argtypes = [ctypes.POINTER(on_scene_desc), ctypes.c_int, ctypes.POINTER(vp)]
        lib.on_scene_destroy.argtypes = [vp]
        lib.on_hit_batch.argtypes = [vp, i32, vp, sz, dbl, dbl, vp, u32, vp]
        lib.on_texture_batch.argtypes = [vp, i32, vp, sz, vp]
        lib.on_pdf_value_batch.argtypes = [vp, i32, vp, sz, vp]
        lib.on_render.argtypes = [vp, ctypes.POINTER(on_camera), ctypes.POINTER(on_render_params), vp, vp, ctypes.POINTER(rt_stats)]
        lib.on_render_host.argtypes = [vp, ctypes.POINTER(on_camera), ctypes.POINTER(on_render_params), vp, ctypes.POINTER(rt_stats)]
        lib.on_resolve.argtypes = [ctypes.c_int, vp, sz, u32, vp, vp]
        _hip = lib
    return _hip


def host():
    """The host scene-API library."""
    global _host
    if _host is None:
        hip()   # libonna_host.so links against it
        if not os.path.exists(HOST_LIB):
            raise ImportError(f"{HOST_LIB} is missing: run `make -C {CSRC}`")
        lib = ctypes.CDLL(HOST_LIB)
        vp = ctypes.c_void_p
        lib.onh_last_error.restype = ctypes.c_char_p
        lib.onh_scene_build.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.POINTER(vp)]
        lib.onh_scene_desc.argtypes = [vp]
        lib.onh_scene_desc.restype = ctypes.POINTER(on_scene_desc)
        lib.onh_scene_free.argtypes = [vp]
        lib.onh_scene_camera.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(on_camera)]
        lib.onh_make_camera.argtypes = [vp, ctypes.POINTER(on_camera)]
        lib.onh_read_png.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), vp, ctypes.c_size_t]
        lib.onh_write_png.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, vp]
        i32, u32, dbl = ctypes.c_int32, ctypes.c_uint32, ctypes.c_double
        lib.onh_builder_new.restype = vp
        lib.onh_builder_free.argtypes = [vp]
        lib.onh_solid.argtypes = [vp, dbl, dbl, dbl]
        lib.onh_chequer.argtypes = [vp, dbl, i32, i32]
        lib.onh_perlin.argtypes = [vp, ctypes.c_uint64, dbl]
        lib.onh_image.argtypes = [vp, u32, u32, vp]
        lib.onh_material.argtypes = [vp, u32, i32, vp, dbl, dbl]
        lib.onh_sphere.argtypes = [vp, vp, vp, dbl, i32]
        lib.onh_quad.argtypes = [vp, vp, vp, vp, i32]
        lib.onh_box.argtypes = [vp, vp, vp, i32]
        lib.onh_list.argtypes = [vp, vp, u32]
        lib.onh_bvh.argtypes = [vp, vp, u32]
        lib.onh_translate.argtypes = [vp, i32, vp]
        lib.onh_rotate_y.argtypes = [vp, i32, dbl]
        lib.onh_medium.argtypes = [vp, i32, dbl, i32]
        lib.onh_builder_finish.argtypes = [vp, i32, i32, ctypes.POINTER(vp)]
        _host = lib
    return _host


def _check(rc, what):
    if rc:
        raise OnnaError(f"{what}: {hip().on_last_error().decode()} (code {rc})")


def _host_check(rc):
    if rc:
        raise OnnaError(host().onh_last_error().decode())


def make_camera(aspect, width, spp, depth, background, vfov, look_from, look_at, up=(0, 1, 0), defocus_angle=0.0, focus_distance=10.0):
    """make-camera (camera.lisp:19-90) on the host."""
    v = np.array([aspect, width, spp, depth, *background, vfov, *look_from, *look_at, *up, defocus_angle, focus_distance], np.float64)
    out = on_camera()
    _host_check(host().onh_make_camera(_ptr(v), ctypes.byref(out)))
    return out


def read_png(path):
    w, h = ctypes.c_uint32(), ctypes.c_uint32()
    _host_check(host().onh_read_png(str(path).encode(), ctypes.byref(w), ctypes.byref(h), None, 0))
    out = np.zeros((h.value, w.value, 3), np.uint8)
    _host_check(host().onh_read_png(str(path).encode(), ctypes.byref(w), ctypes.byref(h), _ptr(out), out.size))
    return out


def write_png(path, rgb):
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 3
    _host_check(host().onh_write_png(str(path).encode(), a.shape[1], a.shape[0], _ptr(a)))


def _vec(v):
    return np.ascontiguousarray(v, dtype=np.float64)


class Builder:
    """The constructors of the host scene API (make_sphere, make_quad, ... of onna.hpp), call by call; handles are integers."""

    def __init__(self):
        self._b = ctypes.c_void_p(host().onh_builder_new())

    def _id(self, i):
        if i < 0:
            raise OnnaError(host().onh_last_error().decode())
        return i

    def solid(self, r, g, b):
        return self._id(host().onh_solid(self._b, r, g, b))

    def chequer(self, scale, even, odd):
        return self._id(host().onh_chequer(self._b, scale, even, odd))

    def perlin(self, seed, scale):
        return self._id(host().onh_perlin(self._b, seed, scale))

    def image(self, rgb):
        a = np.ascontiguousarray(rgb, dtype=np.uint8)
        return self._id(host().onh_image(self._b, a.shape[1], a.shape[0], _ptr(a)))

    def material(self, kind, tex=-1, albedo=(0, 0, 0), fuzz=0.0, refraction_index=0.0):
        return self._id(host().onh_material(self._b, kind, tex, _ptr(_vec(albedo)), fuzz, refraction_index))

    def lambertian(self, tex):
        return self.material(ON_MAT_LAMBERTIAN, tex)

    def metal(self, albedo, fuzz=0.0):
        return self.material(ON_MAT_METAL, albedo=albedo, fuzz=fuzz)

    def dielectric(self, ri):
        return self.material(ON_MAT_DIELECTRIC, refraction_index=ri)

    def diffuse_light(self, tex):
        return self.material(ON_MAT_DIFFUSE_LIGHT, tex)

    def sphere(self, centre, radius, material=-1, to=None):
        return self._id(host().onh_sphere(self._b, _ptr(_vec(centre)), _ptr(_vec(to)) if to is not None else None, radius, material))

    def quad(self, q, u, v, material=-1):
        return self._id(host().onh_quad(self._b, _ptr(_vec(q)), _ptr(_vec(u)), _ptr(_vec(v)), material))

    def box(self, a, b, material=-1):
        return self._id(host().onh_box(self._b, _ptr(_vec(a)), _ptr(_vec(b)), material))

    def list(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        return self._id(host().onh_list(self._b, _ptr(a), len(a)))

    def bvh(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        return self._id(host().onh_bvh(self._b, _ptr(a), len(a)))

    def translate(self, obj, offset):
        return self._id(host().onh_translate(self._b, obj, _ptr(_vec(offset))))

    def rotate_y(self, obj, angle):
        return self._id(host().onh_rotate_y(self._b, obj, angle))

    def medium(self, boundary, density, tex):
        return self._id(host().onh_medium(self._b, boundary, density, tex))

    def finish(self, world, lights=-1):
        h = ctypes.c_void_p()
        _host_check(host().onh_builder_finish(self._b, world, lights, ctypes.byref(h)))
        return Tables(None, handle=h)

    def __del__(self):
        try:
            if self._b:
                host().onh_builder_free(self._b)
                self._b = ctypes.c_void_p()
        except Exception:
            pass


class Tables:
    """A scene built by the host API and flattened: the description both the GPU library and a checker can read.
    By name: a scene of scenes.lisp; by handle: what a Builder finished."""

    def __init__(self, name, seed=1, image_path=None, handle=None):
        self._h = handle if handle is not None else ctypes.c_void_p()
        if handle is None:
            _host_check(host().onh_scene_build(name.encode(), seed, str(image_path).encode() if image_path else None, ctypes.byref(self._h)))
        self.desc_ptr = host().onh_scene_desc(self._h)
        self.desc = self.desc_ptr.contents
        self.name = name

    def camera(self, width=0, spp=0, depth=0):
        out = on_camera()
        _host_check(host().onh_scene_camera(self._h, width, spp, depth, ctypes.byref(out)))
        return out

    def objects(self):
        n = self.desc.n_objects
        buf = (ctypes.c_char * (n * OBJECT_DTYPE.itemsize)).from_address(self.desc.objects)
        return np.frombuffer(buf, dtype=OBJECT_DTYPE, count=n).copy()

    def materials(self):
        n = self.desc.n_materials
        if not n:
            return np.zeros(0, MATERIAL_DTYPE)
        buf = (ctypes.c_char * (n * MATERIAL_DTYPE.itemsize)).from_address(self.desc.materials)
        return np.frombuffer(buf, dtype=MATERIAL_DTYPE, count=n).copy()

    def textures(self):
        n = self.desc.n_textures
        if not n:
            return np.zeros(0, TEXTURE_DTYPE)
        buf = (ctypes.c_char * (n * TEXTURE_DTYPE.itemsize)).from_address(self.desc.textures)
        return np.frombuffer(buf, dtype=TEXTURE_DTYPE, count=n).copy()

    def close(self):
        if self._h:
            host().onh_scene_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """A flattened scene resident on one GPU."""

    def __init__(self, tables, device=-1):
        self.tables = tables
        self._s = ctypes.c_void_p()
        _check(hip().on_scene_create(tables.desc_ptr, device, ctypes.byref(self._s)), "on_scene_create")

    def close(self):
        if self._s:
            hip().on_scene_destroy(self._s)
            self._s = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def hit(self, obj, rays, t_min, t_max, uniforms=None):
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        out = np.zeros(len(rays), HIT_DTYPE)
        k = 0
        if uniforms is not None:
            uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
            k = uniforms.shape[1]
        _check(hip().on_hit_batch(self._s, obj, _ptr(rays), len(rays), t_min, t_max, _ptr(uniforms) if uniforms is not None else None, k, _ptr(out)), "on_hit_batch")
        return out

    def texture(self, tex, uvp):
        uvp = np.ascontiguousarray(uvp, dtype=np.float64)
        out = np.zeros((len(uvp), 3), np.float64)
        _check(hip().on_texture_batch(self._s, tex, _ptr(uvp), len(uvp), _ptr(out)), "on_texture_batch")
        return out

    def pdf_value(self, obj, od):
        od = np.ascontiguousarray(od, dtype=np.float64)
        out = np.zeros(len(od), np.float64)
        _check(hip().on_pdf_value_batch(self._s, obj, _ptr(od), len(od), _ptr(out)), "on_pdf_value_batch")
        return out

    def render(self, cam, seed, s_begin=0, s_end=None):
        """Per-pixel sums over the samples [s_begin, s_end), shape (height, width, 3), plus stats."""
        if s_end is None:
            s_end = cam.sqrt_spp * cam.sqrt_spp
        p = on_render_params(s_begin, s_end, seed)
        out = np.zeros((cam.image_height, cam.image_width, 3), np.float64)
        stats = rt_stats()
        _check(hip().on_render_host(self._s, ctypes.byref(cam), ctypes.byref(p), _ptr(out), ctypes.byref(stats)), "on_render_host")
        return out, stats

    def render_device(self, cam, params, accum_ptr, stream=None, stats=None):
        _check(hip().on_render(self._s, ctypes.byref(cam), ctypes.byref(params), ctypes.c_void_p(accum_ptr), ctypes.c_void_p(stream or 0),
                               ctypes.byref(stats) if stats is not None else None), "on_render")


def resolve(accum_ptr, npix, spp, rgb8_ptr, device=-1, stream=None):
    _check(hip().on_resolve(device, ctypes.c_void_p(accum_ptr), npix, spp, ctypes.c_void_p(rgb8_ptr), ctypes.c_void_p(stream or 0)), "on_resolve")
