"""ctypes bindings over the MI355X path tracer's shared libraries.

Two libraries are loaded from ``csrc/`` next to this file:

* ``librtiow_hip.so``  -- the C ABI of ``include/rtiow_hip.h`` (HIP kernels for gfx950);
* ``librtiow_host.so`` -- the C++ host scene API (world construction, BVH build,
  camera basis, scene recipes) behind a handful of C functions.

This module is plumbing for tests and the benchmark: it moves numpy arrays across the
C ABI.  It contains no arithmetic of its own and has no CPU fallback -- if the HIP
library is missing, importing the bindings fails; if no GPU is present, every compute
call raises ``RtError``.

The directory name is not a Python identifier; import it with
``importlib.import_module("ray-tracing-in-one-weekend_amd")``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
HIP_LIB = os.environ.get("RTIOW_HIP_LIB", os.path.join(CSRC, "librtiow_hip.so"))   # override: A/B builds during tuning
HOST_LIB = os.path.join(CSRC, "librtiow_host.so")

RT_ACCUM_F32, RT_ACCUM_F64 = 0, 1
RT_E_TAPE = 4

# include/rtiow_hip.h: rt_ray is 7 doubles; rt_hit is {int32 hit, int32 prim, double t, p[3], normal[3]}
RAY_DTYPE = np.dtype("<f8")
HIT_DTYPE = np.dtype([("hit", "<i4"), ("prim", "<i4"), ("t", "<f8"), ("p", "<f8", 3), ("normal", "<f8", 3)])
assert HIT_DTYPE.itemsize == 64


class RtError(RuntimeError):
    pass


class rt_camera(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double * 3) for n in ("origin", "u", "v", "w", "lower_left_corner", "horizontal", "vertical")] + [
        ("lens_radius", ctypes.c_double), ("time0", ctypes.c_double), ("time1", ctypes.c_double)]


class rt_render_params(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_uint32), ("ny", ctypes.c_uint32), ("sample_begin", ctypes.c_uint32),
                ("sample_end", ctypes.c_uint32), ("max_depth", ctypes.c_uint32), ("accum_type", ctypes.c_uint32),
                ("t_min", ctypes.c_double), ("seed", ctypes.c_uint64)]


class rt_stats(ctypes.Structure):
    _fields_ = [("paths", ctypes.c_uint64), ("rays", ctypes.c_uint64), ("kernel_ms", ctypes.c_double)]


class rt_node_desc(ctypes.Structure):
    _fields_ = [("box_min", ctypes.c_double * 3), ("box_max", ctypes.c_double * 3), ("left", ctypes.c_int32), ("right", ctypes.c_int32)]


class rt_item_desc(ctypes.Structure):
    _fields_ = [("is_bvh", ctypes.c_uint32), ("index", ctypes.c_int32)]


class rt_scene_desc(ctypes.Structure):
    _fields_ = [("textures", ctypes.c_void_p), ("n_textures", ctypes.c_uint32),
                ("materials", ctypes.c_void_p), ("n_materials", ctypes.c_uint32),
                ("prims", ctypes.c_void_p), ("n_prims", ctypes.c_uint32),
                ("nodes", ctypes.POINTER(rt_node_desc)), ("n_nodes", ctypes.c_uint32),
                ("items", ctypes.POINTER(rt_item_desc)), ("n_items", ctypes.c_uint32),
                ("perlin_ranfloat", ctypes.c_void_p), ("perlin_perm_x", ctypes.c_void_p),
                ("perlin_perm_y", ctypes.c_void_p), ("perlin_perm_z", ctypes.c_void_p)]


# every symbol include/rtiow_hip.h declares
HIP_SYMBOLS = ["rt_device_count", "rt_last_error", "rt_scene_create", "rt_scene_destroy", "rt_trace_batch",
               "rt_aabb_hit_batch", "rt_texture_batch", "rt_scatter_batch", "rt_camera_batch", "rt_render",
               "rt_render_noalloc", "rt_render_host", "rt_render_multi", "rt_resolve"]

_hip = None
_host = None


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _preload_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch ships its own libamdhip64 (soname
    libamdhip64.so.7, the name librtiow_hip.so asks for); if the system copy were loaded
    first and torch imported later, the process would hold two runtimes and the second
    would see no GPU.  Loading torch's copy first, when torch is installed, makes every
    import order end with that single copy.  Torch itself is not imported."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def hip():
    """The GPU library.  Raises ImportError when it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise ImportError(f"{HIP_LIB} is missing: run `make -C {CSRC}` (or __graft_entry__.build())")
        _preload_hip_runtime()
        # global scope, so that librtiow_host.so binds to this copy whichever path it came from
        lib = ctypes.CDLL(HIP_LIB, mode=ctypes.RTLD_GLOBAL)
        lib.rt_last_error.restype = ctypes.c_char_p
        vp, sz, dbl, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_uint32
        lib.rt_trace_batch.argtypes = [vp, vp, sz, dbl, dbl, vp]
        lib.rt_aabb_hit_batch.argtypes = [ctypes.c_int, vp, vp, vp, vp, sz, vp]
        lib.rt_texture_batch.argtypes = [vp, u32, vp, sz, vp]
        lib.rt_scatter_batch.argtypes = [vp, vp, vp, vp, vp, u32, sz, vp, vp, vp, vp]
        lib.rt_camera_batch.argtypes = [ctypes.c_int, ctypes.POINTER(rt_camera), vp, vp, u32, sz, vp, vp]
        lib.rt_render.argtypes = [vp, ctypes.POINTER(rt_camera), ctypes.POINTER(rt_render_params), vp, vp, ctypes.POINTER(rt_stats)]
        lib.rt_render_noalloc.argtypes = [vp, ctypes.POINTER(rt_camera), ctypes.POINTER(rt_render_params), vp, vp]
        lib.rt_render_host.argtypes = [vp, ctypes.POINTER(rt_camera), ctypes.POINTER(rt_render_params), vp, ctypes.POINTER(rt_stats)]
        lib.rt_render_multi.argtypes = [ctypes.POINTER(vp), u32, ctypes.POINTER(rt_camera), ctypes.POINTER(rt_render_params), vp, ctypes.POINTER(rt_stats)]
        lib.rt_resolve.argtypes = [ctypes.c_int, vp, sz, u32, vp, vp]
        lib.rt_scene_destroy.argtypes = [vp]
        _hip = lib
    return _hip


def host():
    """The host scene-API library."""
    global _host
    if _host is None:
        hip()  # librtiow_host.so links against it
        if not os.path.exists(HOST_LIB):
            raise ImportError(f"{HOST_LIB} is missing: run `make -C {CSRC}`")
        lib = ctypes.CDLL(HOST_LIB)
        lib.rth_last_error.restype = ctypes.c_char_p
        lib.rth_generate.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p]
        lib.rth_world_load.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.rth_recipe_copy.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        lib.rth_world_generate.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.rth_world_tables.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        lib.rth_world_free.argtypes = [ctypes.c_void_p]
        lib.rth_world_scene.argtypes = [ctypes.c_void_p]
        lib.rth_world_scene.restype = ctypes.c_void_p
        lib.rth_world_desc.argtypes = [ctypes.c_void_p]
        lib.rth_world_desc.restype = ctypes.POINTER(rt_scene_desc)
        lib.rth_world_seconds.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        lib.rth_world_recipe_ids.argtypes = [ctypes.c_void_p]
        lib.rth_world_recipe_ids.restype = ctypes.POINTER(ctypes.c_int32)
        lib.rth_camera.argtypes = [ctypes.c_void_p, ctypes.POINTER(rt_camera)]
        lib.rth_perlin_tables.argtypes = [ctypes.c_void_p] * 4
        lib.rth_write_image.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _host = lib
    return _host


def _check(rc, what):
    if rc:
        raise RtError(f"{what}: {hip().rt_last_error().decode()} (code {rc})")


def generate(kind, seed, path, n=0):
    """Writes a scene recipe: two_perlin, book1, book2 or stress (n spheres)."""
    if host().rth_generate(kind.encode(), seed, n, path.encode()):
        raise RtError(host().rth_last_error().decode())


def recipe_copy(src, dst):
    """Loads a recipe and saves it again in the writer's own formatting."""
    if host().rth_recipe_copy(str(src).encode(), str(dst).encode()):
        raise RtError(host().rth_last_error().decode())


def camera(cam15):
    """camera::camera on the host: from(3) at(3) up(3) vfov aspect aperture focus_dist t0 t1."""
    a = np.ascontiguousarray(cam15, dtype=np.float64)
    assert a.shape == (15,)
    out = rt_camera()
    if host().rth_camera(_ptr(a), ctypes.byref(out)):
        raise RtError(host().rth_last_error().decode())
    return out


def write_image(path, sums, ns, fmt="p3"):
    """frame::write_ppm / write_ppm_binary / write_png over per-pixel sums [ny][nx][3] of ns samples."""
    a = np.ascontiguousarray(sums, dtype=np.float32)
    assert a.ndim == 3 and a.shape[2] == 3
    if host().rth_write_image(str(path).encode(), fmt.encode(), a.shape[1], a.shape[0], int(ns), _ptr(a)):
        raise RtError(host().rth_last_error().decode())


def perlin_tables():
    ran = np.zeros(256, np.float64)
    px, py, pz = (np.zeros(256, np.uint8) for _ in range(3))
    host().rth_perlin_tables(_ptr(ran), _ptr(px), _ptr(py), _ptr(pz))
    return ran, px, py, pz


def _recipe_ids(h):
    n = host().rth_world_desc(h).contents.n_prims
    if not n:
        return np.zeros(0, np.int32)
    p = host().rth_world_recipe_ids(h)
    if not p:
        raise RtError(host().rth_last_error().decode())
    return np.ctypeslib.as_array(p, shape=(n,)).copy()


def _tables(desc, ids):
    """nodes[n,8] = box min(3) max(3) left right with primitive children as ~(recipe number)."""
    nodes = np.zeros((desc.n_nodes, 8), np.float64)
    ref = lambda c: c if c >= 0 else ~int(ids[~c])
    for i in range(desc.n_nodes):
        n = desc.nodes[i]
        nodes[i] = [*n.box_min, *n.box_max, ref(n.left), ref(n.right)]
    items = np.array([(desc.items[i].is_bvh, desc.items[i].index) for i in range(desc.n_items)], dtype=np.int64).reshape(-1, 2)
    return nodes, items


def world_tables(recipe_path):
    """Builds the recipe's world on the host (BVH included) and returns (nodes[n,8], items[m,2])
    of the flattened description; no GPU is touched."""
    h = ctypes.c_void_p()
    if host().rth_world_tables(recipe_path.encode(), ctypes.byref(h)):
        raise RtError(host().rth_last_error().decode())
    try:
        return _tables(host().rth_world_desc(h).contents, _recipe_ids(h))
    finally:
        host().rth_world_free(h)


def aabb_hit_batch(cases, device=-1):
    """aabb::hit; cases[n,14] = box min(3) max(3), ray origin(3) direction(3), tmin, tmax."""
    c = np.ascontiguousarray(cases, dtype=np.float64)
    n = len(c)
    boxes = np.ascontiguousarray(c[:, 0:6])
    rays = np.ascontiguousarray(np.hstack([c[:, 6:12], np.zeros((n, 1))]))
    tmin, tmax = np.ascontiguousarray(c[:, 12]), np.ascontiguousarray(c[:, 13])
    out = np.zeros(n, np.uint8)
    _check(hip().rt_aabb_hit_batch(device, _ptr(boxes), _ptr(rays), _ptr(tmin), _ptr(tmax), n, _ptr(out)), "rt_aabb_hit_batch")
    return out


def camera_batch(cam, st, tape, device=-1):
    st = np.ascontiguousarray(st, dtype=np.float64)
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    n, k = len(st), tape.shape[1]
    out = np.zeros((n, 7), np.float64)
    used = np.zeros(n, np.int32)
    rc = hip().rt_camera_batch(device, ctypes.byref(cam), _ptr(st), _ptr(tape), k, n, _ptr(out), _ptr(used))
    if rc != RT_E_TAPE:
        _check(rc, "rt_camera_batch")
    return out, used


def render_multi(worlds, cam, nx, ny, s_begin, s_end, seed, accum=RT_ACCUM_F32, max_depth=50, t_min=0.001):
    """rt_render_multi: the frame's samples [s_begin, s_end) split over the GPUs that hold `worlds` (the same
    world uploaded to each), the sums merged on the first one and returned, shape (ny, nx, 3), plus stats."""
    p = rt_render_params(nx, ny, s_begin, s_end, max_depth, accum, t_min, seed)
    out = np.zeros((ny, nx, 3), np.float64 if accum == RT_ACCUM_F64 else np.float32)
    stats = rt_stats()
    handles = (ctypes.c_void_p * len(worlds))(*[w.scene for w in worlds])
    _check(hip().rt_render_multi(handles, len(worlds), ctypes.byref(cam), ctypes.byref(p), _ptr(out), ctypes.byref(stats)), "rt_render_multi")
    return out, stats


class World:
    """A recipe's world, built through the host scene API and resident on one GPU."""

    def __init__(self, recipe_path, device=-1, generator=None):
        """The world of a recipe file, or -- generator=(kind, seed, n) -- of a scene generator directly."""
        self._h = ctypes.c_void_p()
        self.device = device
        if generator is not None:
            kind, seed, n = generator
            rc = host().rth_world_generate(kind.encode(), seed, n, device, ctypes.byref(self._h))
        else:
            rc = host().rth_world_load(recipe_path.encode(), device, ctypes.byref(self._h))
        if rc:
            raise RtError(host().rth_last_error().decode())
        self.scene = host().rth_world_scene(self._h)
        self._recipe_ids = self._table_ids = None

    @property
    def recipe_ids(self):
        """table primitive -> recipe primitive (worked out on first use: a million-sphere world that is only rendered never asks)"""
        if self._recipe_ids is None:
            self._recipe_ids = _recipe_ids(self._h)
        return self._recipe_ids

    @property
    def table_ids(self):
        if self._table_ids is None:
            self._table_ids = np.argsort(self.recipe_ids).astype(np.int32)
        return self._table_ids

    def setup_seconds(self):
        """Seconds spent making the recipe (generator or file), in the scene API (objects + bvh_node build),
        flattening, and in rt_scene_create (layout + upload)."""
        out = (ctypes.c_double * 4)()
        host().rth_world_seconds(self._h, out)
        return {"recipe_s": out[0], "objects_and_bvh_build_s": out[1], "flatten_s": out[2], "scene_create_s": out[3]}

    def close(self):
        if self._h:
            host().rth_world_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tables(self):
        return _tables(host().rth_world_desc(self._h).contents, self.recipe_ids)

    def trace(self, rays, tmin, tmax):
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        out = np.zeros(len(rays), HIT_DTYPE)
        _check(hip().rt_trace_batch(self.scene, _ptr(rays), len(rays), tmin, tmax, _ptr(out)), "rt_trace_batch")
        m = out["hit"] != 0
        out["prim"][m] = self.recipe_ids[out["prim"][m]]   # report recipe numbering
        return out

    def texture(self, tex, points):
        p = np.ascontiguousarray(points, dtype=np.float64)
        out = np.zeros((len(p), 3), np.float64)
        _check(hip().rt_texture_batch(self.scene, tex, _ptr(p), len(p), _ptr(out)), "rt_texture_batch")
        return out

    def scatter(self, prim, rays, hits, tape):
        prim = np.ascontiguousarray(self.table_ids[np.asarray(prim, dtype=np.int64)], dtype=np.int32)   # recipe -> table numbering
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        tape = np.ascontiguousarray(tape, dtype=np.float64)
        n, k = len(prim), tape.shape[1]
        h = np.zeros(n, HIT_DTYPE)
        h["hit"] = 1
        h["prim"] = prim
        h["t"] = hits[:, 0]
        h["p"] = hits[:, 1:4]
        h["normal"] = hits[:, 4:7]
        scattered = np.zeros(n, np.uint8)
        att = np.zeros((n, 3), np.float64)
        out = np.zeros((n, 7), np.float64)
        used = np.zeros(n, np.int32)
        rc = hip().rt_scatter_batch(self.scene, _ptr(prim), _ptr(rays), _ptr(h), _ptr(tape), k, n, _ptr(scattered), _ptr(att), _ptr(out), _ptr(used))
        if rc != RT_E_TAPE:
            _check(rc, "rt_scatter_batch")
        return scattered, att, out, used

    def params(self, nx, ny, s_begin, s_end, seed, accum=RT_ACCUM_F32, max_depth=50, t_min=0.001):
        return rt_render_params(nx, ny, s_begin, s_end, max_depth, accum, t_min, seed)

    def render(self, cam, nx, ny, s_begin, s_end, seed, accum=RT_ACCUM_F64, max_depth=50, t_min=0.001):
        """Per-pixel SUMS over samples [s_begin, s_end), shape (ny, nx, 3), top row first, plus stats."""
        p = self.params(nx, ny, s_begin, s_end, seed, accum, max_depth, t_min)
        out = np.zeros((ny, nx, 3), np.float64 if accum == RT_ACCUM_F64 else np.float32)
        stats = rt_stats()
        _check(hip().rt_render_host(self.scene, ctypes.byref(cam), ctypes.byref(p), _ptr(out), ctypes.byref(stats)), "rt_render_host")
        return out, stats

    def render_device(self, cam, params, accum_ptr, stream=None, stats=None):
        """rt_render into DEVICE memory (an integer address), asynchronously on `stream`."""
        _check(hip().rt_render(self.scene, ctypes.byref(cam), ctypes.byref(params), ctypes.c_void_p(accum_ptr),
                               ctypes.c_void_p(stream or 0), ctypes.byref(stats) if stats is not None else None), "rt_render")

    def render_device_noalloc(self, cam, params, accum_ptr, stream=None):
        """rt_render_noalloc: the same frame out of the library's per-GPU scratch array, no allocation, no statistics."""
        _check(hip().rt_render_noalloc(self.scene, ctypes.byref(cam), ctypes.byref(params), ctypes.c_void_p(accum_ptr),
                                       ctypes.c_void_p(stream or 0)), "rt_render_noalloc")
