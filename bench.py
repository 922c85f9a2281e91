#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5] [--spp S | --total-spp S]

One step = one pass of the hot path over one frame's worth of samples: every rank renders its
range of global sample indices of every pixel, then the float accumulation buffers are summed with
one RCCL all-reduce.  The frame is BASELINE.json configs[1]: the 1920x1080 random-spheres frame
at 1024 samples per pixel.  With one GPU it renders them all.  With N > 1 the frame's 1024 samples
are SPLIT across the ranks, rank g taking [g*S/N, (g+1)*S/N) -- strong scaling, the split the
reference's README proposes; the per-(pixel, sample) random streams make the union of the ranks'
paths the same set for any N.  `--spp S` instead gives every rank S samples of its own (weak
scaling, rank r taking [r*S, (r+1)*S)); `--total-spp S` / `--config c4` (3840x2160, S = 4096,
BASELINE.json configs[3]) split other totals.  The scene is uploaded once and stays resident in
HBM; only kernels and the all-reduce are timed.  The first frame of a view (the warm-up) also runs
the short costing pass that ranks the tiles; frames of the same view reuse that order
(`first_frame_kernel_ms` beside `roofline.kernel_ms`).

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts its N ranks itself, as
fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before anything in the
parent has touched the GPU (launch_ranks); under torch.distributed.run it finds those variables
set and is one of the ranks.

Rank 0 prints ONE JSON line.  `value` is whole-job Mpaths/s (1 path = 1 camera sample followed
through all its bounces).

`roofline` prices the kernel against the bound that binds it, the f64 vector ALU (the tree lives
in LDS and HBM is idle): `achieved` = ALGORITHMIC flops of one launch -- rays/path x (box tests/ray
x 27 + sphere tests/ray x 30 + 60), the counts being those of the REFERENCE's both-children
traversal measured by the oracle on the same scene and camera -- / the launch's duration measured
live with HIP events; `frac` = achieved / 78.6 TFLOP/s.  Beside it, from the PMC pass over the
same configuration (profiles/valu.json, keyed by configuration): `valu_slot_occupancy` (VALU
lane-slots the kernel occupied x 2 / time / peak: what the same slots would deliver if every
instruction were an f64 FMA), `lanes_active`, `valu_busy`, `occupancy`, `executed_f64_tflops`.
`algorithmic_hbm` keeps the figure that prices the reference's traversal in bytes (32 B per box
test, ...) against HBM peak: this kernel fetches none of them -- it can exceed 1.

`cpu_baseline` times the compiled reference classes (oracle/_ref/ref_capture) -- or the C
oracle when that binary is absent -- on one host core over a bounded sample of the frame.
`extra_configs` (N = 1 only) adds short runs of the other BASELINE configurations.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E
FP64_PEAK_TFLOPS = 78.6      # MI355X vector FP64 = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
LANE_SLOTS_PER_S = 256 * 4 * 16 * 2.4e9

WORKLOADS = {
    "book1": "book-1 random spheres, ~485 static spheres, lambertian/metal/dielectric (BASELINE configs[1])",
    "book2": "book-2 scene: moving spheres + BVH + chequer ground, main.cpp:61-99 as written (BASELINE configs[2])",
    "two_perlin": "two Perlin spheres through object_list, main.cpp:52-59 (the reference's default scene)",
    "stress": "synthetic sphere field, one bvh over everything (BASELINE configs[4])",
}

# BASELINE.json configs[1..4]
CONFIGS = {
    "c2": dict(scene="book1", nx=1920, ny=1080, spp=1024, total_spp=None, camera="book"),
    "c3": dict(scene="book2", nx=1920, ny=1080, spp=1024, total_spp=None, camera="book"),
    "c4": dict(scene="book1", nx=3840, ny=2160, spp=None, total_spp=4096, camera="book"),
    "c5": dict(scene="stress:1000000", nx=1920, ny=1080, spp=64, total_spp=None, camera="0,25,60,0,0,0,40"),
}


def book_camera(aspect):
    # main.cpp:114-120
    return [13.0, 2.0, 3.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 20.0, aspect, 0.0, 10.0, 0.0, 1.0]


def camera15(spec, aspect):
    cam15 = book_camera(aspect)
    if spec != "book":            # 'fx,fy,fz,ax,ay,az,vfov'
        v = [float(x) for x in spec.split(",")]
        cam15[0:6] = v[0:6]
        cam15[9] = v[6]
    return cam15


def scene_file(rt, scene):
    if scene.startswith("stress:"):          # e.g. stress:1000000 -- generated on the fly (configs[4])
        import tempfile
        path = os.path.join(tempfile.gettempdir(), f"rtiow_{scene.replace(':', '_')}_{os.getpid()}.recipe")
        rt.generate("stress", 1, path, n=int(scene.split(":")[1]))
        return path
    return scene if os.path.exists(scene) else os.path.join(SCENES, scene + ".recipe")


def algorithmic_counts(scene_path, cam15):
    """Per-path work of the reference's traversal, measured by the oracle on a small frame of
    the same scene and camera (the per-path averages do not depend on resolution)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    s = oracle_lib.Scene(scene_path)
    nx, ny, ns = 192, 108, 4
    _, c = s.render(cam15, nx, ny, 0, ns, 1, oracle_lib.ORC_RNG_STREAM)
    paths = float(c.paths)
    return {"rays_per_path": c.rays / paths, "box_tests_per_ray": c.box_tests / c.rays,
            "sphere_tests_per_ray": c.sphere_tests / c.rays, "moving_tests_per_ray": c.moving_tests / c.rays}


def _ref_render(binary, scene_path, cam15, nx, ny, spp, seed):
    return subprocess.Popen([binary, "render", scene_path, str(seed), str(nx), str(ny), str(spp), *map(repr, cam15), "-"],
                            stdout=subprocess.PIPE, text=True)


def cpu_baseline(scene_path, cam15, nx, ny):
    """The reference renderer on the host over a bounded sample of the same frame: one core
    with the reference's own flags (the headline `value`), one core without the debug-mode
    standard library, and one process per available core, each rendering the frame with its
    own seed (the parallelisation the reference's README suggests)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_capture")
    spp = 2
    if os.path.exists(ref):
        r = json.loads(_ref_render(ref, scene_path, cam15, nx, ny, spp, 1).communicate()[0])
        out = {"value": r["mpaths_per_s"], "unit": "Mpaths/s", "cores": 1, "kind": "reference",
               "sample": f"{nx}x{ny} at {spp} spp ({nx * ny * spp / 1e6:.2f} M paths, {r['render_seconds']:.1f} s), "
                         "reference classes built with the reference's flags (-O3 -D_GLIBCXX_DEBUG -DNDEBUG), single thread as shipped"}
        fast = ref + "_nodebug"
        if os.path.exists(fast):
            r = json.loads(_ref_render(fast, scene_path, cam15, nx, ny, spp, 1).communicate()[0])
            out["without_debug_stdlib"] = {"value": r["mpaths_per_s"], "cores": 1, "sample": f"same frame, -O3 -DNDEBUG, {r['render_seconds']:.1f} s"}
            cores = min(16, len(os.sched_getaffinity(0)))
            t0 = time.perf_counter()
            procs = [_ref_render(fast, scene_path, cam15, nx, ny, 1, 100 + k) for k in range(cores)]
            outs = [pr.communicate()[0] for pr in procs]
            dt = time.perf_counter() - t0
            if all(pr.returncode == 0 for pr in procs):
                out["all_cores"] = {"value": sum(json.loads(o)["mpaths_per_s"] for o in outs), "cores": cores,
                                    "sample": f"{cores} concurrent processes x ({nx}x{ny} at 1 spp, own seed), -O3 -DNDEBUG, "
                                              f"sum of their rates, wall {dt:.1f} s"}
        return out
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    s = oracle_lib.Scene(scene_path)
    t0 = time.perf_counter()
    s.render(cam15, nx, ny, 0, 1, 1, oracle_lib.ORC_RNG_MT)
    dt = time.perf_counter() - t0
    return {"value": nx * ny / dt / 1e6, "unit": "Mpaths/s", "cores": 1, "kind": "port",
            "sample": f"{nx}x{ny} at 1 spp ({dt:.1f} s), C restatement of the reference (oracle/rt_oracle.c)"}


def image_check(rt, world, scene):
    """RMSE of a 400x200, 1024-spp GPU frame against a frame the reference rendered at the same
    size and sample count (tests/golden/frames_1024.npz), beside the RMSE between two reference
    seeds: two independent estimates of the same image differ by about that much."""
    path = os.path.join(ROOT, "tests", "golden", "frames_1024.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    if scene + "_spp1024_seed1" not in g.files:
        return None
    a, b = g[scene + "_spp1024_seed1"].astype(np.float64), g[scene + "_spp1024_seed2"].astype(np.float64)
    sums, _ = world.render(rt.camera(g["camera"]), 400, 200, 0, 1024, 77)
    img = sums / 1024
    return {"rmse_vs_reference": float(np.sqrt(np.mean((img - a) ** 2))), "rmse_between_reference_seeds": float(np.sqrt(np.mean((a - b) ** 2))),
            "mean_vs_reference": [float(x) for x in (img.mean(axis=(0, 1)) - 0.5 * (a + b).mean(axis=(0, 1)))],
            "frame": "400x200, 1024 spp, linear radiance"}


def sample_range(rank, spp, world=1, total_spp=None):
    """Global sample indices [begin, end) that rank `rank` renders of every pixel: disjoint
    across ranks, and -- the streams being keyed by (pixel, global sample) -- together the same
    set of paths a single GPU would trace.  Weak scaling: `spp` per rank.  Strong scaling
    (`total_spp` given): the frame's total_spp samples split [g*S/G, (g+1)*S/G)."""
    if total_spp is not None:
        return rank * total_spp // world, (rank + 1) * total_spp // world
    return rank * spp, (rank + 1) * spp


def _keyed(path, config_key):
    """The record of `config_key` in a profiles/*.json file written by tools/roofline_inputs.py."""
    if not os.path.exists(path):
        return None
    return json.load(open(path)).get("configs", {}).get(config_key)


def roofline(config_key, counts, launch_paths, my_spp, kernel_ms):
    """The `roofline` object of one configuration: algorithmic flops of a launch / its measured
    duration against the f64 vector peak, with the PMC figures of the same configuration beside it
    when tools/profile_round.sh collected them (profiles/valu.json, profiles/traffic.json)."""
    k_s = float(np.mean(kernel_ms)) / 1e3
    flops_per_path = counts["rays_per_path"] * (counts["box_tests_per_ray"] * 27 + (counts["sphere_tests_per_ray"] +
                                                counts["moving_tests_per_ray"]) * 30 + 60)
    bytes_per_path = counts["rays_per_path"] * (counts["box_tests_per_ray"] * 32 + counts["sphere_tests_per_ray"] * 16 +
                                                counts["moving_tests_per_ray"] * 40) + 12.0 / max(1, my_spp)
    achieved = launch_paths * flops_per_path / k_s / 1e12
    out = {"bound": "valu", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
           "meaning": "algorithmic flops of one launch = paths x rays/path x (box tests/ray x 27 + sphere tests/ray x 30 + 60), counts of the "
                      "reference's both-children traversal measured by the oracle on this scene and camera, / the launch's duration (HIP events), "
                      "against the f64 vector peak 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz",
           "algorithmic_flops_per_launch": launch_paths * flops_per_path, "algorithmic_flops_per_path": flops_per_path, "counts": counts,
           "config_key": config_key, "kernel": "k_render (timed frames reuse the warm-up frame's tile order: one launch per frame)",
           "kernel_ms": k_s * 1e3, "kernel_ms_min": float(np.min(kernel_ms)), "kernel_ms_steps": [round(float(x), 3) for x in kernel_ms]}
    v = _keyed(os.path.join(ROOT, "profiles", "valu.json"), config_key)
    if v is None:
        out["pmc"] = f"no PMC pass over '{config_key}' in profiles/valu.json"
    else:
        slots = v["SQ_THREAD_CYCLES_VALU"]
        out["valu_slot_occupancy"] = 2.0 * slots / k_s / 1e12 / FP64_PEAK_TFLOPS
        out["lane_slots_per_launch"] = slots
        out["pmc"] = "profiles/valu.json <- " + v.get("source", "?")
        for k in ("lanes_active", "valu_busy", "occupancy", "valu_insts_per_launch", "f64_flops_per_launch", "pmc_kernel_ms"):
            if k in v:
                out[k] = v[k]
        if "f64_flops_per_launch" in v:
            out["executed_f64_tflops"] = v["f64_flops_per_launch"] / k_s / 1e12
    # HBM bytes of one launch from the PMC passes (FETCH_SIZE doubled for gfx950 + WRITE_SIZE)
    t = _keyed(os.path.join(ROOT, "profiles", "traffic.json"), config_key)
    traffic = t["hbm_bytes_per_launch"] if t else None
    alg_gbs = launch_paths * bytes_per_path / k_s / 1e9
    out.update({"traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC, profiles/traffic.json)",
                "hbm_frac": (traffic / k_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "algorithmic_hbm": {"what": "reference-traversal bytes (32 B per box test, 16 B per static-sphere test, 40 B per moving-sphere test, 12 B of "
                                            "framebuffer per pixel) / kernel time against HBM peak; the kernel fetches none of them when the tree is in LDS",
                                    "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS,
                                    "algorithmic_bytes_per_launch": launch_paths * bytes_per_path, "algorithmic_bytes_per_path": bytes_per_path}})
    return out


class Ranks:
    """The process group of one run: barrier, all-reduce of a frame's sums, max over ranks."""

    def __init__(self, torch, dist, world):
        self.torch, self.dist, self.world = torch, dist, world

    def fence(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_seconds(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def timed_frames(ranks, rt, world_obj, cam, params, accum, stream, steps, warmup):
    """`warmup` untimed and `steps` timed frames: each rank renders its samples, then the sums are
    all-reduced (RCCL, ordered after the render on the same stream).  The timed frames are
    bracketed by a barrier and a device synchronisation on both sides; the time is the maximum over
    the ranks.  Returns a dict with the wall seconds, per-frame kernel and all-reduce milliseconds
    and the last frame's statistics."""
    torch = ranks.torch
    kernel_ms, reduce_events, first_ms, last = [], [], None, None

    def step(timed):
        accum.zero_()
        stats = rt.rt_stats()
        world_obj.render_device(cam, params, accum.data_ptr(), stream, stats)
        if ranks.world > 1:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ranks.dist.all_reduce(accum)
            b.record()
            if timed:
                reduce_events.append((a, b))
        if timed:
            kernel_ms.append(stats.kernel_ms)
        return stats

    for k in range(warmup):
        s = step(False)
        if k == 0:
            first_ms = s.kernel_ms
    ranks.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step(True)
    ranks.fence()
    elapsed = ranks.max_seconds(time.perf_counter() - t0)
    return {"seconds": elapsed, "kernel_ms": kernel_ms, "all_reduce_ms": [a.elapsed_time(b) for a, b in reduce_events],
            "first_frame_kernel_ms": first_ms, "last": last}


def run_config(ranks, rt, world_obj, rank, name, scene, nx, ny, cam_spec, spp, total_spp, steps, warmup, setup=None):
    """One short run of another configuration; every rank takes part, rank 0 gets the record."""
    torch = ranks.torch
    cam = rt.camera(camera15(cam_spec, nx / ny))
    accum = torch.zeros(ny, nx, 3, dtype=torch.float32, device="cuda")
    s_begin, s_end = sample_range(rank, spp, ranks.world, total_spp)
    frame_spp = total_spp if total_spp is not None else spp * ranks.world
    params = world_obj.params(nx, ny, s_begin, s_end, seed=1, accum=rt.RT_ACCUM_F32)
    t = timed_frames(ranks, rt, world_obj, cam, params, accum, torch.cuda.current_stream().cuda_stream, steps, warmup)
    paths = nx * ny * frame_spp
    rec = {"config": name, "scene": scene, "nx": nx, "ny": ny, "spp_per_gpu": s_end - s_begin, "spp_per_frame": frame_spp,
           "scaling": "strong" if total_spp is not None else "weak", "value": paths * steps / t["seconds"] / 1e6, "unit": "Mpaths/s",
           "ms_per_step": t["seconds"] / steps * 1e3, "kernel_ms": float(np.mean(t["kernel_ms"])), "kernel_ms_steps": t["kernel_ms"],
           "rays_per_path": t["last"].rays / t["last"].paths, "mean_radiance": float(accum.mean().item()) / frame_spp,
           "sample_range": [s_begin, s_end]}
    if ranks.world > 1:
        rec["all_reduce_ms"] = float(np.mean(t["all_reduce_ms"]))
    if setup:
        rec["setup_s"] = setup
    del accum
    return rec


def extra_configs(ranks, rt, main_world, rank, local_rank):
    """Short runs of the other BASELINE configurations (1 warm-up + 2 steps each).  One GPU: the
    shares an 8-GPU split would give it, the book-2 scene, the 4K frame, the 10^6-sphere scene,
    each with its own roofline.  Several GPUs: the weak-scaling figure (1024 samples per rank) and
    the 3840x2160 frame's 4096 samples split across the ranks (BASELINE.json configs[3])."""
    c5cam = CONFIGS["c5"]["camera"]
    if ranks.world > 1:
        cases = [("c2 weak: 1920x1080, 1024 spp on every GPU", "book1", 1920, 1080, 1024, None, "book", 2),
                 (f"c4: 3840x2160, 4096 spp split over {ranks.world} GPUs", "book1", 3840, 2160, None, 4096, "book", 1)]
    else:
        cases = [("c2 at 128 spp (one GPU's share of 1024 spp split over 8)", "book1", 1920, 1080, 128, None, "book", 2),
                 ("c2 at 16 spp", "book1", 1920, 1080, 16, None, "book", 2),
                 ("c3: book-2 scene, moving spheres", "book2", 1920, 1080, 1024, None, "book", 2),
                 ("c4 frame on one GPU: 3840x2160 at 512 spp (one GPU's share of 4096 split over 8)", "book1", 3840, 2160, 512, None, "book", 2),
                 ("c5: 10^6 spheres, 64 spp (tree over the spheres built on the GPU: the default for large scenes)", "stress:1000000", 1920, 1080, 64, None, c5cam, 2),
                 ("c5: 10^6 spheres, 8 spp (one GPU's share of 64 split over 8)", "stress:1000000", 1920, 1080, 8, None, c5cam, 2),
                 ("c5: 10^6 spheres, 64 spp, tree built on the host (RTIOW_BUILD=host: surface-area clustering)", "stress:1000000@host", 1920, 1080, 64, None, c5cam, 2)]
    out = []
    worlds = {("book1", None): (main_world, None)}
    counts = {}
    for name, scene, nx, ny, spp, total_spp, cam_spec, steps in cases:
        build = None
        if "@" in scene:                         # scene@host / scene@device: where the tree over the spheres is built
            scene, build = scene.split("@")
        key = (scene, build)
        if key not in worlds and (scene, None) in worlds and build is None:
            key = (scene, None)
        if key not in worlds:
            if build:
                os.environ["RTIOW_BUILD"] = build
            t0 = time.perf_counter()
            if scene.startswith("stress:"):      # generated in memory: scene API objects -> reference BVH build -> layout -> upload
                w_new = rt.World(None, device=local_rank, generator=("stress", 1, int(scene.split(":")[1])))
            else:
                w_new = rt.World(scene_file(rt, scene), device=local_rank)
            # (the first large world of a process also pays for its fresh heap pages: ~0.1-0.15 s for 10^6 spheres)
            worlds[key] = (w_new, dict(w_new.setup_seconds(), total_s=time.perf_counter() - t0,
                                       first_large_world_of_the_process=not any(k[0].startswith("stress:") for k in worlds)))
            if build:
                del os.environ["RTIOW_BUILD"]
        w, setup = worlds[key]
        rec = run_config(ranks, rt, w, rank, name, scene, nx, ny, cam_spec, spp, total_spp, steps, 1, setup)
        if rank == 0 and ranks.world == 1:
            if (scene, cam_spec) not in counts:
                path = scene_file(rt, scene)
                counts[(scene, cam_spec)] = algorithmic_counts(path, camera15(cam_spec, nx / ny))
                if scene.startswith("stress:"):
                    os.unlink(path)
            b, e = rec["sample_range"]
            rec["roofline"] = roofline(f"{scene}{'@' + build if build else ''} {nx}x{ny} samples [{b},{e})", counts[(scene, cam_spec)], nx * ny * (e - b), e - b, rec["kernel_ms_steps"])
        out.append(rec)
    for w, setup in worlds.values():
        if setup is not None:
            w.close()
    if ranks.world == 1 and rank == 0:
        out.extend(onna_configs(local_rank))
    return out


def onna_configs(device):
    """The features that only the reference's Lisp tracer has (include/onna_hip.h): the Cornell box of onna/scenes.lisp:157-182 with
    the camera of onna/onna.lisp:9-17 and the book's final scene (scenes.lisp:208-294), 100 stratified samples a pixel,
    on_render into device memory timed by its own HIP events; beside the first the checker's rate on one host core
    (oracle/onna_oracle.c on a 120x120 frame of the same scene and camera)."""
    import tempfile
    import torch
    on = importlib.import_module("ray-tracing-in-one-weekend_amd.onna")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import onna_oracle_lib as oo   # (the checker: for the CPU figure beside the GPU's)
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        yy, xx = np.mgrid[0:256, 0:512]
        on.write_png(os.path.join(tmp, "earth.png"), np.stack([(xx // 2) % 256, yy % 256, ((xx + yy) // 3) % 256], -1).astype(np.uint8))
        for name, width in (("cornell-box", 600), ("final-scene", 400)):
            t = on.Tables(name, seed=1, image_path=os.path.join(tmp, "earth.png"))
            s = on.Scene(t, device)
            cam = t.camera(width=width, spp=100, depth=50)
            accum = torch.zeros(cam.image_height, cam.image_width, 3, dtype=torch.float64, device="cuda")
            stream = torch.cuda.current_stream().cuda_stream
            params = on.on_render_params(0, 100, 1)
            s.render_device(cam, params, accum.data_ptr(), stream)   # warm-up
            torch.cuda.synchronize()
            ms = []
            for _ in range(2):
                st = on.rt_stats()
                s.render_device(cam, params, accum.data_ptr(), stream, st)
                ms.append(st.kernel_ms)
            rec = {"config": f"onna {name}: {cam.image_width}x{cam.image_height}, 100 stratified samples, depth 50, f64 sums", "scene": name,
                   "value": st.paths / np.mean(ms) / 1e3, "unit": "Mpaths/s", "ms_per_step": float(np.mean(ms)), "kernel_ms_steps": ms, "rays_per_path": st.rays / st.paths,
                   "mean_radiance": float(accum.mean().item()) / 300, "parity": "checker unpinned (no Lisp here); GPU == checker in tests/test_onna_gpu.py"}
            # the checker on one host core over a small frame of the same scene and camera: the CPU figure, and the
            # counts of the Lisp traversal (box, sphere and quad tests per ray) for the algorithmic flops per path
            small = t.camera(width=120, spp=16, depth=50)
            t0 = time.perf_counter()
            _, c = oo.render(t, small, 1)
            dt = time.perf_counter() - t0
            rec["cpu_checker"] = {"value": c.paths / dt / 1e6, "unit": "Mpaths/s", "cores": 1, "kind": "port", "sample": f"120x120 at 16 samples ({c.paths} paths, {dt:.1f} s)"}
            per_ray = {"box_tests": c.box_tests / c.rays, "sphere_tests": c.sphere_tests / c.rays, "quad_tests": c.quad_tests / c.rays}
            # a box test 27 flops and a sphere test 30 as for the C++ path; a quad test 39 (two dot products, a division,
            # the point, two cross products and two more dot products: scene.lisp:240-255); 60 for shading a segment
            flop_path = (c.rays / c.paths) * (per_ray["box_tests"] * 27 + per_ray["sphere_tests"] * 30 + per_ray["quad_tests"] * 39 + 60)
            achieved = flop_path * st.paths / (np.mean(ms) * 1e-3) / 1e12
            rec["roofline"] = {"bound": "valu", "what": "algorithmic flops of the Lisp traversal (counts from the checker) / kernel time against the f64 vector peak",
                               "achieved": achieved, "peak": 78.6, "unit": "TFLOP/s", "frac": achieved / 78.6, "flop_per_path": flop_path, "per_ray": per_ray, "traffic": None}
            out.append(rec)
            del accum
            s.close()
    return out


def launch_ranks(n, argv, script=None, timeout=None):
    """Starts ranks 0..n-1 of `script` (this file) as fresh child processes, one per GPU, with the
    rendezvous variables torch.distributed reads, and waits for them.  The caller must not have
    touched the GPU: nothing here replaces a process that has.  Rank 0's standard output is this
    process's.  When a rank fails the others are given ten seconds and then ended (by their own
    process ids).  Returns the largest exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        # (only rank 0 reports: what another rank or its libraries print goes to standard error)
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__), *argv], env=env, stdout=None if r == 0 else sys.stderr))
    deadline = None if timeout is None else time.monotonic() + timeout
    failed_at = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.05)
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.monotonic()
        if (failed_at is not None and time.monotonic() - failed_at > 10) or (deadline is not None and time.monotonic() > deadline):
            for p in procs:
                if p.poll() is None:
                    p.kill()
            if failed_at is None:
                failed_at = time.monotonic()
    codes = [p.wait() for p in procs]
    return max((c if c >= 0 else 128 - c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="a BASELINE.json configuration; explicit flags override its fields")
    ap.add_argument("--scene", default=None, help="recipe name under tests/golden/scenes, a recipe path, or stress:N")
    ap.add_argument("--nx", type=int, default=None)
    ap.add_argument("--ny", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel per GPU per step (weak scaling)")
    ap.add_argument("--total-spp", type=int, default=None, help="samples per pixel per step in all, split across the GPUs (strong scaling; the default with several GPUs: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other configurations")
    ap.add_argument("--timed-only", action="store_true", help="profiling runs: nothing but the warm-up and the timed steps touches the GPU")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--camera", default=None, help="book (main.cpp:114-120) or 'fx,fy,fz,ax,ay,az,vfov'")
    ap.add_argument("--dump-accum", default=None, help="rank 0 writes the reduced float sums of the last step to this .npy file")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher (it has not imported torch)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, or without RANK set")

    cfg = dict(CONFIGS[args.config or "c2"])
    for k in ("scene", "nx", "ny", "camera"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    if args.total_spp is not None:
        cfg["total_spp"], cfg["spp"] = args.total_spp, None
    elif args.spp is not None:
        cfg["spp"], cfg["total_spp"] = args.spp, None
    elif world > 1 and cfg["total_spp"] is None:
        cfg["total_spp"], cfg["spp"] = cfg["spp"], None      # several GPUs split the frame's samples
    if args.timed_only:
        args.no_cpu_baseline = args.no_extra = True
    scene, nx, ny = cfg["scene"], cfg["nx"], cfg["ny"]
    strong = cfg["total_spp"] is not None

    import torch
    import torch.distributed as dist

    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    ranks = Ranks(torch, dist, world)

    rt = importlib.import_module("ray-tracing-in-one-weekend_amd")
    cam15 = camera15(cfg["camera"], nx / ny)
    t_build = time.perf_counter()
    if scene.startswith("stress:"):      # generated in memory: scene API objects -> reference BVH build -> layout -> upload
        w = rt.World(None, device=local_rank, generator=("stress", 1, int(scene.split(":")[1])))
    else:
        w = rt.World(scene_file(rt, scene), device=local_rank)
    t_build = time.perf_counter() - t_build
    scene_path = scene_file(rt, scene) if rank == 0 else None   # (the oracle and the reference renderer read the recipe)
    cam = rt.camera(cam15)

    accum = torch.zeros(ny, nx, 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    s_begin, s_end = sample_range(rank, cfg["spp"], world, cfg["total_spp"])
    my_spp = s_end - s_begin
    frame_spp = cfg["total_spp"] if strong else cfg["spp"] * world
    params = w.params(nx, ny, s_begin, s_end, seed=1, accum=rt.RT_ACCUM_F32)
    t = timed_frames(ranks, rt, w, cam, params, accum, stream, args.steps, args.warmup)
    elapsed, kernel_ms, last = t["seconds"], t["kernel_ms"], t["last"]

    paths_per_step = nx * ny * frame_spp
    value = paths_per_step * args.steps / elapsed / 1e6
    mean = float(accum.mean().item()) / frame_spp
    if rank == 0 and args.dump_accum:
        np.save(args.dump_accum, accum.cpu().numpy())
    extras = None
    if not args.no_extra:
        extras = extra_configs(ranks, rt, w, rank, local_rank)

    if rank == 0:
        counts = algorithmic_counts(scene_path, cam15)
        config_key = f"{scene} {nx}x{ny} samples [{s_begin},{s_end})"
        metric = "Mpaths/s at 1080p random-spheres" if (nx, ny) == (1920, 1080) else f"Mpaths/s at {nx}x{ny} random-spheres"
        line = {
            "metric": metric, "value": value, "unit": "Mpaths/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS.get(scene.split(':')[0], scene)} [{scene}] ({nx}x{ny}, "
                                   + (f"{frame_spp} spp per step split across {world} GPU(s)" if strong else f"{cfg['spp']} spp per GPU per step")
                                   + ", max depth 50, BVH built by the reference algorithm)",
                       "nx": nx, "ny": ny, "spp_per_gpu": my_spp, "spp_per_frame": frame_spp,
                       "parallelism": f"sample-split x{world}, RCCL all-reduce of f32 sums"},
            "ranks": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else None),
                      "launched_by": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or ("bench.py" if world > 1 else None)},
            "first_frame_kernel_ms": t["first_frame_kernel_ms"],
            "tile_order": "the warm-up frame measures the tile costs (one more short launch); the timed frames, same view, reuse its order",
            "rays_per_path": last.rays / last.paths, "mean_radiance": mean, "scene_build_upload_s": t_build, "scene_setup": w.setup_seconds(),
            "roofline": roofline(config_key, counts, nx * ny * my_spp, my_spp, kernel_ms),
        }
        if world > 1:
            line["all_reduce_ms_per_step"] = float(np.mean(t["all_reduce_ms"])) if t["all_reduce_ms"] else None
            line["all_reduce_ms_steps"] = [round(float(x), 3) for x in t["all_reduce_ms"]]
            line["all_reduce_bytes"] = nx * ny * 3 * 4
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene_path, cam15, nx, ny)
        if world == 1 and not args.timed_only:
            line["image_check"] = image_check(rt, w, scene)
        if extras is not None:
            line["extra_configs"] = extras
        print(json.dumps(line), flush=True)
    w.close()
    if scene.startswith("stress:") and scene_path:
        os.unlink(scene_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
