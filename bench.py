#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one frame's worth of samples: every rank renders
`--spp` samples of every pixel of the 1920x1080 random-spheres frame (BASELINE.json
configs[1]) with the per-(pixel, sample) random streams, rank r taking the global sample
indices [r*spp, (r+1)*spp), and the float accumulation buffers are summed with one RCCL
all-reduce (weak scaling: per-GPU work is fixed as N grows).  The scene is uploaded once
and stays resident in HBM; only kernels and the all-reduce are inside the timed region.

Rank 0 prints ONE JSON line.  `value` is whole-job Mpaths/s (1 path = 1 camera sample
followed through all its bounces).  `roofline` prices the render kernel against HBM peak
with the algorithmic bytes of reference-semantics traversal (32 B per box test, 16 B per
static-sphere test, 40 B per moving-sphere test, 12 B of framebuffer per pixel), the
per-path counts measured by the CPU oracle on the same scene and camera.  `cpu_baseline`
times the compiled reference classes (oracle/_ref/ref_capture) -- or the C oracle when
that binary is absent -- on one host core over a bounded sample of the same frame.
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
FP64_PEAK_TFLOPS = 78.6      # MI355X vector FP64


WORKLOADS = {
    "book1": "book-1 random spheres, ~485 static spheres, lambertian/metal/dielectric (BASELINE configs[1])",
    "book2": "book-2 scene: moving spheres + BVH + chequer ground, main.cpp:61-99 as written (BASELINE configs[2])",
    "two_perlin": "two Perlin spheres through object_list, main.cpp:52-59 (the reference's default scene)",
    "stress": "synthetic sphere field, one bvh over everything (BASELINE configs[4])",
}


def book_camera(aspect):
    # main.cpp:114-120
    return [13.0, 2.0, 3.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 20.0, aspect, 0.0, 10.0, 0.0, 1.0]


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
    """RMSE of a 120x60, 256-spp GPU frame against a frame the reference rendered at the same
    size and sample count (tests/golden/frames.npz), beside the RMSE between two reference
    seeds: two independent estimates of the same image differ by about that much."""
    path = os.path.join(ROOT, "tests", "golden", "frames.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    if scene + "_spp256_seed1" not in g.files:
        return None
    a, b = g[scene + "_spp256_seed1"].astype(np.float64), g[scene + "_spp256_seed2"].astype(np.float64)
    sums, _ = world.render(rt.camera(g["camera"]), 120, 60, 0, 256, 77)
    img = sums / 256
    return {"rmse_vs_reference": float(np.sqrt(np.mean((img - a) ** 2))), "rmse_between_reference_seeds": float(np.sqrt(np.mean((a - b) ** 2))),
            "frame": "120x60, 256 spp, linear radiance"}


def sample_range(rank, spp):
    """Global sample indices [begin, end) that rank `rank` renders of every pixel: disjoint
    across ranks, and -- the streams being keyed by (pixel, global sample) -- together the same
    set of paths a single GPU would trace for world*spp samples."""
    return rank * spp, (rank + 1) * spp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="book1", help="recipe name under tests/golden/scenes or a recipe path")
    ap.add_argument("--nx", type=int, default=1920)
    ap.add_argument("--ny", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024, help="samples per pixel per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the multi-rank path)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--camera", default="book", help="book (main.cpp:114-120) or 'fx,fy,fz,ax,ay,az,vfov'")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    rt = importlib.import_module("ray-tracing-in-one-weekend_amd")
    if args.scene.startswith("stress:"):          # e.g. stress:1000000 -- generated on the fly (configs[4])
        import tempfile
        scene_path = os.path.join(tempfile.gettempdir(), f"rtiow_{args.scene.replace(':', '_')}_{os.getpid()}.recipe")
        rt.generate("stress", 1, scene_path, n=int(args.scene.split(":")[1]))
    else:
        scene_path = args.scene if os.path.exists(args.scene) else os.path.join(SCENES, args.scene + ".recipe")
    cam15 = book_camera(args.nx / args.ny)
    if args.camera != "book":
        v = [float(x) for x in args.camera.split(",")]
        cam15[0:6] = v[0:6]
        cam15[9] = v[6]
    t_build = time.perf_counter()
    w = rt.World(scene_path, device=local_rank)
    t_build = time.perf_counter() - t_build
    cam = rt.camera(cam15)

    accum = torch.zeros(args.ny, args.nx, 3, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    s_begin, s_end = sample_range(rank, args.spp)
    params = w.params(args.nx, args.ny, s_begin, s_end, seed=1, accum=rt.RT_ACCUM_F32)
    kernel_ms = []

    def step(record):
        accum.zero_()
        stats = rt.rt_stats() if record else None
        w.render_device(cam, params, accum.data_ptr(), stream, stats)
        if world > 1:
            dist.all_reduce(accum)          # RCCL sum over xGMI
        if record:
            kernel_ms.append(stats.kernel_ms)
            return stats
        return None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    paths_per_step = args.nx * args.ny * args.spp * world
    value = paths_per_step * args.steps / elapsed / 1e6
    mean = float(accum.mean().item()) / (args.spp * world)

    if rank == 0:
        counts = algorithmic_counts(scene_path, cam15)
        bytes_per_path = counts["rays_per_path"] * (counts["box_tests_per_ray"] * 32 + counts["sphere_tests_per_ray"] * 16 +
                                                    counts["moving_tests_per_ray"] * 40) + 12.0 / args.spp
        flops_per_path = counts["rays_per_path"] * (counts["box_tests_per_ray"] * 27 + (counts["sphere_tests_per_ray"] +
                                                    counts["moving_tests_per_ray"]) * 30 + 60)
        k_s = float(np.mean(kernel_ms)) / 1e3
        launch_paths = args.nx * args.ny * args.spp
        achieved = launch_paths * bytes_per_path / k_s / 1e9
        # HBM bytes of one k_render launch from the PMC passes of tools/profile_round.sh (FETCH_SIZE
        # doubled for gfx950 + WRITE_SIZE); only valid for the configuration it was collected on
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and (args.scene, args.nx, args.ny, args.spp) == ("book1", 1920, 1080, 1024):
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
        line = {
            "metric": "Mpaths/s at 1080p random-spheres", "value": value, "unit": "Mpaths/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS.get(args.scene.split(':')[0], args.scene)} [{args.scene}] ({args.nx}x{args.ny}, "
                                   f"{args.spp} spp per GPU per step, max depth 50, BVH built by the reference algorithm)",
                       "nx": args.nx, "ny": args.ny, "spp_per_gpu": args.spp, "parallelism": f"sample-split x{world}, RCCL all-reduce of f32 sums"},
            "rays_per_path": last.rays / last.paths, "mean_radiance": mean, "scene_build_upload_s": t_build,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "HBM bytes per step: costing pass + main launch (PMC, profiles/traffic.json)",
                         "algorithmic_bytes_per_launch": launch_paths * bytes_per_path, "kernel": "k_render<float, lds_tree, 4 waves> (+ its costing pass and tile sort)", "kernel_ms": k_s * 1e3, "kernel_ms_min": float(np.min(kernel_ms)), "kernel_ms_steps": [round(float(x), 3) for x in kernel_ms],
                         "algorithmic_bytes_per_path": bytes_per_path, "counts": counts,
                         "valu_f64": {"achieved_tflops": launch_paths * flops_per_path / k_s / 1e12, "peak_tflops": FP64_PEAK_TFLOPS,
                                      "frac": launch_paths * flops_per_path / k_s / 1e12 / FP64_PEAK_TFLOPS}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene_path, cam15, args.nx, args.ny)
        if world == 1:
            line["image_check"] = image_check(rt, w, args.scene)
        print(json.dumps(line), flush=True)
    w.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
