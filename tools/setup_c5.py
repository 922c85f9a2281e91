#!/usr/bin/env python3
"""Set-up seconds by phase and render milliseconds of the 10^6-sphere scene (BASELINE.json configs[4]) under
the tree build named on the command line: tools/setup_c5.py [host|device] [spp]  (run on the GPU box)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
if len(sys.argv) > 1:
    os.environ["RTIOW_BUILD"] = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rt = importlib.import_module("ray-tracing-in-one-weekend_amd")
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")                     # the HIP context exists before the clock starts, as in bench.py's extra configurations
out = []
for attempt in range(3):
    t0 = time.perf_counter()
    w = rt.World(None, device=0, generator=("stress", 1, 1000000))
    total = time.perf_counter() - t0
    cam = rt.camera(bench.camera15(bench.CONFIGS["c5"]["camera"], 1920 / 1080))
    accum = torch.zeros(1080, 1920, 3, dtype=torch.float32, device="cuda")
    params = w.params(1920, 1080, 0, spp, seed=1, accum=rt.RT_ACCUM_F32)
    ms = []
    for k in range(3):
        accum.zero_()
        st = rt.rt_stats()
        w.render_device(cam, params, accum.data_ptr(), torch.cuda.current_stream().cuda_stream, st)
        ms.append(st.kernel_ms)
    out.append(dict(w.setup_seconds(), total_s=total, first_frame_ms=ms[0], frame_ms=min(ms[1:]), rays_per_path=st.rays / st.paths, mean=float(accum.mean()) / spp))
    w.close()
print(json.dumps({"build": os.environ.get("RTIOW_BUILD", "default"), "spp": spp, "runs": out}))
