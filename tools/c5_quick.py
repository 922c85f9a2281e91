#!/usr/bin/env python3
"""Render milliseconds of the 10^6-sphere frame (1080p) under the environment given: tools/c5_quick.py [spp] (run on the GPU box)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rt = importlib.import_module("ray-tracing-in-one-weekend_amd")
torch.cuda.set_device(0)
w = rt.World(None, device=0, generator=("stress", 1, 1000000))
cam = rt.camera(bench.camera15(bench.CONFIGS["c5"]["camera"], 1920 / 1080))
accum = torch.zeros(1080, 1920, 3, dtype=torch.float32, device="cuda")
params = w.params(1920, 1080, 0, spp, seed=1, accum=rt.RT_ACCUM_F32)
ms = []
for k in range(4):
    accum.zero_()
    st = rt.rt_stats()
    w.render_device(cam, params, accum.data_ptr(), torch.cuda.current_stream().cuda_stream, st)
    ms.append(round(st.kernel_ms, 3))
print({k: v for k, v in os.environ.items() if k.startswith("RTIOW_")}, "ms", ms, "Mpaths/s", round(1920 * 1080 * spp / min(ms[1:]) / 1e3))
w.close()
