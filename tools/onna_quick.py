"""Times on_render on a few scenes of scenes.lisp (ONNA_HIP_LIB selects an A/B build of the library)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
on = importlib.import_module("ray-tracing-in-one-weekend_amd.onna")

yy, xx = np.mgrid[0:256, 0:512]
on.write_png("/tmp/onna_quick_earth.png", np.stack([(xx // 2) % 256, yy % 256, ((xx + yy) // 3) % 256], -1).astype(np.uint8))
cases = [("cornell-box", 600, 100), ("cornell-smoke", 600, 100), ("final-scene", 400, 100), ("bouncing-spheres", 400, 100)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, width, spp in cases:
    t = on.Tables(name, seed=1, image_path="/tmp/onna_quick_earth.png")
    s = on.Scene(t)
    cam = t.camera(width=width, spp=spp, depth=50)
    s.render(cam, 1, 0, 1)
    best = None
    for _ in range(2):
        _, st = s.render(cam, 1)
        best = st if best is None or st.kernel_ms < best.kernel_ms else best
    print(f"{os.path.basename(on.HIP_LIB)} {name}: {best.kernel_ms:.1f} ms, {best.paths / best.kernel_ms / 1e3:.1f} Mpaths/s, {best.rays / best.paths:.2f} rays/path", flush=True)
    s.close()
