#!/usr/bin/env python3
"""The GPU's unit-sphere sampler (csrc/sampler_dump) against the oracle's restatement, bit for bit."""
import sys, os, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib as orc
lanes, n = 4096, 64
subprocess.run([os.path.join(ROOT, 'ray-tracing-in-one-weekend_amd/csrc/sampler_dump'), '7', '1000', '3', str(lanes), str(n), '/tmp/s.bin'], check=True)
g = np.fromfile('/tmp/s.bin').reshape(lanes, n, 3)
bad = sum(int((orc.stream_unit_sphere(7, 1000 + i, 3, n) != g[i]).any(axis=1).sum()) for i in range(lanes))
print('mismatching points', bad, 'of', lanes * n)
