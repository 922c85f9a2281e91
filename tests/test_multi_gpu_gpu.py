"""The multi-GPU path on hardware, as far as one GPU allows: RCCL beside the render kernel in a
one-rank group, two ranks sharing the GPU over gloo with the real kernel on each, the strong-
scaling split of bench.py, and the 3840x2160 frame of BASELINE.json configs[3].

Child processes are started fresh (never forked from this one, which holds a HIP context) and
under a time limit, so that a hang in a collective ends the test instead of the session."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, recipe

pytestmark = pytest.mark.gpu

ENV = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")

ONE_RANK_NCCL = r'''
import ctypes, importlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rt = importlib.import_module("ray-tracing-in-one-weekend_amd")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, init_method="tcp://127.0.0.1:" + sys.argv[2], device_id=torch.device("cuda", 0))
w = rt.World(os.path.join(sys.argv[1], "tests", "golden", "scenes", "book1.recipe"), device=0)
cam = rt.camera([13.0, 2.0, 3.0, 0, 0, 0, 0, 1.0, 0, 20.0, 1920 / 1080, 0.0, 10.0, 0.0, 1.0])
accum = torch.zeros(1080, 1920, 3, dtype=torch.float32, device="cuda")   # the 24.9 MB buffer of the benchmark
params = w.params(1920, 1080, 0, 2, seed=5, accum=rt.RT_ACCUM_F32)
stream = torch.cuda.current_stream().cuda_stream
w.render_device(cam, params, accum.data_ptr(), stream, None)
alone = accum.clone()                 # ordered after the render on the same stream
dist.all_reduce(accum)                # RCCL, one rank: the sum over ranks is the buffer itself
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(accum, alone), "all_reduce over one rank changed the buffer"
assert float(accum.sum()) > 0
again = torch.zeros_like(accum)
w.render_device(cam, params, again.data_ptr(), stream, None)
torch.cuda.synchronize()
assert torch.equal(again, alone), "the frame changed after RCCL was initialised"
w.close()
dist.destroy_process_group()
print("ok", float(alone.mean()))
'''


def _port(k=0):
    return str(29600 + 2 * (os.getpid() % 1500) + k)


def test_rccl_beside_the_render_kernel_in_a_one_rank_group():
    # RCCL is loaded and initialised beside the library's HIP runtime, an all-reduce of the frame
    # buffer is queued behind rt_render on the same stream, and the buffer comes back bit for bit.
    r = subprocess.run([sys.executable, "-c", ONE_RANK_NCCL, ROOT, _port()], env=ENV, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.strip().splitlines()[-1].startswith("ok")


def test_two_ranks_on_one_gpu_sum_to_one_rank(rt, tmp_path):
    # bench.py's strong-scaling mode, world size 2, both ranks on GPU 0, sums exchanged over gloo:
    # each rank renders half of the 8 samples with the real kernel; the reduced buffer must be the
    # frame one process renders with all 8 (the halves are rounded to float before they are added).
    out = str(tmp_path / "sum.npy")
    # (plain `python bench.py --gpus 2`: bench.py starts its two ranks itself, as the driver's command does)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--scene", "book2", "--nx", "320", "--ny", "180", "--total-spp", "8", "--backend", "gloo", "--share-device",
           "--timed-only", "--dump-accum", out]
    env = {k: v for k, v in ENV.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["spp_per_gpu"] == 4 and line["config"]["spp_per_frame"] == 8
    got = np.load(out)
    cam15 = [13.0, 2.0, 3.0, 0, 0, 0, 0, 1.0, 0, 20.0, 320 / 180, 0.0, 10.0, 0.0, 1.0]
    w = rt.World(recipe("book2"))
    try:
        whole, stats = w.render(rt.camera(cam15), 320, 180, 0, 8, 1, rt.RT_ACCUM_F64)
        a, _ = w.render(rt.camera(cam15), 320, 180, 0, 4, 1, rt.RT_ACCUM_F32)
        b, _ = w.render(rt.camera(cam15), 320, 180, 4, 8, 1, rt.RT_ACCUM_F32)
    finally:
        w.close()
    assert np.array_equal(got, a + b)                                   # exactly the two ranks' float buffers, added
    np.testing.assert_allclose(got, whole, rtol=3e-7, atol=1e-6)       # = the whole frame up to float rounding
    assert abs(line["mean_radiance"] - whole.mean() / 8) < 1e-6
    assert line["ranks"]["world_size"] == 2 and line["ranks"]["launched_by"] == "bench.py" and line["all_reduce_ms_per_step"] is not None


def test_two_ranks_under_torch_distributed_run(tmp_path):
    # the same two ranks started by torch.distributed.run, as the driver's multi-GPU command does; several
    # GPUs and no sample flags: the frame's 1024 samples are split (here a small frame of the book-1 scene)
    out = str(tmp_path / "sum.npy")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _port(1), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--nx", "64", "--ny", "36", "--backend", "gloo", "--share-device", "--timed-only", "--dump-accum", out]
    r = subprocess.run(cmd, env=ENV, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["spp_per_gpu"] == 512 and line["config"]["spp_per_frame"] == 1024
    assert line["roofline"]["frac"] > 0 and line["ranks"]["launched_by"] == "torch.distributed.run"
    assert np.load(out).shape == (36, 64, 3)


def test_4k_frame_properties(rt):
    # BASELINE.json configs[3] is 3840x2160: no oracle renders that in seconds, so the checks are the
    # size-independent ones -- additivity over the sample ranges two GPUs would take, determinism,
    # ray counts, bounded radiance -- and agreement, up to Monte-Carlo noise, with the 1920x1080
    # frame of the same view at four times the samples (the same samples per image area).
    cam15 = np.array(golden("frames")["camera"])
    cam15[10] = 3840 / 2160
    cam = rt.camera(cam15)
    w = rt.World(recipe("book1"))
    try:
        a, sa = w.render(cam, 3840, 2160, 0, 2, 21, rt.RT_ACCUM_F64)
        b, sb = w.render(cam, 3840, 2160, 2, 4, 21, rt.RT_ACCUM_F64)
        ab, sab = w.render(cam, 3840, 2160, 0, 4, 21, rt.RT_ACCUM_F64)
        again, s2 = w.render(cam, 3840, 2160, 0, 4, 21, rt.RT_ACCUM_F64)
        half, sh = w.render(cam, 1920, 1080, 0, 16, 22, rt.RT_ACCUM_F64)
    finally:
        w.close()
    assert np.array_equal(ab, again) and sab.rays == s2.rays
    np.testing.assert_allclose(a + b, ab, rtol=1e-13, atol=1e-13)
    assert sa.rays + sb.rays == sab.rays and sab.paths == 3840 * 2160 * 4
    assert ab.min() >= 0 and ab.max() <= 4.0 + 1e-9
    assert 2.0 < sab.rays / sab.paths < 4.0
    # the same image at half the resolution with four times the samples: equal up to Monte-Carlo noise
    small = ab.reshape(1080, 2, 1920, 2, 3).sum(axis=(1, 3)) / 16
    assert abs(small.mean() - half.mean() / 16) < 2e-4
    coarse = lambda x: x.reshape(135, 8, 240, 8, 3).mean(axis=(1, 3))
    assert np.sqrt(np.mean((coarse(small) - coarse(half / 16)) ** 2)) < 0.01


def test_render_multi_on_the_gpus_at_hand(rt):
    # rt_render_multi, the C entry point behind render_options::devices: one scene is rt_render's frame to
    # the bit; two uploads of the world on the same GPU split the samples [g S / 2, (g + 1) S / 2) and add
    # their float sums in that GPU's buffer -- the sum of the two ranges rendered alone, to the bit, and the
    # whole range up to float rounding; ray counts add up.
    cam15 = np.array(golden("frames")["camera"])
    cam15[10] = 96 / 54
    cam = rt.camera(cam15)
    w, w2 = rt.World(recipe("book2"), device=0), rt.World(recipe("book2"), device=0)
    try:
        whole, st = w.render(cam, 96, 54, 3, 12, 9, rt.RT_ACCUM_F32)
        one, st1 = rt.render_multi([w], cam, 96, 54, 3, 12, 9)
        a, sa = w.render(cam, 96, 54, 3, 7, 9, rt.RT_ACCUM_F32)
        b, sb = w.render(cam, 96, 54, 7, 12, 9, rt.RT_ACCUM_F32)
        two, st2 = rt.render_multi([w, w2], cam, 96, 54, 3, 12, 9)
        three, st3 = rt.render_multi([w, w2, w], cam, 96, 54, 0, 2, 9)          # fewer samples than scenes: one share is empty
        d64, _ = rt.render_multi([w, w2], cam, 96, 54, 3, 12, 9, accum=rt.RT_ACCUM_F64)
        whole64, _ = w.render(cam, 96, 54, 3, 12, 9, rt.RT_ACCUM_F64)
        few, _ = w.render(cam, 96, 54, 0, 2, 9, rt.RT_ACCUM_F32)
    finally:
        w.close()
        w2.close()
    assert np.array_equal(one, whole) and st1.rays == st.rays and st1.paths == st.paths
    assert np.array_equal(two, a + b) and st2.rays == sa.rays + sb.rays == st.rays
    np.testing.assert_allclose(two, whole, rtol=3e-7, atol=1e-6)
    np.testing.assert_allclose(three, few, rtol=3e-7, atol=1e-6)
    assert st3.paths == 96 * 54 * 2
    np.testing.assert_allclose(d64, whole64, rtol=1e-13, atol=1e-13)


RCCL_IN_THE_LIBRARY = r'''
import importlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
rt = importlib.import_module("ray-tracing-in-one-weekend_amd")
cam15 = [13.0, 2.0, 3.0, 0, 0, 0, 0, 1.0, 0, 20.0, 2.0, 0.0, 10.0, 0.0, 1.0]
cam = rt.camera(cam15)
w = rt.World(os.path.join(sys.argv[1], "tests", "golden", "scenes", "book2.recipe"), device=0)
plain, st = rt.render_multi([w], cam, 64, 32, 0, 8, 5)
assert "librccl" not in open("/proc/self/maps").read(), "one GPU, no collective: RCCL must not have been loaded"
os.environ["RTIOW_FORCE_RCCL"] = "1"
forced, st2 = rt.render_multi([w], cam, 64, 32, 0, 8, 5)
again, st3 = rt.render_multi([w], cam, 64, 32, 0, 8, 5)          # the communicator is kept between calls
assert "librccl" in open("/proc/self/maps").read()
assert np.array_equal(plain, forced) and np.array_equal(plain, again) and st.rays == st2.rays == st3.rays
w.close()
print("ok", float(plain.mean()))
'''


def test_the_librarys_own_rccl_reduction_over_one_rank():
    # rt_render_multi opens librccl.so.1 itself, makes its communicators with ncclCommInitAll and merges the
    # GPUs' sums with ncclReduce.  One GPU needs none of it (and loads none of it); RTIOW_FORCE_RCCL=1 runs
    # the same calls over the one-rank communicator: the frame comes back unchanged.
    r = subprocess.run([sys.executable, "-c", RCCL_IN_THE_LIBRARY, ROOT], env=ENV, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.strip().splitlines()[-1].startswith("ok")
