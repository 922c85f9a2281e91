"""The N>1 protocol on CPU: world size 2 over gloo.  Each rank renders its sample range of
every pixel (here with the oracle standing in for the GPU kernel, which renders the same
per-(pixel, sample) streams), the float sums are all-reduced as bench.py does over RCCL,
and the result must equal a single rank rendering all the samples."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden, recipe


def _worker(rank, world, port, spp, out_path, total_spp=None):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cam15 = np.load(os.path.join(ROOT, "tests", "golden", "frames.npz"))["camera"]
    s0, s1 = bench.sample_range(rank, spp, world, total_spp)
    sums, _ = oracle_lib.Scene(os.path.join(ROOT, "tests", "golden", "scenes", "mixed.recipe")).render(
        cam15, 32, 16, s0, s1, 99, oracle_lib.ORC_RNG_STREAM)
    accum = torch.from_numpy(sums.astype(np.float32))
    dist.all_reduce(accum)
    if rank == 0:
        np.save(out_path, accum.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sample_ranges_partition():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        r = [bench.sample_range(k, 1024) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == world * 1024
        assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))


def test_strong_scaling_ranges_partition_the_frame():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 3, 4, 8):
        for total in (4096, 1024, 10, 7):
            r = [bench.sample_range(k, None, world, total) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


def test_two_ranks_sum_to_one_rank(orc, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "sum.npy")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, 3, out), nprocs=2, join=True)
    got = np.load(out)
    whole, _ = orc.Scene(recipe("mixed")).render(golden("frames")["camera"], 32, 16, 0, 6, 99, orc.ORC_RNG_STREAM)
    np.testing.assert_allclose(got, whole.astype(np.float32), rtol=2e-6, atol=1e-6)
    assert got.max() > 0


def test_two_ranks_split_one_frames_samples(orc, tmp_path):
    # strong scaling: the frame's 7 samples split 3 + 4 over two ranks ([g*S/G, (g+1)*S/G))
    import torch.multiprocessing as mp
    out = str(tmp_path / "sum.npy")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, None, out, 7), nprocs=2, join=True)
    got = np.load(out)
    whole, _ = orc.Scene(recipe("mixed")).render(golden("frames")["camera"], 32, 16, 0, 7, 99, orc.ORC_RNG_STREAM)
    np.testing.assert_allclose(got, whole.astype(np.float32), rtol=2e-6, atol=1e-6)


RANK_STUB = r"""
import os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1"
if "--fail" in sys.argv and rank == 1:
    sys.exit(3)
dist.init_process_group("gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if rank == 0:
    print("sum", t.item(), dist.get_world_size(), *sys.argv[1:], flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_bench_starts_its_own_ranks(tmp_path):
    # `python bench.py --gpus N` without a launcher: launch_ranks starts N fresh processes with the
    # rendezvous variables set; here they run a stand-in script over gloo (bench.py itself needs a GPU)
    import subprocess
    stub = tmp_path / "stub.py"
    stub.write_text(RANK_STUB)
    code = "import sys; sys.path.insert(0, sys.argv[1]); import bench; sys.exit(bench.launch_ranks(2, ['--x', '1'], script=sys.argv[2], timeout=120))"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code, ROOT, str(stub)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    report = [l for l in r.stdout.splitlines() if l.startswith("sum")]      # (gloo prints its own connection notes)
    assert report == ["sum 3.0 2 --x 1"]


def test_a_failing_rank_ends_the_launch(tmp_path):
    import subprocess
    import time
    stub = tmp_path / "stub.py"
    stub.write_text(RANK_STUB)
    code = "import sys; sys.path.insert(0, sys.argv[1]); import bench; sys.exit(bench.launch_ranks(2, ['--fail'], script=sys.argv[2], timeout=120))"
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-c", code, ROOT, str(stub)], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0          # rank 1's exit code 3, or rank 0 ended while it waited for its peer
    assert time.monotonic() - t0 < 100
