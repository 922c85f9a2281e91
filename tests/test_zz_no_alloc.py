"""rt_render_noalloc: the entry point that renders out of the library's per-GPU scratch array and
allocates nothing.  Its frames must be rt_render's frames bit for bit -- here on two calls in a
row, the second with a smaller input, so that what the first call left in the array (counters,
tile costs and order) must not leak into the second -- and its limits must be reported."""
import ctypes
import os

import numpy as np
import pytest

from conftest import golden, recipe

pytestmark = pytest.mark.gpu


def _frame(rt, w, cam, nx, ny, s0, s1, seed, accum_type, noalloc):
    import torch
    dtype = torch.float32 if accum_type == rt.RT_ACCUM_F32 else torch.float64
    accum = torch.zeros(ny, nx, 3, dtype=dtype, device="cuda")
    params = w.params(nx, ny, s0, s1, seed=seed, accum=accum_type)
    stream = torch.cuda.current_stream().cuda_stream
    if noalloc:
        w.render_device_noalloc(cam, params, accum.data_ptr(), stream)
    else:
        w.render_device(cam, params, accum.data_ptr(), stream, None)
    torch.cuda.synchronize()
    return accum.cpu().numpy()


@pytest.mark.parametrize("scene", ["book1", "book2"])
def test_noalloc_frames_equal_rt_render_frames(rt, scene):
    cam15 = np.array(golden("frames")["camera"])
    w = rt.World(recipe(scene))
    try:
        # first call: a deep frame with many tiles (costing pass, ordered tiles, 4x4 tiles)
        cam15[10] = 960 / 540
        cam = rt.camera(cam15)
        big = _frame(rt, w, cam, 960, 540, 0, 48, 7, rt.RT_ACCUM_F32, True)
        assert np.array_equal(big, _frame(rt, w, cam, 960, 540, 0, 48, 7, rt.RT_ACCUM_F32, False))
        assert big.any()
        # second call, right after: a smaller, ragged frame, also ordered, other sample range
        cam15[10] = 333 / 217
        cam = rt.camera(cam15)
        small = _frame(rt, w, cam, 333, 217, 5, 40, 8, rt.RT_ACCUM_F64, True)
        assert np.array_equal(small, _frame(rt, w, cam, 333, 217, 5, 40, 8, rt.RT_ACCUM_F64, False))
        # third: a shallow frame (no costing pass, 8x8 tiles), then the first one again
        tiny = _frame(rt, w, cam, 65, 33, 0, 2, 9, rt.RT_ACCUM_F32, True)
        assert np.array_equal(tiny, _frame(rt, w, cam, 65, 33, 0, 2, 9, rt.RT_ACCUM_F32, False))
        cam15[10] = 960 / 540
        assert np.array_equal(big, _frame(rt, w, rt.camera(cam15), 960, 540, 0, 48, 7, rt.RT_ACCUM_F32, True))
    finally:
        w.close()


def test_noalloc_with_a_tree_that_spills(rt, orc, monkeypatch):
    # the large-tree kernel shape (top levels in LDS, deep stack entries in the scratch array)
    monkeypatch.setenv("RTIOW_RENDER_LDS_NODES", "40")
    monkeypatch.setenv("RTIOW_RENDER_STACK_CAP", "3")
    w = rt.World(recipe("stress4096"))
    monkeypatch.delenv("RTIOW_RENDER_LDS_NODES")
    monkeypatch.delenv("RTIOW_RENDER_STACK_CAP")
    try:
        cam15 = np.array(golden("frames")["camera"])
        cam15[0:3] = [0.0, 25.0, 60.0]
        cam15[9] = 40.0
        cam = rt.camera(cam15)
        a = _frame(rt, w, cam, 640, 320, 0, 36, 4, rt.RT_ACCUM_F64, True)
        assert np.array_equal(a, _frame(rt, w, cam, 640, 320, 0, 36, 4, rt.RT_ACCUM_F64, False))
        b = _frame(rt, w, cam, 100, 50, 0, 3, 4, rt.RT_ACCUM_F64, True)
        assert np.array_equal(b, _frame(rt, w, cam, 100, 50, 0, 3, 4, rt.RT_ACCUM_F64, False))
        exp, _ = orc.Scene(recipe("stress4096")).render(cam15, 100, 50, 0, 3, 4, orc.ORC_RNG_STREAM)
        np.testing.assert_allclose(b, exp, rtol=1e-11, atol=1e-13)
    finally:
        w.close()


def test_noalloc_reports_its_limits(rt):
    import torch
    w = rt.World(recipe("book1"))
    try:
        cam = rt.camera(golden("frames")["camera"])
        # 2^19 - 8 ordered tiles of 4x4 pixels at most: 4096 x 2064 is 528 384 tiles
        params = w.params(4096, 2064, 0, 32, seed=1, accum=rt.RT_ACCUM_F32)
        # (a buffer of the frame's full size: if the limit ever moves, the call renders instead of refusing)
        accum = torch.zeros(2064, 4096, 3, dtype=torch.float32, device="cuda")
        if "RTIOW_TILE_SHIFT" in os.environ:
            pytest.skip("the limit is stated for the default 4x4 tiles")
        with pytest.raises(rt.RtError, match="scratch"):
            w.render_device_noalloc(cam, params, accum.data_ptr(), None)
        del accum
        # ... while 3840 x 2160 fits
        params = w.params(3840, 2160, 0, 32, seed=1, accum=rt.RT_ACCUM_F32)
        big = torch.zeros(2160, 3840, 3, dtype=torch.float32, device="cuda")
        w.render_device_noalloc(cam, params, big.data_ptr(), None)
        torch.cuda.synchronize()
        assert float(big.min()) >= 0 and 0.2 < float(big.mean()) / 32 < 1.0
        with pytest.raises(rt.RtError, match="null"):
            w.render_device_noalloc(cam, params, 0, None)
    finally:
        w.close()
