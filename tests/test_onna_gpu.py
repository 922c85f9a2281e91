"""GPU parity tests of libonna_hip.so (include/onna_hip.h) against the checker oracle/onna_oracle.c, through the C ABI.

Bar: a hit's object, t, point, normal and facing are bit-exact (they involve +, -, *, /, sqrt only); its texture
coordinates, Perlin values and anything downstream of sin / cos / acos / atan2 / log within 1e-13 (the GPU's and
glibc's libm differ in the last place); frames within 1e-11 relative (the recursion's products are taken front to
back on the GPU), with the same number of ray segments.  The checker itself is PARITY UNPINNED against the Lisp
reference (see its header); tests/test_onna_cpu.py holds it to closed-form facts.
"""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
on = importlib.import_module("ray-tracing-in-one-weekend_amd.onna")
import onna_oracle_lib as oo  # noqa: E402

pytestmark = pytest.mark.gpu
INF = float("inf")
SCENES = ["four-spheres", "bouncing-spheres", "chequered-spheres", "earth", "perlin-spheres", "quads", "simple-light", "cornell-box", "cornell-smoke", "final-scene"]


@pytest.fixture(scope="module")
def earth_png(tmp_path_factory):
    yy, xx = np.mgrid[0:64, 0:128]
    img = np.stack([(xx * 2) % 256, (yy * 4) % 256, (xx + yy) % 256], -1).astype(np.uint8)
    path = tmp_path_factory.mktemp("onna") / "earth.png"
    on.write_png(path, img)
    return path


def same_hits(got, exp, media=False):
    assert np.array_equal(got["hit"], exp["hit"])
    m = exp["hit"] == 1
    assert np.array_equal(got["object"][m], exp["object"][m]) and np.array_equal(got["material"][m], exp["material"][m])
    assert np.array_equal(got["front_face"][m], exp["front_face"][m])
    if media:   # a medium's hit time comes out of log()
        assert np.allclose(got["t"][m], exp["t"][m], rtol=1e-13, atol=0) and np.allclose(got["p"][m], exp["p"][m], rtol=1e-12, atol=1e-9)
    else:
        assert np.array_equal(got["t"][m], exp["t"][m]) and np.array_equal(got["p"][m], exp["p"][m])
    assert np.array_equal(got["normal"][m], exp["normal"][m])
    assert np.allclose(got["u"][m], exp["u"][m], rtol=0, atol=1e-13) and np.allclose(got["v"][m], exp["v"][m], rtol=0, atol=1e-13)


def probe_rays(tables, cam, n, seed):
    """camera rays, rays from random points towards the scene's box, and secondary rays leaving hit points"""
    rng = np.random.default_rng(seed)
    w, h = cam.image_width, cam.image_height
    px = rng.integers(0, w, n)
    py = rng.integers(0, h, n)
    first = np.array([oo.camera_ray(cam, int(px[k]), int(py[k]), 0, 0, 9, int(py[k] * w + px[k]), k) for k in range(n)])
    box = tables.objects()[tables.desc.world]["box"]
    lo, hi = box[0::2], box[1::2]
    lo, hi = np.maximum(lo, -2000), np.minimum(hi, 2000)
    o = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), (n, 3))
    d = rng.uniform(lo, hi, (n, 3)) - o
    loose = np.hstack([o, d, rng.random((n, 1))])
    u = rng.random((n, 4))
    hits = oo.hit(tables, tables.desc.world, first, 0.001, INF, u)
    m = hits["hit"] == 1
    dirs = rng.normal(size=(m.sum(), 3))
    second = np.hstack([hits["p"][m], dirs, first[m][:, 6:7]])
    return np.vstack([first, loose, second])


@pytest.mark.parametrize("name", SCENES)
def test_hit_test_matches_the_checker(name, earth_png):
    t = on.Tables(name, seed=5, image_path=earth_png)
    s = on.Scene(t)
    cam = t.camera(width=160, spp=1, depth=4)
    rays = probe_rays(t, cam, 1500, 3)
    u = np.random.default_rng(4).random((len(rays), 6))
    media = bool((t.objects()["kind"] == on.ON_OBJ_MEDIUM).any())
    got = s.hit(t.desc.world, rays, 0.001, INF, u)
    exp = oo.hit(t, t.desc.world, rays, 0.001, INF, u)
    assert exp["hit"].sum() > len(rays) // 4
    same_hits(got, exp, media)
    # a narrow interval, as a container hands down once something nearer was found
    mid = float(np.median(exp["t"][exp["hit"] == 1]))
    same_hits(s.hit(t.desc.world, rays, 0.001, mid, u), oo.hit(t, t.desc.world, rays, 0.001, mid, u), media)
    s.close()


def test_two_spheres_has_no_materials_and_is_hit_tested_only(earth_png):
    t = on.Tables("two-spheres")
    s = on.Scene(t)
    rays = probe_rays(t, t.camera(width=64, spp=1), 500, 1)
    same_hits(s.hit(t.desc.world, rays, 0.001, INF), oo.hit(t, t.desc.world, rays, 0.001, INF))
    with pytest.raises(on.OnnaError, match="no material"):
        s.render(t.camera(width=8, spp=1), 1)
    s.close()


def test_textures_match_the_checker(earth_png):
    rng = np.random.default_rng(6)
    t = on.Tables("final-scene", seed=5, image_path=earth_png)
    s = on.Scene(t)
    tex = t.textures()
    uvp = np.hstack([rng.random((20000, 2)), rng.uniform(-400, 400, (20000, 3))])
    uvp[:4, :2] = [[0, 0], [1, 1], [1, 0], [0, 1]]
    for i, k in enumerate(tex["kind"]):
        got, exp = s.texture(i, uvp), oo.texture(t, i, uvp)
        if k == on.ON_TEX_PERLIN:
            assert np.allclose(got, exp, rtol=0, atol=1e-12)   # sin() of a sum of seven octaves times ten
        else:
            assert np.array_equal(got, exp)
    s.close()
    b = on.Builder()
    cheq = b.chequer(0.32, b.solid(0.2, 0.3, 0.1), b.chequer(0.05, b.solid(1, 0, 0), b.solid(0, 0, 1)))
    t = b.finish(b.sphere([0, 0, 0], 1, b.lambertian(cheq)))
    s = on.Scene(t)
    i = int(np.nonzero(t.textures()["kind"] == on.ON_TEX_CHEQUER)[0][-1])
    uvp[:, 2:] = rng.uniform(-3, 3, (20000, 3))
    assert np.array_equal(s.texture(i, uvp), oo.texture(t, i, uvp))
    s.close()


def test_light_densities_match_the_checker():
    rng = np.random.default_rng(7)
    t = on.Tables("cornell-box")
    s = on.Scene(t)
    n = 50000
    origin = rng.uniform(1, 554, (n, 3))
    tape = rng.random((n, 3))
    towards = oo.random_direction(t, t.desc.lights, origin, tape)   # directions the sampler would choose ...
    stray = rng.normal(size=(n, 3))                                   # ... and directions it would not
    for d in (towards, stray):
        od = np.hstack([origin, d])
        got, exp = s.pdf_value(t.desc.lights, od), oo.pdf_value(t, t.desc.lights, od)
        assert np.allclose(got, exp, rtol=1e-14, atol=0, equal_nan=True)   # (an origin inside the sphere: the square root of a negative number, on both sides)
    assert (oo.pdf_value(t, t.desc.lights, np.hstack([origin, towards])) > 0).mean() > 0.97   # (origins inside the glass sphere give NaN)
    with pytest.raises(on.OnnaError):
        s.pdf_value(t.desc.world, np.hstack([origin, stray]))   # the world is not a list of spheres and quads
    s.close()


@pytest.mark.parametrize("name", SCENES)
def test_frames_match_the_checker(name, earth_png):
    t = on.Tables(name, seed=5, image_path=earth_png)
    s = on.Scene(t)
    cam = t.camera(width=40, spp=9, depth=12)
    got, stats = s.render(cam, seed=11)
    exp, c = oo.render(t, cam, seed=11)
    assert stats.paths == c.paths == cam.image_width * cam.image_height * 9
    media = bool((t.objects()["kind"] == on.ON_OBJ_MEDIUM).any())
    if not media:
        assert stats.rays == c.rays
    else:   # a free path within an ulp of the boundary can end a path on one side only
        assert abs(int(stats.rays) - int(c.rays)) <= 2
    close = np.isclose(got, exp, rtol=1e-11, atol=1e-13)
    assert close.mean() > (0.999 if media else 0.9999), (name, (~close).sum())
    assert np.isfinite(got).all()
    s.close()


def test_a_larger_frame_matches_the_checker():
    # 225 tiles of 8x8 pixels, the last row and column of them ragged, several runs of samples per tile
    t = on.Tables("cornell-smoke")
    s = on.Scene(t)
    cam = t.camera(width=117, spp=16, depth=20)
    got, stats = s.render(cam, seed=21)
    exp, c = oo.render(t, cam, seed=21)
    assert stats.paths == c.paths == 117 * 117 * 16
    assert abs(int(stats.rays) - int(c.rays)) <= 4   # (media: a free path within an ulp of the boundary)
    close = np.isclose(got, exp, rtol=1e-11, atol=1e-13)
    assert close.mean() > 0.999, (~close).sum()
    s.close()


def test_sample_ranges_add_up_and_frames_repeat():
    t = on.Tables("cornell-box")
    s = on.Scene(t)
    cam = t.camera(width=48, spp=16, depth=10)
    whole, _ = s.render(cam, seed=3)
    again, _ = s.render(cam, seed=3)
    assert np.array_equal(whole, again)
    parts = s.render(cam, seed=3, s_begin=0, s_end=5)[0] + s.render(cam, seed=3, s_begin=5, s_end=16)[0]
    assert np.allclose(whole, parts, rtol=1e-13, atol=1e-15)
    other, _ = s.render(cam, seed=4)
    assert not np.array_equal(whole, other)
    with pytest.raises(on.OnnaError):
        s.render(cam, seed=3, s_begin=0, s_end=17)   # beyond the camera's strata
    s.close()


def test_frames_do_not_depend_on_how_the_samples_are_dealt_out(monkeypatch):
    # A wave takes a run of samples of its 8x8 pixels and its lanes share the run's paths; every path's colour has its own
    # slot and a pixel's slots are added in sample order.  However the runs are cut, the frame and the count are the same.
    t = on.Tables("cornell-smoke")
    s = on.Scene(t)
    cam = t.camera(width=44, spp=16, depth=10)   # 44 x 44: tiles that reach beyond the image's edge
    plain, st_plain = s.render(cam, seed=9)
    for runs in ("1", "16", "7,3", "5,4,3,2,1"):
        monkeypatch.setenv("ONNA_RUNS", runs)
        got, st = s.render(cam, seed=9)
        assert np.array_equal(got, plain), runs
        assert st.rays == st_plain.rays and st.paths == st_plain.paths == 44 * 44 * 16
    monkeypatch.delenv("ONNA_RUNS")
    # more samples than the slot buffer holds go through in passes, each pass's sum added to the image in turn
    monkeypatch.setenv("ONNA_WINDOW", "5")
    passes, st = s.render(cam, seed=9)
    monkeypatch.delenv("ONNA_WINDOW")
    by_hand = np.zeros_like(plain)
    for a in range(0, 16, 5):
        by_hand = by_hand + s.render(cam, seed=9, s_begin=a, s_end=min(16, a + 5))[0]
    assert np.array_equal(passes, by_hand) and st.rays == st_plain.rays
    assert np.allclose(passes, plain, rtol=1e-13, atol=1e-15) and plain.any()
    # a small frame with many samples: dozens of short runs by the default rule against one run of them all
    cam = t.camera(width=16, spp=256, depth=6)
    many, st_many = s.render(cam, seed=2)
    monkeypatch.setenv("ONNA_RUNS", "256")
    one, st_one = s.render(cam, seed=2)
    monkeypatch.delenv("ONNA_RUNS")
    assert np.array_equal(many, one) and st_many.rays == st_one.rays and many.any()
    s.close()


def test_sampling_the_lights_changes_the_noise_not_the_picture():
    t = on.Tables("cornell-box")
    cam = t.camera(width=32, spp=1024, depth=20)
    s = on.Scene(t)
    a1, a2 = s.render(cam, seed=1)[0] / 1024, s.render(cam, seed=2)[0] / 1024
    s.close()
    t.desc.lights = -1   # the material's own density alone
    s = on.Scene(t)
    b1, b2 = s.render(cam, seed=1)[0] / 1024, s.render(cam, seed=2)[0] / 1024
    s.close()
    # block means over 8x8 pixels: the same expectation, within what 65 536 samples a block allow
    blocks = lambda x: x.reshape(4, 8, 4, 8, 3).mean(axis=(1, 3))
    assert np.allclose(blocks(a1), blocks(b1), rtol=0.12, atol=0.01)
    assert abs(a1.mean() - b1.mean()) < 0.03 * b1.mean()
    # ... and less noise with the lights sampled (two seeds apart): that is what the list of lights is for
    assert np.abs(a1 - a2).mean() < 0.6 * np.abs(b1 - b2).mean()


def test_resolve_matches_the_checker_and_the_host_api():
    import torch
    t = on.Tables("simple-light", seed=2)
    s = on.Scene(t)
    cam = t.camera(width=64, spp=16, depth=10)
    sums, _ = s.render(cam, seed=5)
    dev = torch.from_numpy(sums).cuda()
    out = torch.zeros(sums.shape, dtype=torch.uint8, device="cuda")
    on.resolve(dev.data_ptr(), sums.size // 3, 16, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oo.resolve(sums, 16))
    # on_render into device memory adds to what is there
    acc = torch.ones(sums.shape, dtype=torch.float64, device="cuda")
    s.render_device(cam, on.on_render_params(0, 16, 5), acc.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), sums + 1.0)
    s.close()


def test_shapes_the_walk_cannot_take_are_refused():
    b = on.Builder()
    inner = b.sphere([0, 0, 0], 1.0, b.lambertian(b.solid(0.5, 0.5, 0.5)))
    fog = b.medium(inner, 0.1, b.solid(1, 1, 1))
    with pytest.raises(on.OnnaError, match="inside the boundary"):
        on.Scene(b.finish(b.medium(fog, 0.1, b.solid(1, 1, 1))))
    b = on.Builder()
    o = b.sphere([0, 0, 0], 1.0, b.lambertian(b.solid(0.5, 0.5, 0.5)))
    for k in range(5):
        o = b.translate(o, [1, 0, 0])
    t = b.finish(o)
    s = on.Scene(t)
    with pytest.raises(on.OnnaError, match="nested deeper"):
        s.hit(t.desc.world, np.zeros((1, 7)), 0.001, INF)
    with pytest.raises(on.OnnaError, match="nested deeper"):
        s.render(on.make_camera(1.0, 8, 1, 5, (0, 0, 0), 40, (0, 0, 5), (0, 0, 0)), 1)
    s.close()
    # four instances are fine, and equal to the checker
    b = on.Builder()
    o = b.box([-1, -1, -1], [1, 1, 1], b.lambertian(b.solid(0.5, 0.5, 0.5)))
    o = b.translate(b.rotate_y(b.translate(b.rotate_y(o, 30), [0.5, 0, 0]), -50), [0, 0.25, 0])
    t = b.finish(b.list([o, b.medium(b.sphere([0, 0, 0], 3.0), 0.2, b.solid(1, 1, 1))]))
    s = on.Scene(t)
    rng = np.random.default_rng(2)
    origin = rng.normal(size=(3000, 3)) * 4
    rays = np.hstack([origin, rng.normal(size=(3000, 3)) * 0.3 - origin, np.zeros((3000, 1))])
    u = rng.random((3000, 2))
    same_hits(s.hit(t.desc.world, rays, 0.001, INF, u), oo.hit(t, t.desc.world, rays, 0.001, INF, u), media=True)
    with pytest.raises(on.OnnaError, match="uniforms"):
        s.hit(t.desc.world, rays, 0.001, INF)
    s.close()


def test_host_api_through_cxx():
    exe = os.path.join(ROOT, "ray-tracing-in-one-weekend_amd", "csrc", "onna_api_check")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "onna api check ok" in r.stdout
