"""GPU parity: every entry point of include/rtiow_hip.h against the golden vectors
captured from the compiled reference and against the oracle on the same inputs.

Bars (BASELINE.json north_star): hit/miss and primitive id bit-exact; t, p, normal within
1e-5 relative -- these tests hold them to bit-exact, which the double-precision,
unfused arithmetic achieves.  Frames: the GPU renders the very paths the oracle renders
when both use the per-(pixel, sample) streams, so sums agree to rounding (1e-11 relative;
the product of attenuations is taken in the opposite order); against the reference's own
mt19937 frames the comparison is statistical, with the Monte-Carlo noise floor measured
from two reference seeds.
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import HIT_SCENES, TRACE_SCENES, golden, recipe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def worlds(rt):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = rt.World(recipe(name))
        return cache[name]
    yield get
    for w in cache.values():
        w.close()


def test_device_present(rt):
    assert rt.hip().rt_device_count() >= 1


def test_aabb_hit(rt, orc):
    g = golden("aabb")
    got = rt.aabb_hit_batch(g["cases"])
    assert np.array_equal(got, g["hit"])
    assert np.array_equal(got, orc.aabb_batch(g["cases"]))


@pytest.mark.parametrize("scene", TRACE_SCENES)
def test_uploaded_tree_is_the_reference_tree(worlds, scene):
    nodes, _ = worlds(scene).tables()
    assert np.array_equal(nodes, golden("trace_" + scene)["topology"])


@pytest.mark.parametrize("scene", HIT_SCENES)
def test_object_hit(worlds, scene):
    g = golden("trace_" + scene)
    w = worlds(scene)
    for rays, tmin, tmax, exp in [(g["rays"], float(g["tmin"]), float(g["tmax"]), g["expected"]),
                                  (g["rays2"], float(g["tmin2"]), float(g["tmax2"]), g["expected2"])]:
        h = w.trace(rays, tmin, tmax)
        assert np.array_equal(h["hit"], exp[:, 0].astype(np.int32))          # hit / miss
        assert np.array_equal(h["prim"], exp[:, 1].astype(np.int32))         # object id
        m = h["hit"] == 1
        assert np.array_equal(h["t"][m], exp[m, 2])
        assert np.array_equal(h["p"][m], exp[m, 3:6])
        assert np.array_equal(h["normal"][m], exp[m, 6:9])


def test_object_hit_random_rays_against_oracle(worlds, orc):
    # a larger batch than the fixtures hold, incoherent, with times over the whole [-1, 1) range
    rng = np.random.default_rng(7)
    for scene, centre, spread in [("book2", [0, 0.5, 0], 8.0), ("stress4096", [0, 0.5, 0], 30.0), ("mixed", [0, 0.3, -1], 3.0)]:
        n = 60000
        o = np.array(centre) + spread * rng.uniform(-1, 1, (n, 3))
        tgt = np.array(centre) + 0.6 * spread * rng.uniform(-1, 1, (n, 3))
        rays = np.hstack([o, (tgt - o) * rng.uniform(0.2, 3, (n, 1)), rng.uniform(-1, 1, (n, 1))])
        h = worlds(scene).trace(rays, 0.001, np.finfo(np.float64).max)
        hit, prim, tpn = orc.Scene(recipe(scene)).trace(rays, 0.001, np.finfo(np.float64).max)
        assert np.array_equal(h["hit"], hit.astype(np.int32))
        assert np.array_equal(h["prim"], prim)
        m = hit == 1
        assert np.array_equal(h["t"][m], tpn[m, 0])
        assert np.array_equal(h["p"][m], tpn[m, 1:4])
        assert np.array_equal(h["normal"][m], tpn[m, 4:7])
        assert 0.2 < m.mean() < 0.99


def _random_recipe(rng, path, moving=True):
    """A random world: bare spheres and bvh groups of awkward sizes, static and (unless moving=False)
    moving spheres with assorted time ranges, coincident centres (ties), flat and huge radii, at a
    random scale."""
    scale = float(rng.choice([0.01, 1.0, 1.0, 30.0, 1e4]))
    lines = ["rtiow-recipe 1", "texture constant 0.5 0.5 0.5", "material lambertian 0", "material metal 0.8 0.8 0.8 0.1", "material dielectric 1.5"]

    def sphere():
        c = rng.uniform(-1, 1, 3) * scale
        if rng.random() < 0.15:
            c = np.round(c / scale * 2) / 2 * scale                     # a coarse lattice: coincident centres
        r = scale * float(rng.choice([0.02, 0.05, 0.1, 0.2, 0.6])) * rng.uniform(0.5, 1.5)
        m = int(rng.integers(0, 3))
        if moving and rng.random() < 0.35:
            c1 = c + rng.uniform(-0.5, 0.5, 3) * scale
            t0, t1 = [(0, 1), (0, 1), (0.25, 2), (-1, 3)][int(rng.integers(0, 4))]
            return f"moving_sphere {m} " + " ".join(repr(float(v)) for v in [*c, *c1, t0, t1, r])
        return f"sphere {m} " + " ".join(repr(float(v)) for v in [*c, r])

    for _ in range(int(rng.integers(1, 5))):
        kind = rng.random()
        if kind < 0.3:
            lines.append(sphere())
        else:
            n = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 17, 40, 150]))
            # (the bvh's own times decide which positions the moving spheres' boxes cover: moving_sphere.cpp:51-60)
            t0, t1 = [(0, 1), (0, 1), (1, 2), (-3, -1), (0.25, 0.75), (2, 2)][int(rng.integers(0, 6))] if moving else (0, 1)
            lines.append(f"bvh_begin {int(rng.integers(1, 1000))} {t0} {t1}")
            k = 0
            while k < n:
                if rng.random() < 0.15:     # an object_list as one object of the bvh_node (1 to 5 members)
                    members = int(rng.integers(1, 6))
                    lines += ["list_begin"] + [sphere() for _ in range(members)] + ["list_end"]
                    k += members
                else:
                    lines.append(sphere())
                    k += 1
            lines.append("bvh_end")
    path.write_text("\n".join(lines) + "\n")
    return scale


def test_random_worlds_against_oracle(rt, orc, tmp_path):
    # object::hit on worlds nobody designed: every hit flag, primitive, t, p and normal must equal
    # the oracle's to the bit, on both tree layouts.  Every other world has no moving sphere and so
    # gets the tree over single spheres and the static-scene kernel.
    rng = np.random.default_rng(int(os.environ.get("RTIOW_FUZZ_SEED", "2024")))
    for k in range(int(os.environ.get("RTIOW_FUZZ_WORLDS", "40"))):
        path = tmp_path / f"w{k}.recipe"
        scale = _random_recipe(rng, path, moving=k % 2 == 0)
        n = 20000
        o = rng.uniform(-3, 3, (n, 3)) * scale
        tgt = rng.uniform(-1.2, 1.2, (n, 3)) * scale
        d = (tgt - o) * rng.uniform(0.05, 20, (n, 1))
        axis = rng.integers(0, 3, n)
        flat = rng.random(n) < 0.05
        d[flat, axis[flat]] = 0.0                                       # rays parallel to a slab
        rays = np.hstack([o, d, rng.uniform(-1, 1, (n, 1))])
        tmin, tmax = (0.001, np.finfo(np.float64).max) if k % 3 else (float(rng.uniform(-2, 0.5)), float(rng.uniform(1, 50)))
        hit, prim, tpn = orc.Scene(str(path)).trace(rays, tmin, tmax)
        for layout in ("", "reference"):
            if layout:
                os.environ["RTIOW_TREE"] = layout
            try:
                w = rt.World(str(path))
            finally:
                os.environ.pop("RTIOW_TREE", None)
            h = w.trace(rays, tmin, tmax)
            if not layout:      # and the render kernel's walk (LDS-resident tree) on the same world
                cam15 = np.array([3 * scale, 2 * scale, 3 * scale, 0, 0, 0, 0, 1, 0, 40, 2.0, 0.0, 4 * scale, 0, 1], dtype=np.float64)
                got, st = w.render(rt.camera(cam15), 40, 20, 0, 3, 100 + k, rt.RT_ACCUM_F64)
                exp, counters = orc.Scene(str(path)).render(cam15, 40, 20, 0, 3, 100 + k, orc.ORC_RNG_STREAM)
                assert st.rays == counters.rays, k
                np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-13)
            w.close()
            assert np.array_equal(h["hit"], hit.astype(np.int32)), (k, layout)
            assert np.array_equal(h["prim"], prim), (k, layout)
            m = hit == 1
            assert np.array_equal(h["t"][m], tpn[m, 0]), (k, layout)
            assert np.array_equal(h["p"][m], tpn[m, 1:4]) and np.array_equal(h["normal"][m], tpn[m, 4:7]), (k, layout)


@pytest.mark.parametrize("times", [(1, 2), (-3, -1), (0, 1), (0.5, 0.5)])
def test_moving_spheres_under_other_bvh_times(rt, orc, tmp_path, times):
    # A bvh_node built for times other than its spheres' own (bvh_node.hpp:17 takes any): the boxes then
    # surround where the spheres are at THOSE times (moving_sphere.cpp:51-60), far from where a ray of
    # time 0 finds them.  The reference tries a sphere whenever the ray crosses its box and takes the
    # nearest hit, wherever the sphere is by then -- also in front of a static sphere that the ray meets
    # before it reaches the box.  Rows of (moving sphere, static sphere in its way), rays along the rows.
    lines = ["rtiow-recipe 1", "texture constant 0.5 0.5 0.5", "material lambertian 0", f"bvh_begin 7 {times[0]} {times[1]}"]
    rows = 24
    for k in range(rows):
        y, z = 3.0 * (k % 6), 3.0 * (k // 6)
        lines.append(f"moving_sphere 0 0 {y} {z} 4 {y} {z} 0 1 1")          # at x = 4 t
        lines.append(f"sphere 0 2 {y} {z} 0.5")                             # between x = 1.5 and 2.5
        lines.append(f"sphere 0 -6 {y} {z} 0.5")
    lines.append("bvh_end")
    path = tmp_path / "times.recipe"
    path.write_text("\n".join(lines) + "\n")
    rng = np.random.default_rng(11)
    n = 40000
    k = rng.integers(0, rows, n)
    o = np.stack([rng.uniform(-12, 12, n), 3.0 * (k % 6) + rng.uniform(-1.2, 1.2, n), 3.0 * (k // 6) + rng.uniform(-1.2, 1.2, n)], axis=1)
    d = np.stack([rng.choice([-1.0, 1.0], n) * rng.uniform(0.5, 2, n), rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n)], axis=1)
    rays = np.hstack([o, d, rng.uniform(-1, 1, (n, 1))])
    hit, prim, tpn = orc.Scene(str(path)).trace(rays, 0.001, np.finfo(np.float64).max)
    moving_hits = (hit == 1) & (prim % 3 == 0)
    assert moving_hits.mean() > 0.02
    w = rt.World(str(path))
    try:
        h = w.trace(rays, 0.001, np.finfo(np.float64).max)
        cam15 = np.array([-14, 7.5, 4.5, 0, 7.5, 4.5, 0, 1, 0, 60, 2.0, 0.0, 10, 0, 1], dtype=np.float64)
        got, st = w.render(rt.camera(cam15), 80, 40, 0, 4, 5, rt.RT_ACCUM_F64)
    finally:
        w.close()
    assert np.array_equal(h["hit"], hit.astype(np.int32)) and np.array_equal(h["prim"], prim)
    m = hit == 1
    assert np.array_equal(h["t"][m], tpn[m, 0]) and np.array_equal(h["normal"][m], tpn[m, 4:7])
    exp, counters = orc.Scene(str(path)).render(cam15, 80, 40, 0, 4, 5, orc.ORC_RNG_STREAM)
    assert st.rays == counters.rays
    np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-13)


def test_a_list_as_the_only_object_of_a_half(rt, orc, tmp_path, monkeypatch):
    # bvh_node.cpp:31-34: a span of one object makes that object both children.  When the object is an
    # object_list its subtree hangs under the node twice in the description; the device tree must hold
    # it once (same hits, no doubled tables), under both layouts.  Group 1: a list alone in its bvh_node.
    # Group 2: three objects whose first half (one object, whatever the axis) is a list.
    lines = ["rtiow-recipe 1", "texture constant 0.5 0.5 0.5", "material lambertian 0", "material metal 0.8 0.8 0.8 0.1",
             "bvh_begin 3 0 1", "list_begin", "sphere 0 0 0 0 0.5", "sphere 1 0.6 0 0 0.5", "moving_sphere 0 1.2 0 0 1.2 0.4 0 0 1 0.5", "list_end", "bvh_end",
             "bvh_begin 4 0 1", "list_begin", "sphere 0 -9 -9 -9 0.5", "sphere 1 -9.5 -9.2 -9.1 0.7", "sphere 0 -8.8 -9 -9.3 0.4", "sphere 0 -9 -9 -9 0.5", "list_end",
             "sphere 1 3 3 3 0.5", "sphere 0 4 4 4 0.5", "bvh_end"]
    path = tmp_path / "halves.recipe"
    path.write_text("\n".join(lines) + "\n")
    rng = np.random.default_rng(5)
    n = 30000
    tgt = np.where(rng.random((n, 1)) < 0.5, np.array([[0.6, 0, 0]]), np.array([[-9.1, -9.1, -9.1]])) + rng.uniform(-1, 1, (n, 3))
    o = tgt + rng.uniform(-6, 6, (n, 3))
    rays = np.hstack([o, tgt - o, rng.uniform(-1, 1, (n, 1))])
    hit, prim, tpn = orc.Scene(str(path)).trace(rays, 0.001, np.finfo(np.float64).max)
    assert 0.3 < (hit == 1).mean() < 0.99
    tables = {}
    for layout in ("", "reference"):
        if layout:
            monkeypatch.setenv("RTIOW_TREE", layout)
        w = rt.World(str(path))
        try:
            h = w.trace(rays, 0.001, np.finfo(np.float64).max)
        finally:
            w.close()
        assert np.array_equal(h["hit"], hit.astype(np.int32)) and np.array_equal(h["prim"], prim), layout
        m = hit == 1
        assert np.array_equal(h["t"][m], tpn[m, 0]) and np.array_equal(h["normal"][m], tpn[m, 4:7]), layout


def test_degenerate_rays(worlds, orc):
    # zero and negative-zero direction components, a zero direction, huge and tiny scales
    rays = np.array([
        [0, 5, 0, 0, -1, 0, 0.5], [0, 5, 0, -0.0, -1, 0.0, 0.5], [13, 2, 3, 0, 0, 0, 0.0],
        [0, 1, 0, 1e-300, 1e-300, 1e-300, 0.0], [0, 1, 0, 1e150, 1e150, -1e150, 0.0],
        [4, 1, 5, 0, 0, -1, -0.99], [0, -1000, -1, 0, 1, 0, 0.0], [0, 3000, 0, 0, -1, 0, 0.0],
    ], dtype=np.float64)
    for scene in ["book2", "mixed"]:
        h = worlds(scene).trace(rays, 0.001, np.finfo(np.float64).max)
        hit, prim, tpn = orc.Scene(recipe(scene)).trace(rays, 0.001, np.finfo(np.float64).max)
        assert np.array_equal(h["hit"], hit.astype(np.int32))
        assert np.array_equal(h["prim"], prim)
        m = hit == 1
        assert np.array_equal(h["t"][m], tpn[m, 0])


def test_empty_batches_and_bad_arguments(rt, worlds):
    w = worlds("mixed")
    assert len(w.trace(np.zeros((0, 7)), 0.001, 1e9)) == 0
    assert len(w.texture(0, np.zeros((0, 3)))) == 0
    with pytest.raises(rt.RtError, match="out of range"):
        w.texture(999, np.zeros((1, 3)))
    cam = rt.camera(golden("frames")["camera"])
    with pytest.raises(rt.RtError, match="empty image"):
        w.render(cam, 0, 4, 0, 1, 1)
    with pytest.raises(rt.RtError, match="sample_end"):
        w.render(cam, 4, 4, 2, 1, 1)
    out, stats = w.render(cam, 8, 8, 5, 5, 1)   # an empty sample range renders nothing
    assert not out.any() and stats.paths == 0


def test_textures(worlds):
    g = golden("texture")
    perl = worlds("two_perlin").texture(0, g["points"])
    assert np.array_equal(perl, g["perlin"])
    assert perl[4000, 0] == 0.50292955069550871
    cheq = worlds("book2").texture(2, g["points"])
    # sin() differs from glibc's in the last bit at most; only its sign is used
    assert np.array_equal(cheq, g["chequer"])


def test_chequer_near_the_zeros_of_sine(orc, worlds):
    # The chequer uses the sign of sin(10x) sin(10y) sin(10z) (texture.cpp:10-13).  Points whose
    # coordinates put 10x within a few ulps of a multiple of pi, at zero, far out, or not finite
    # are where a shortcut for the sign would go wrong first.
    rng = np.random.default_rng(5)
    k = np.concatenate([np.arange(-40, 41), rng.integers(-600000, 600000, 400)]).astype(np.float64)
    near = k * np.pi / 10
    xs = [near]
    for _ in range(3):
        xs.append(np.nextafter(xs[-1], np.inf))
    lo = near
    for _ in range(3):
        lo = np.nextafter(lo, -np.inf)
        xs.append(lo)
    special = np.array([0.0, -0.0, 1e-310, -1e-310, 2e5, 209715.2, 209715.3, 1e7, -3e9, 1e300, np.inf, -np.inf, np.nan])
    x = np.concatenate(xs + [special, rng.uniform(-50, 50, 2000)])
    pts = np.stack([x, rng.permutation(x), rng.uniform(-20, 20, len(x))], axis=1)
    pts = np.concatenate([pts, pts[:, [2, 0, 1]], pts[:, [1, 2, 0]]])
    got = worlds("book2").texture(2, pts)
    exp = orc.Scene(recipe("book2")).texture(2, pts)
    assert np.array_equal(got, exp)
    assert len(np.unique(got[:, 0])) == 2          # both colours occur


@pytest.mark.parametrize("scene", ["book2", "mixed"])
def test_scatter(worlds, scene):
    g = golden("scatter_" + scene)
    items, exp, k = g["items"], g["expected"], int(g["k"])
    sc, att, out, used = worlds(scene).scatter(items[:, 0].astype(np.int32), items[:, 1:8], items[:, 8:15], exp[:, 12:12 + k])
    assert np.array_equal(used, exp[:, 11].astype(np.int32))     # same number of draws
    assert np.array_equal(sc, exp[:, 0].astype(np.uint8))        # scattered or absorbed
    m = sc == 1
    assert np.array_equal(att[m], exp[m, 1:4])
    # lambertian and metal directions use + - * / sqrt only: bit-exact.  A dielectric's
    # reflect/refract choice compares a uniform with schlick(), whose pow() may differ
    # from glibc's in the last bit; both candidate directions are themselves exact.
    assert np.array_equal(out[m], exp[m, 4:11])


def test_scatter_reports_short_tape(rt, worlds):
    g = golden("scatter_book2")
    items, exp = g["items"][:64], g["expected"][:64]
    sc, att, out, used = worlds("book2").scatter(items[:, 0].astype(np.int32), items[:, 1:8], items[:, 8:15], exp[:, 12:13])
    need = exp[:, 11]
    assert np.array_equal(used == -1, need > 1)
    assert (used == -1).any()


def test_camera_get_ray(rt):
    g = golden("camera")
    k = int(g["k"])
    for cam15, exp in zip(g["cameras"], g["expected"]):
        out, used = rt.camera_batch(rt.camera(cam15), g["st"], exp[:, 8:8 + k])
        assert np.array_equal(used, exp[:, 7].astype(np.int32))
        assert np.array_equal(out, exp[:, 0:7])


FRAME_CASES = [("two_perlin", 48, 24, 6), ("book1", 64, 32, 8), ("book2", 64, 32, 8), ("mixed", 40, 40, 8), ("ties", 33, 17, 4), ("lists", 48, 24, 8)]


@pytest.mark.parametrize("scene,nx,ny,ns", FRAME_CASES)
def test_frame_matches_oracle_on_the_same_streams(rt, orc, worlds, scene, nx, ny, ns):
    cam15 = np.array(golden("trace_lists")["camera"] if scene == "lists" else golden("frames")["camera"])
    cam15[10] = nx / ny
    got, stats = worlds(scene).render(rt.camera(cam15), nx, ny, 0, ns, 2025, rt.RT_ACCUM_F64)
    exp, counters = orc.Scene(recipe(scene)).render(cam15, nx, ny, 0, ns, 2025, orc.ORC_RNG_STREAM)
    assert stats.paths == nx * ny * ns == counters.paths
    assert stats.rays == counters.rays                      # every path took the same number of bounces
    np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-13)


def test_sample_ranges_add_up(rt, worlds):
    # what the multi-GPU split relies on: disjoint sample ranges of the same seed sum to the whole
    cam = rt.camera(golden("frames")["camera"])
    w = worlds("book2")
    whole, _ = w.render(cam, 80, 40, 0, 12, 9)
    a, _ = w.render(cam, 80, 40, 0, 5, 9)
    b, _ = w.render(cam, 80, 40, 5, 12, 9)
    np.testing.assert_allclose(a + b, whole, rtol=1e-13, atol=1e-13)
    again, _ = w.render(cam, 80, 40, 0, 12, 9)
    assert np.array_equal(whole, again)                      # deterministic
    other, _ = w.render(cam, 80, 40, 0, 12, 10)
    assert not np.array_equal(whole, other)                  # the seed matters


def test_f32_accumulation_is_the_rounded_f64_sum(rt, worlds):
    cam = rt.camera(golden("frames")["camera"])
    w = worlds("book1")
    f64, _ = w.render(cam, 64, 32, 0, 6, 3, rt.RT_ACCUM_F64)
    f32, _ = w.render(cam, 64, 32, 0, 6, 3, rt.RT_ACCUM_F32)
    assert np.array_equal(f32, f64.astype(np.float32))


@pytest.mark.parametrize("scene", ["two_perlin", "book1", "book2"])
def test_frame_statistics_against_reference_renders(rt, worlds, scene):
    # The reference's frames come from one shared mt19937, the GPU's from per-path streams:
    # the images agree as two independent Monte-Carlo estimates of the same integrand do.
    g = golden("frames")
    ref_a, ref_b = g[scene + "_spp256_seed1"].astype(np.float64), g[scene + "_spp256_seed2"].astype(np.float64)
    floor = np.sqrt(np.mean((ref_a - ref_b) ** 2))          # RMSE between two reference seeds
    sums, _ = worlds(scene).render(rt.camera(g["camera"]), 120, 60, 0, 256, 77)
    img = sums / 256
    for ref in (ref_a, ref_b):
        rmse = np.sqrt(np.mean((img - ref) ** 2))
        assert rmse <= 1.15 * floor, (rmse, floor)
    # bias check: per-channel means within 4 sigma of the floor's standard error
    se = floor / np.sqrt(img.shape[0] * img.shape[1]) * 4 + 1e-4
    ref_mean = 0.5 * (ref_a + ref_b)
    assert np.all(np.abs(img.mean(axis=(0, 1)) - ref_mean.mean(axis=(0, 1))) < se)
    # 8-bit output as main.cpp:26-28 quantises it
    q = lambda x: (255.99 * np.sqrt(x)).astype(int)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((q(img) - q(ref_a)) ** 2))
    psnr_floor = 10 * np.log10(255.0 ** 2 / np.mean((q(ref_b) - q(ref_a)) ** 2))
    assert psnr >= psnr_floor - 1.0


@pytest.mark.parametrize("scene", ["two_perlin", "book1", "book2"])
def test_deep_frames_against_reference_renders(rt, worlds, scene):
    # 400x200 at 1024 samples per pixel, two reference seeds per scene (tests/golden/frames_1024.npz):
    # the noise floor between the seeds is ~0.004-0.008 in linear radiance, so the bounds below see a
    # bias of a few 1e-4 in the image.  The GPU frame is rendered twice: at the reference's size, and
    # at 2000x1000 with 41 samples per pixel box-filtered 5x5 down to 400x200 -- 1025 samples spread
    # over each reference pixel's footprint, the same estimator through a full-size frame.
    g = golden("frames_1024")
    ref_a, ref_b = g[scene + "_spp1024_seed1"].astype(np.float64), g[scene + "_spp1024_seed2"].astype(np.float64)
    floor = np.sqrt(np.mean((ref_a - ref_b) ** 2))
    cam = rt.camera(g["camera"])
    direct, _ = worlds(scene).render(cam, 400, 200, 0, 1024, 78)
    wide, _ = worlds(scene).render(cam, 2000, 1000, 0, 41, 79)
    q = lambda x: (255.99 * np.sqrt(x)).astype(int)
    psnr_floor = 10 * np.log10(255.0 ** 2 / np.mean((q(ref_b) - q(ref_a)) ** 2))
    ref_mean = 0.5 * (ref_a + ref_b)
    for img in (direct / 1024, wide.reshape(200, 5, 400, 5, 3).sum(axis=(1, 3)) / (25 * 41)):
        for ref in (ref_a, ref_b):
            rmse = np.sqrt(np.mean((img - ref) ** 2))
            assert rmse <= 1.04 * floor, (rmse, floor)       # two independent estimates differ by the floor
        # bias: per-channel means within 4 standard errors of the difference (+ 2e-5)
        se = 4 * floor * np.sqrt(1.5 / (img.shape[0] * img.shape[1])) + 2e-5
        assert np.all(np.abs(img.mean(axis=(0, 1)) - ref_mean.mean(axis=(0, 1))) < se), (img.mean(axis=(0, 1)) - ref_mean.mean(axis=(0, 1)), se)
        # the 8-bit output (main.cpp:26-28): PSNR against a reference frame within 0.3 dB of the other reference frame's
        psnr = 10 * np.log10(255.0 ** 2 / np.mean((q(img) - q(ref_a)) ** 2))
        assert psnr >= psnr_floor - 0.3, (psnr, psnr_floor)
        # no structured error: the difference image, box-filtered 20x20, is as flat as the references' own
        blur = lambda x: x.reshape(10, 20, 20, 20, 3).mean(axis=(1, 3))
        assert np.abs(blur(img - ref_mean)).max() <= 1.5 * np.abs(blur(ref_a - ref_b)).max() + 1e-4


def test_full_size_frame_properties(rt, worlds):
    # BASELINE.json's size (1920x1080); the oracle cannot render it in seconds, so the
    # checks are size-independent properties: determinism, additivity over sample
    # ranges, bounded radiance (test_deep_frames_against_reference_renders holds a full-size
    # frame against the reference's images).
    g = golden("frames")
    cam15 = np.array(g["camera"])
    cam15[10] = 1920 / 1080
    cam = rt.camera(cam15)
    w = worlds("book1")
    a, sa = w.render(cam, 1920, 1080, 0, 2, 11, rt.RT_ACCUM_F64)
    b, sb = w.render(cam, 1920, 1080, 2, 4, 11, rt.RT_ACCUM_F64)
    ab, sab = w.render(cam, 1920, 1080, 0, 4, 11, rt.RT_ACCUM_F64)
    np.testing.assert_allclose(a + b, ab, rtol=1e-13, atol=1e-13)
    assert sa.rays + sb.rays == sab.rays and sab.paths == 1920 * 1080 * 4
    assert ab.min() >= 0 and ab.max() <= 4.0 + 1e-9          # every sample's radiance is within [0, 1]
    assert 2.0 < sab.rays / sab.paths < 4.0                  # bounces per path of this scene


def test_deep_frames_at_full_size(rt, worlds, monkeypatch):
    # 64 samples per pixel at 1920x1080: the path the benchmark takes (costing pass, tiles dearest
    # first, 4x4-pixel tiles).  Size-independent properties: run-to-run determinism, additivity
    # over sample ranges (the halves are rendered in 8x8 tiles, the whole in 4x4: same paths,
    # sums in another order), and the same frame from other tile sizes and from image order.
    cam15 = np.array(golden("frames")["camera"])
    cam15[10] = 1920 / 1080
    cam = rt.camera(cam15)
    w = worlds("book2")
    whole, s = w.render(cam, 1920, 1080, 0, 64, 3, rt.RT_ACCUM_F64)
    again, s2 = w.render(cam, 1920, 1080, 0, 64, 3, rt.RT_ACCUM_F64)
    assert np.array_equal(whole, again) and s.rays == s2.rays
    a, sa = w.render(cam, 1920, 1080, 0, 32, 3, rt.RT_ACCUM_F64)
    b, sb = w.render(cam, 1920, 1080, 32, 64, 3, rt.RT_ACCUM_F64)
    assert sa.rays + sb.rays == s.rays
    np.testing.assert_allclose(a + b, whole, rtol=1e-12, atol=1e-12)
    # a ragged size, every tile size, with and without ordering
    cam15[10] = 1001 / 701
    cam = rt.camera(cam15)
    base, sbase = w.render(cam, 1001, 701, 5, 69, 3, rt.RT_ACCUM_F64)
    for shift in ("1", "2", "3"):
        monkeypatch.setenv("RTIOW_TILE_SHIFT", shift)
        other, so = w.render(cam, 1001, 701, 5, 69, 3, rt.RT_ACCUM_F64)
        monkeypatch.setenv("RTIOW_TILE_ORDER_OFF", "1")
        plain, sp = w.render(cam, 1001, 701, 5, 69, 3, rt.RT_ACCUM_F64)
        monkeypatch.delenv("RTIOW_TILE_ORDER_OFF")
        monkeypatch.delenv("RTIOW_TILE_SHIFT")
        assert so.rays == sbase.rays == sp.rays
        assert np.array_equal(other, plain)
        np.testing.assert_allclose(other, base, rtol=1e-12, atol=1e-12)


def test_resolve(rt):
    import torch
    sums = torch.rand(4096 * 3, device="cuda") * 8
    out = torch.zeros(4096 * 3, dtype=torch.uint8, device="cuda")
    rc = rt.hip().rt_resolve(0, ctypes.c_void_p(sums.data_ptr()), 4096, 8, ctypes.c_void_p(out.data_ptr()), None)
    assert rc == 0
    torch.cuda.synchronize()
    exp = (255.99 * np.sqrt(sums.cpu().numpy().astype(np.float64) / 8)).astype(int)
    assert np.array_equal(out.cpu().numpy().astype(int), exp)


def test_million_sphere_tree(rt, orc, tmp_path):
    # BASELINE configs[4]: 10^6 spheres -> 1 048 575 reference nodes, 20 levels; the tree does not
    # fit LDS, so this exercises the device-memory walk.  Hits are compared with the oracle.
    path = str(tmp_path / "stress1m.recipe")
    rt.generate("stress", 1, path, n=1000000)
    w = rt.World(path)
    nodes, _ = w.tables()
    assert len(nodes) == 1048575
    rng = np.random.default_rng(3)
    n = 20000
    o = np.column_stack([rng.uniform(-500, 500, n), rng.uniform(0.5, 40, n), rng.uniform(-500, 500, n)])
    tgt = np.column_stack([rng.uniform(-500, 500, n), rng.uniform(-2, 1, n), rng.uniform(-500, 500, n)])
    rays = np.hstack([o, tgt - o, rng.uniform(-1, 1, (n, 1))])
    h = w.trace(rays, 0.001, np.finfo(np.float64).max)
    hit, prim, tpn = orc.Scene(path).trace(rays, 0.001, np.finfo(np.float64).max)
    assert np.array_equal(h["hit"], hit.astype(np.int32))
    assert np.array_equal(h["prim"], prim)
    m = hit == 1
    assert np.array_equal(h["t"][m], tpn[m, 0]) and np.array_equal(h["normal"][m], tpn[m, 4:7])
    assert m.mean() > 0.9
    # a small frame through the device-memory variant of the render kernel, against the oracle
    cam15 = [0.0, 25.0, 60.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 40.0, 2.0, 0.0, 10.0, 0.0, 1.0]
    got, stats = w.render(rt.camera(cam15), 48, 24, 0, 2, 5, rt.RT_ACCUM_F64)
    exp, counters = orc.Scene(path).render(cam15, 48, 24, 0, 2, 5, orc.ORC_RNG_STREAM)
    assert stats.rays == counters.rays
    np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-13)
    w.close()


@pytest.mark.parametrize("scene", ["book1", "book2", "stress4096", "mixed", "ties", "lists"])
def test_reclustered_tree_equals_reference_layout(rt, scene, monkeypatch):
    # The device tree is re-clustered over the reference's leaves (surface-area heuristic); with
    # RTIOW_TREE=reference the reference's own topology is uploaded instead.  Both must give the
    # same hits to the bit, and the same frames: which leaves a ray reaches depends only on the
    # leaves' boxes.
    g = golden("trace_" + scene)
    monkeypatch.setenv("RTIOW_TREE", "reference")
    w_ref = rt.World(recipe(scene))
    monkeypatch.delenv("RTIOW_TREE")
    w_new = rt.World(recipe(scene))
    a = w_ref.trace(g["rays"], float(g["tmin"]), float(g["tmax"]))
    b = w_new.trace(g["rays"], float(g["tmin"]), float(g["tmax"]))
    assert a.tobytes() == b.tobytes()
    cam = rt.camera(g_cam := golden("frames")["camera"])
    fa, sa = w_ref.render(cam, 64, 32, 0, 6, 4)
    fb, sb = w_new.render(cam, 64, 32, 0, 6, 4)
    assert sa.rays == sb.rays
    np.testing.assert_allclose(fa, fb, rtol=1e-13, atol=1e-13)
    w_ref.close()
    w_new.close()


@pytest.mark.parametrize("scene,lds_nodes,stack_cap", [("book1", 40, 3), ("book2", 0, 1), ("stress4096", 300, 4), ("mixed", 7, 2)])
def test_large_tree_kernel_shape_on_small_scenes(rt, orc, scene, lds_nodes, stack_cap, monkeypatch):
    # Trees too large for LDS keep only their top levels (breadth-first) there and spill deep stack
    # entries to device memory.  Forcing that shape on scenes the oracle renders in seconds: frames
    # must equal the oracle's and the all-in-LDS shape's, whatever the split.
    cam15 = np.array(golden("frames")["camera"])
    cam15[10] = 2.0
    w_lds = rt.World(recipe(scene))
    monkeypatch.setenv("RTIOW_RENDER_LDS_NODES", str(lds_nodes))
    monkeypatch.setenv("RTIOW_RENDER_STACK_CAP", str(stack_cap))
    w_big = rt.World(recipe(scene))
    monkeypatch.delenv("RTIOW_RENDER_LDS_NODES")
    monkeypatch.delenv("RTIOW_RENDER_STACK_CAP")
    monkeypatch.setenv("RTIOW_RENDER_WAVES", "3")      # the 768-thread shape of trees that only just fit
    w_768 = rt.World(recipe(scene))
    monkeypatch.delenv("RTIOW_RENDER_WAVES")
    try:
        fa, sa = w_lds.render(rt.camera(cam15), 72, 36, 0, 5, 21)
        fb, sb = w_big.render(rt.camera(cam15), 72, 36, 0, 5, 21)
        fc, sc = w_768.render(rt.camera(cam15), 72, 36, 0, 5, 21)
        assert sc.rays == sa.rays and np.array_equal(fa, fc)
        exp, counters = orc.Scene(recipe(scene)).render(cam15, 72, 36, 0, 5, 21, orc.ORC_RNG_STREAM)
        assert sa.rays == sb.rays == counters.rays
        # (at 5 samples the large-tree shape renders in 8x8 tiles, the other in 4x4: a pixel's samples
        # are summed in another order, which integer sums do not feel)
        assert np.array_equal(fa, fb)
        np.testing.assert_allclose(fb, exp, rtol=1e-11, atol=1e-13)
        g = golden("trace_" + scene)      # the batch walk reads the renumbered tree too
        a = w_lds.trace(g["rays"], float(g["tmin"]), float(g["tmax"]))
        b = w_big.trace(g["rays"], float(g["tmin"]), float(g["tmax"]))
        assert a.tobytes() == b.tobytes()
    finally:
        w_lds.close()
        w_big.close()
        w_768.close()


@pytest.mark.parametrize("scene", ["book1", "book2", "stress4096", "mixed", "ties"])
def test_postponed_leaf_box_tests_change_nothing(rt, orc, scene, monkeypatch):
    # Scenes without moving spheres get their own kernel variant and tree: leaves of two spheres are
    # split into single spheres with their own boxes, a sphere is tried before its leaf's box, and the
    # reference's test of that box (bvh_node.cpp:53 -> aabb.cpp:8-57) runs for the winning leaf only,
    # with a second walk that tests every box up front if it fails.  Off, on, on with every hit sent
    # through the second walk, and the same on the large-tree shape: one frame, to the bit, and the
    # oracle's.  (Scenes with moving spheres always run the general kernel: the switch does nothing.)
    cam15 = np.array(golden("frames")["camera"])
    cam15[10] = 2.0
    frames = []
    for defer, recheck, lds_nodes in [("0", False, None), ("1", False, None), ("1", True, None), ("1", False, "9"), ("1", True, "9")]:
        monkeypatch.setenv("RTIOW_STATIC_VARIANT", defer)
        if recheck:
            monkeypatch.setenv("RTIOW_RECHECK_ALL_LEAVES", "1")
        if lds_nodes:
            monkeypatch.setenv("RTIOW_RENDER_LDS_NODES", lds_nodes)
            monkeypatch.setenv("RTIOW_RENDER_STACK_CAP", "2")
        w = rt.World(recipe(scene))
        for name in ("RTIOW_STATIC_VARIANT", "RTIOW_RECHECK_ALL_LEAVES", "RTIOW_RENDER_LDS_NODES", "RTIOW_RENDER_STACK_CAP"):
            monkeypatch.delenv(name, raising=False)
        try:
            frames.append(w.render(rt.camera(cam15), 72, 36, 0, 5, 33))
        finally:
            w.close()
    exp, counters = orc.Scene(recipe(scene)).render(cam15, 72, 36, 0, 5, 33, orc.ORC_RNG_STREAM)
    for f, st in frames:
        assert st.rays == counters.rays
        assert np.array_equal(f, frames[0][0])   # (other tile sizes, other orders of summation: integer sums do not feel them)
    np.testing.assert_allclose(frames[0][0], exp, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("scene", ["book1", "stress4096", "ties"])
def test_tree_built_on_the_device_changes_nothing(rt, orc, scene, monkeypatch, capfd):
    # RTIOW_BUILD=device: the tree over a static scene's spheres is a linear BVH built by the GPU (Morton
    # codes of the centres, radix sort, one thread per inner node, boxes bottom-up) instead of the host's
    # surface-area clustering.  Another tree over the same leaves: object::hit and the frames must not move
    # a bit, whatever the shape -- coincident centres (equal codes) included -- and equal the oracle's.
    # (Large scenes take this path by default: test_million_sphere_tree.)
    rng = np.random.default_rng(17)
    n = 30000
    o = rng.uniform(-1, 1, (n, 3)) * [30, 6, 30] + [0, 3, 0]
    tgt = rng.uniform(-1, 1, (n, 3)) * [12, 0.5, 12]
    rays = np.hstack([o, (tgt - o) * rng.uniform(0.1, 3, (n, 1)), rng.uniform(-1, 1, (n, 1))])
    cam15 = np.array(golden("frames")["camera"])
    cam15[10] = 2.0
    monkeypatch.setenv("RTIOW_VERBOSE", "1")
    got = []
    for build in ("host", "device"):
        monkeypatch.setenv("RTIOW_BUILD", build)
        capfd.readouterr()
        w = rt.World(recipe(scene))
        said = capfd.readouterr().err
        assert ("built on the device" in said) == (build == "device"), said
        try:
            got.append((w.trace(rays, 0.001, np.finfo(np.float64).max), w.render(rt.camera(cam15), 72, 36, 0, 5, 33), w.render(rt.camera(cam15), 72, 36, 0, 40, 33)))
        finally:
            w.close()
    (ha, fa, da), (hb, fb, db) = got
    for field in ("hit", "prim", "t", "p", "normal"):
        assert np.array_equal(ha[field], hb[field]), field
    assert np.array_equal(fa[0], fb[0]) and fa[1].rays == fb[1].rays
    assert np.array_equal(da[0], db[0]) and da[1].rays == db[1].rays
    hit, prim, tpn = orc.Scene(recipe(scene)).trace(rays, 0.001, np.finfo(np.float64).max)
    assert np.array_equal(hb["hit"], hit.astype(np.int32)) and np.array_equal(hb["prim"], prim)
    exp, counters = orc.Scene(recipe(scene)).render(cam15, 72, 36, 0, 5, 33, orc.ORC_RNG_STREAM)
    assert fb[1].rays == counters.rays
    np.testing.assert_allclose(fb[0], exp, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("scene", ["book1", "stress4096", "ties", "lists_static"])
def test_four_wide_walk_changes_nothing(rt, orc, scene, monkeypatch, tmp_path, capfd):
    # RTIOW_WIDE=1: scenes without moving spheres walk a tree of four-child nodes (every other level of
    # the binary tree folded away).  Same candidates, same tie rule (bvh_node.cpp:57-61 -> the later
    # primitive among equals): the frame of the binary walk, to the bit, with the tree in LDS, with the
    # tree in device memory, with a stack that spills after four entries, with every hit sent through the
    # second, exact walk, and on the reference's own tree layout; and the oracle's frame.
    if scene == "lists_static":      # object_lists below bvh nodes, bare spheres beside them, no moving sphere
        rng = np.random.default_rng(3)
        path = tmp_path / "ls.recipe"
        while True:
            _random_recipe(rng, path, moving=False)
            if "list_begin" in path.read_text() and path.read_text().count("sphere") > 40:
                break
        scene_path = str(path)
        cam15 = np.array([3.0, 2.0, 3.0, 0, 0, 0, 0, 1, 0, 40, 2.0, 0.0, 4.0, 0, 1], dtype=np.float64)
        scale = float(np.abs(np.array([[float(v) for v in l.split()[2:5]] for l in path.read_text().splitlines() if l.startswith("sphere")])).max())
        cam15[[0, 1, 2, 12]] *= max(scale, 1e-3)
    else:
        scene_path = recipe(scene)
        cam15 = np.array(golden("frames")["camera"])
        cam15[10] = 2.0
    frames = []
    cases = [{}, {"RTIOW_WIDE": "1"}, {"RTIOW_WIDE": "1", "RTIOW_RECHECK_ALL_LEAVES": "1"}, {"RTIOW_WIDE": "1", "RTIOW_WIDE_STACK_CAP": "4"},
             {"RTIOW_WIDE": "1", "RTIOW_RENDER_LDS_NODES": "0"}, {"RTIOW_WIDE": "1", "RTIOW_RENDER_LDS_NODES": "0", "RTIOW_WIDE_STACK_CAP": "4", "RTIOW_RECHECK_ALL_LEAVES": "1"},
             {"RTIOW_WIDE": "1", "RTIOW_TREE": "reference"}, {"RTIOW_WIDE": "1", "RTIOW_RENDER_WAVES": "3"}]
    monkeypatch.setenv("RTIOW_VERBOSE", "1")
    for env in cases:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        capfd.readouterr()
        w = rt.World(scene_path)
        said = capfd.readouterr().err
        for k in env:
            monkeypatch.delenv(k, raising=False)
        if env:     # the shape asked for is the shape in use
            in_lds = "RTIOW_RENDER_LDS_NODES" not in env and scene != "stress4096"     # (4096 spheres do not fit)
            assert ("four-wide tree whole in LDS" if in_lds else "four-wide tree in device memory") in said, said
            assert ("768 threads" in said) == ("RTIOW_RENDER_WAVES" in env and in_lds)
        else:
            assert "four-wide" not in said
        try:
            frames.append(w.render(rt.camera(cam15), 72, 36, 0, 5, 33))
            frames.append(w.render(rt.camera(cam15), 72, 36, 0, 40, 33))     # deep enough for the costing pass and ordered tiles
        finally:
            w.close()
    exp, counters = orc.Scene(scene_path).render(cam15, 72, 36, 0, 5, 33, orc.ORC_RNG_STREAM)
    for k, (f, st) in enumerate(frames):
        assert np.array_equal(f, frames[k % 2][0]), cases[k // 2]
        assert st.rays == frames[k % 2][1].rays, cases[k // 2]
    assert frames[0][1].rays == counters.rays
    np.testing.assert_allclose(frames[0][0], exp, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("scene", ["book1", "stress4096", "ties"])
def test_grazing_rays_reach_their_spheres(rt, orc, scene):
    # In a scene without moving spheres every sphere sits behind a box of its own (single precision,
    # padded, tested conservatively).  A ray the reference's sphere test accepts by the last bit of its
    # discriminant must still get through that box: rays aimed at the limb of random spheres, from near
    # and far, off by 0, a few ulps, ... 1e-9 of the radius either way, must give the oracle's hits.
    spheres = np.array([[float(x) for x in l.split()[2:6]] for l in open(recipe(scene)) if l.startswith("sphere ")])
    rng = np.random.default_rng(77)
    n = 60000
    pick = spheres[rng.integers(0, len(spheres), n)]
    c, r = pick[:, :3], pick[:, 3]
    axis = rng.normal(size=(n, 3)); axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    side = np.cross(axis, rng.normal(size=(n, 3))); side /= np.linalg.norm(side, axis=1, keepdims=True)
    # The rounding of the reference's discriminant grows with the distance L of the origin: lines up
    # to 2^-50 L^2/r outside the sphere can still be "hit" (9 % of the radius at L = 10^7 r), so the
    # offsets reach that far too.
    eps = rng.choice([0.0, 1e-16, -1e-16, 4e-16, -4e-16, 1e-14, -1e-14, 1e-12, -1e-12, 1e-9, -1e-9, 1e-6, -1e-6,
                      1e-4, -1e-4, 3e-3, -3e-3, 0.03, -0.03, 0.08], n)
    limb = c + side * (np.abs(r) * (1.0 + eps))[:, None]
    dist = np.abs(r)[:, None] * rng.choice([1.5, 10.0, 300.0, 5000.0, 1e5, 1e6, 1e7, 1e9], n)[:, None]
    o = limb - axis * dist
    d = axis * rng.choice([1.0, 0.37, 12.5], n)[:, None]
    rays = np.hstack([o, d, np.zeros((n, 1))])
    w = rt.World(recipe(scene))
    try:
        h = w.trace(rays, 0.001, np.finfo(np.float64).max)
    finally:
        w.close()
    hit, prim, tpn = orc.Scene(recipe(scene)).trace(rays, 0.001, np.finfo(np.float64).max)
    assert np.array_equal(h["hit"], hit.astype(np.int32))
    assert np.array_equal(h["prim"], prim)
    m = hit == 1
    assert np.array_equal(h["t"][m], tpn[m, 0])
    assert 0.2 < m.mean() < 1.0
    far = np.linalg.norm(o - c, axis=1) > 9e4 * np.abs(r)
    assert far.sum() > 20000 and m[far].sum() > 1000             # half of the origins are that far away


def test_cost_ordered_tiles_render_the_same_frame(rt, worlds, monkeypatch):
    # Deep frames with many tiles get a costing pass and are rendered dearest tile first.  Which wave
    # renders which tile when must not change a bit of the image or the ray count.
    cam15 = np.array(golden("frames")["camera"])
    cam15[10] = 800 / 640
    w = worlds("book1")
    for accum in (rt.RT_ACCUM_F64, rt.RT_ACCUM_F32):
        ordered, s_ordered = w.render(rt.camera(cam15), 800, 640, 3, 3 + 32, 8, accum)
        monkeypatch.setenv("RTIOW_TILE_ORDER_OFF", "1")
        plain, s_plain = w.render(rt.camera(cam15), 800, 640, 3, 3 + 32, 8, accum)
        monkeypatch.delenv("RTIOW_TILE_ORDER_OFF")
        assert s_ordered.rays == s_plain.rays and s_ordered.paths == s_plain.paths == 800 * 640 * 32
        assert np.array_equal(ordered, plain)
        assert ordered.any()
        # ... nor must a strip of the frame and a tile queue per XCD (the default for trees in device memory; forced here)
        monkeypatch.setenv("RTIOW_TILE_REGIONS", "8")
        strips, s_strips = w.render(rt.camera(cam15), 800, 640, 3, 3 + 32, 9, accum)
        monkeypatch.setenv("RTIOW_TILE_REGIONS", "1")
        single, s_single = w.render(rt.camera(cam15), 800, 640, 3, 3 + 32, 9, accum)
        monkeypatch.delenv("RTIOW_TILE_REGIONS")
        assert s_strips.rays == s_single.rays and s_strips.paths == 800 * 640 * 32
        assert np.array_equal(strips, single) and strips.any() and not np.array_equal(strips, ordered)
        # ... also for a frame of few samples, whose tiles are not ordered by cost (a strip's tiles then go row by row)
        monkeypatch.setenv("RTIOW_TILE_REGIONS", "8")
        strips, s_strips = w.render(rt.camera(cam15), 800, 640, 2, 2 + 6, 9, accum)
        monkeypatch.setenv("RTIOW_TILE_REGIONS", "1")
        single, s_single = w.render(rt.camera(cam15), 800, 640, 2, 2 + 6, 9, accum)
        monkeypatch.delenv("RTIOW_TILE_REGIONS")
        assert s_strips.rays == s_single.rays and s_strips.paths == 800 * 640 * 6
        assert np.array_equal(strips, single) and strips.any()


EDGE_FRAMES = [
    # nx, ny, first sample, one past the last, max depth, aperture
    (1, 1, 0, 3, 50, 0.0),        # a single pixel: one ragged tile
    (9, 3, 4, 5, 50, 0.0),        # one sample that is not sample 0
    (16, 8, 0, 0, 50, 0.0),       # an empty sample range adds nothing
    (24, 12, 0, 4, 0, 0.0),       # depth 0: scatter is still evaluated, then the path ends black (main.cpp:33-42)
    (24, 12, 0, 4, 1, 0.0),
    (40, 20, 0, 6, 50, 0.5),      # a real lens: the disk rejection loop runs (camera.cpp:16-21,39)
]


@pytest.mark.parametrize("nx,ny,s0,s1,depth,aperture", EDGE_FRAMES)
def test_edge_frames_match_oracle(rt, orc, worlds, nx, ny, s0, s1, depth, aperture):
    cam15 = np.array(golden("frames")["camera"])
    cam15[10] = nx / ny
    cam15[11] = aperture
    got, stats = worlds("book2").render(rt.camera(cam15), nx, ny, s0, s1, 11, rt.RT_ACCUM_F64, max_depth=depth)
    exp, counters = orc.Scene(recipe("book2")).render(cam15, nx, ny, s0, s1, 11, orc.ORC_RNG_STREAM, max_depth=depth)
    assert stats.paths == nx * ny * (s1 - s0) == counters.paths
    assert stats.rays == counters.rays
    np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-13)
    if s1 == s0:
        assert not got.any()


@pytest.mark.parametrize("scene,distance", [("book1", 1e5), ("book1", 3e7), ("book2", 1e6), ("stress4096", 1e6)])
def test_frames_from_far_away_match_oracle(rt, orc, worlds, scene, distance):
    # A camera far outside the scene: the rounding of the reference's sphere test then reaches a
    # sizeable part of the small spheres' radii (2^-50 L^2 / r), beyond the padding of the boxes the
    # static scenes give each sphere, and every segment has to widen the boxes it tests by that slack.
    cam15 = np.array(golden("frames")["camera"])
    cam15[0:3] = np.array([13.0, 4.0, 3.0]) * (distance / 13.9)
    cam15[9] = np.degrees(2 * np.arctan(6.0 / distance))      # the scene's middle fills the frame
    cam15[12] = distance
    got, stats = worlds(scene).render(rt.camera(cam15), 96, 48, 0, 3, 17, rt.RT_ACCUM_F64)
    exp, counters = orc.Scene(recipe(scene)).render(cam15, 96, 48, 0, 3, 17, orc.ORC_RNG_STREAM)
    assert stats.rays == counters.rays
    np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-13)
    assert stats.rays > 1.5 * stats.paths                       # the rays do meet the scene


def test_radiance_beyond_one_and_deep_calls(rt, orc, tmp_path):
    # A pixel's samples are summed as integers in units of 2^-50, which holds radiance up to the sky's 1;
    # a scene with albedos beyond 1 (nothing in the reference forbids them) is summed in doubles instead.
    # And a launch adds at most 4096 samples to a pixel: deeper calls are cut into several.
    path = tmp_path / "bright.recipe"
    path.write_text("rtiow-recipe 1\ntexture constant 1.9 0.4 2.5\ntexture constant 0.9 1.6 0.2\nmaterial lambertian 0\n"
                    "material metal 2.2 1.4 0.7 0.2\nmaterial lambertian 1\nmaterial dielectric 1.5\n"
                    "sphere 0 0 -100.5 -1 100\nbvh_begin 3 0 1\nsphere 1 0 0 -1 0.5\nsphere 2 1 0 -1 0.5\nsphere 3 -1 0 -1 0.5\nbvh_end\n")
    cam15 = np.array(golden("frames")["camera"])
    cam15[0:3] = [0.0, 1.0, 3.0]; cam15[3:6] = [0.0, 0.0, -1.0]; cam15[9] = 50.0
    w = rt.World(str(path))
    try:
        got, stats = w.render(rt.camera(cam15), 64, 32, 0, 6, 5, rt.RT_ACCUM_F64)
        exp, counters = orc.Scene(str(path)).render(cam15, 64, 32, 0, 6, 5, orc.ORC_RNG_STREAM)
        assert stats.rays == counters.rays and got.max() > 6.5          # some pixels are brighter than any sky
        np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-13)
        deep, sd = w.render(rt.camera(cam15), 24, 12, 0, 4100, 6, rt.RT_ACCUM_F64)
        a, sa = w.render(rt.camera(cam15), 24, 12, 0, 4096, 6, rt.RT_ACCUM_F64)
        b, sb = w.render(rt.camera(cam15), 24, 12, 4096, 4100, 6, rt.RT_ACCUM_F64)
        assert sd.rays == sa.rays + sb.rays and sd.paths == 24 * 12 * 4100
        np.testing.assert_allclose(deep, a + b, rtol=1e-13, atol=1e-13)
    finally:
        w.close()


def test_empty_world_is_all_sky(rt, orc, tmp_path):
    # object_list::hit over no objects never hits (object_list.cpp:12-26): every path is one sky ray
    path = tmp_path / "empty.recipe"
    path.write_text("rtiow-recipe 1\n")
    cam15 = np.array(golden("frames")["camera"])
    w = rt.World(str(path))
    try:
        got, stats = w.render(rt.camera(cam15), 20, 10, 0, 2, 5, rt.RT_ACCUM_F64)
        exp, counters = orc.Scene(str(path)).render(cam15, 20, 10, 0, 2, 5, orc.ORC_RNG_STREAM)
        assert stats.rays == stats.paths == 400 == counters.rays
        np.testing.assert_allclose(got, exp, rtol=1e-13)
        rays = np.zeros((3, 7)); rays[:, 5] = 1.0
        assert not w.trace(rays, 0.001, 1e30)["hit"].any()
    finally:
        w.close()
