"""Pins the oracle (oracle/rt_oracle.c) to the compiled reference.

The reference has no tests or fixtures of its own; tests/golden/ holds outputs of the
reference's classes captured by oracle/make_goldens.py.  Every comparison here is
bit-exact unless a libm call is involved.
"""
import ctypes

import numpy as np
import pytest

from conftest import HIT_SCENES, TRACE_SCENES, golden, recipe


def test_mt19937_known_answer(orc):
    # the C++ standard fixes the 10000th output of a default-seeded mt19937
    st = ctypes.create_string_buffer(624 * 4 + 8)
    orc.lib().orc_mt_seed(st, 5489)
    v = 0
    for _ in range(10000):
        v = orc.lib().orc_mt_next(st)
    assert v == 4123659995


def test_philox_known_answers(orc):
    # Random123 known-answer vectors for philox4x32-10
    def run(ctr, key):
        c = (ctypes.c_uint32 * 4)(*ctr)
        k = (ctypes.c_uint32 * 2)(*key)
        o = (ctypes.c_uint32 * 4)()
        orc.lib().orc_philox4x32_10(k, c, o)
        return [hex(x) for x in o]
    assert run([0, 0, 0, 0], [0, 0]) == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_stream_uniforms_are_uniform(orc):
    st = ctypes.create_string_buffer(16)
    orc.lib().orc_stream_init(st, 12345, 7, 3)
    u = np.array([orc.lib().orc_stream_canonical(st) for _ in range(20000)])
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005


def test_aabb_hit(orc):
    g = golden("aabb")
    got = orc.aabb_batch(g["cases"])
    assert np.array_equal(got, g["hit"])
    assert 0 < g["hit"].sum() < len(g["hit"])


@pytest.mark.parametrize("scene", TRACE_SCENES)
def test_bvh_topology(orc, scene):
    g = golden("trace_" + scene)
    nodes = orc.Scene(recipe(scene)).nodes()
    assert nodes.shape == g["topology"].shape
    assert np.array_equal(nodes, g["topology"])   # boxes bit-exact, children identical


@pytest.mark.parametrize("scene", HIT_SCENES)
def test_object_hit(orc, scene):
    g = golden("trace_" + scene)
    s = orc.Scene(recipe(scene))
    for rays, tmin, tmax, exp in [(g["rays"], float(g["tmin"]), float(g["tmax"]), g["expected"]),
                                  (g["rays2"], float(g["tmin2"]), float(g["tmax2"]), g["expected2"])]:
        hit, prim, tpn = s.trace(rays, tmin, tmax)
        assert np.array_equal(hit, exp[:, 0].astype(np.uint8))
        assert np.array_equal(prim, exp[:, 1].astype(np.int32))
        assert np.array_equal(tpn, exp[:, 2:9])   # t, p, normal: bit-exact


def test_perlin_tables_and_noise(orc):
    g = golden("texture")
    s = orc.Scene(recipe("two_perlin"))
    got = s.texture(0, g["points"])
    assert np.array_equal(got, g["perlin"])
    # the two constants verified against the reference during the survey
    assert got[4000, 0] == 0.50292955069550871
    assert got[4001, 0] == 0.21579950322084329


def test_chequer_texture(orc):
    g = golden("texture")
    s = orc.Scene(recipe("book2"))
    got = s.texture(2, g["points"])
    assert np.array_equal(got, g["chequer"])
    assert len(np.unique(got[:, 0])) == 2


@pytest.mark.parametrize("scene", ["book2", "mixed"])
def test_scatter(orc, scene):
    g = golden("scatter_" + scene)
    items, exp, k = g["items"], g["expected"], int(g["k"])
    s = orc.Scene(recipe(scene))
    sc, att, out, used = s.scatter(items[:, 0].astype(np.int32), items[:, 1:8], items[:, 8:15], exp[:, 12:12 + k])
    assert np.array_equal(used, exp[:, 11].astype(np.int32))
    assert np.array_equal(sc, exp[:, 0].astype(np.uint8))
    m = sc == 1
    assert np.array_equal(att[m], exp[m, 1:4])
    # directions are exact except where schlick's pow() decides between reflect and refract:
    # both sides call the same libm here, so they are exact too
    assert np.array_equal(out[m], exp[m, 4:11])


def test_camera_get_ray(orc):
    g = golden("camera")
    k = int(g["k"])
    for cam, exp in zip(g["cameras"], g["expected"]):
        out, used = orc.camera_batch(cam, g["st"], exp[:, 8:8 + k])
        assert np.array_equal(used, exp[:, 7].astype(np.int32))
        assert np.array_equal(out, exp[:, 0:7])
    # the shutter time reuses the {-1,1} distribution: times span [-1, 1), not [0, 1)
    t = g["expected"][0][:, 6]
    assert t.min() < -0.9 and t.max() > 0.9


@pytest.mark.parametrize("scene", ["two_perlin", "book1", "book2", "mixed"])
def test_frame_bit_exact(orc, scene):
    # same mt19937 seed, same draw order as main.cpp:126-141 -> the same frame, to the bit
    g = golden("frames")
    exp = g[scene + "_exact"]
    s = orc.Scene(recipe(scene))
    sums, counters = s.render(g["camera"], 48, 24, 0, 3, 31337, orc.ORC_RNG_MT)
    assert np.array_equal(sums / 3, exp)
    assert counters.paths == 48 * 24 * 3 and counters.rays >= counters.paths


def test_frame_bit_exact_with_lists_below_bvh_nodes(orc):
    # object_lists as objects of a bvh_node (bvh_node.cpp:34-40, object_list.cpp:12-48)
    g = golden("trace_lists")
    sums, counters = orc.Scene(recipe("lists")).render(g["camera"], 48, 24, 0, 3, 31337, orc.ORC_RNG_MT)
    assert np.array_equal(sums / 3, g["frame_exact"])
    prims = g["expected"][g["expected"][:, 0] > 0, 1].astype(int)
    # of coincident spheres a list keeps its first member (2 not 3, 12 not 13, 16 not 17)
    assert {2, 12, 16} <= set(prims) and not {3, 13, 17} & set(prims)


def test_stream_mode_matches_mt_mode_statistically(orc):
    # the per-(pixel, sample) streams must not bias the image
    g = golden("frames")
    s = orc.Scene(recipe("book2"))
    sums, _ = s.render(g["camera"], 120, 60, 0, 16, 5, orc.ORC_RNG_STREAM)
    ref = g["book2_spp256_seed1"].astype(np.float64)
    assert abs((sums / 16).mean() - ref.mean()) < 0.01


def test_stream_unit_sphere_sampler_is_uniform_in_the_ball(orc):
    # material.cpp:35-40 draws points of the cube until one lies inside the unit sphere.  The per-path
    # streams produce the same distribution without the loop: a direction by the equal-area octahedral
    # map of two 24-bit uniforms, a radius as the largest of three 32-bit uniforms.  The points must be
    # what the reference's are: strictly inside the unit sphere, radius^3 uniform, every coordinate with
    # the ball's marginal, the direction uniform on the sphere (z and the azimuth uniform, octants equally
    # filled) and independent of the radius.
    from scipy import stats
    n = 1000000
    p = orc.stream_unit_sphere(7, 123, 45, n)
    r2 = (p * p).sum(axis=1)
    assert (r2 < 1).all() and r2.max() > 0.999
    r = np.sqrt(r2)
    assert stats.kstest(r2 ** 1.5, "uniform").pvalue > 1e-3
    for a in range(3):
        assert stats.kstest(p[:, a], lambda t: (2 + 3 * t - t ** 3) / 4).pvalue > 1e-3
    u = p / r[:, None]
    assert stats.kstest((u[:, 2] + 1) / 2, "uniform").pvalue > 1e-3
    assert stats.kstest((np.arctan2(u[:, 1], u[:, 0]) + np.pi) / (2 * np.pi), "uniform").pvalue > 1e-3
    octant = (u[:, 0] > 0) * 4 + (u[:, 1] > 0) * 2 + (u[:, 2] > 0)
    assert stats.chisquare(np.bincount(octant, minlength=8)).pvalue > 1e-3
    # a finer look at the sphere: 48 equal-area cells (6 bands in z x 8 sectors in azimuth)
    cell = np.minimum((u[:, 2] + 1) * 3, 5).astype(int) * 8 + np.minimum((np.arctan2(u[:, 1], u[:, 0]) + np.pi) * (4 / np.pi), 7).astype(int)
    assert stats.chisquare(np.bincount(cell, minlength=48)).pvalue > 1e-3
    for a in range(3):
        assert abs(np.corrcoef(r, u[:, a])[0, 1]) < 0.005 and abs(np.corrcoef(r, np.abs(u[:, a]))[0, 1]) < 0.005
    # second moments of an isotropic ball: E[x^2] = 1/5, E[xy] = 0
    assert np.allclose((p * p).mean(axis=0), 0.2, atol=2e-3) and abs((p[:, 0] * p[:, 2]).mean()) < 1e-3
    # the same points from the reference's own loop over a generator of its own: two samples of one distribution
    rng = np.random.default_rng(5)
    q = rng.uniform(-1, 1, (3 * n, 3))
    q = q[(q * q).sum(axis=1) < 1][:n]
    for a in range(3):
        assert stats.ks_2samp(p[:, a], q[:, a]).pvalue > 1e-3
    assert stats.ks_2samp(r, np.sqrt((q * q).sum(axis=1))).pvalue > 1e-3
    assert abs(np.corrcoef(p[:, 0], p[:, 1])[0, 1]) < 0.01 and abs(np.corrcoef(p[:, 1], p[:, 2])[0, 1]) < 0.01
