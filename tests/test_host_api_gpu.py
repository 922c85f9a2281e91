"""The host scene API (rtiow.hpp) used the way the reference's own classes are used -- one
object::hit / scatter / value / get_ray call at a time -- checked against the golden vectors.
Every one of those calls runs on the GPU through the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden, recipe

pytestmark = pytest.mark.gpu
TOOL = os.path.join(ROOT, "ray-tracing-in-one-weekend_amd", "csrc", "host_api_check")


def run(*args):
    return subprocess.run([TOOL, *map(str, args)], check=True, capture_output=True, text=True, timeout=300).stdout


def put(path, a):
    np.ascontiguousarray(a, dtype="<f8").tofile(path)


def get(path, cols):
    return np.fromfile(path, dtype="<f8").reshape(-1, cols)


def decode_png(data):
    """8-bit RGB, filter 0 only -- what frame::write_png emits -- with every checksum verified."""
    import struct
    import zlib
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(data):
        n, kind = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        assert struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0] == zlib.crc32(kind + body)
        chunks.append((kind, body))
        at += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, ctype, comp, filt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype, comp, filt, lace) == (8, 2, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("scene", ["mixed", "book2", "ties", "lists"])
def test_object_hit_one_ray_at_a_time(tmp_path, scene):
    g = golden("trace_" + scene)
    rays, exp = g["rays"][-300:], g["expected"][-300:]
    put(tmp_path / "r.bin", rays)
    run("hit", recipe(scene), tmp_path / "r.bin", repr(float(g["tmin"])), "max", tmp_path / "o.bin")
    assert np.array_equal(get(tmp_path / "o.bin", 9), exp)


def test_scatter_consumes_the_shared_generator_like_the_reference(tmp_path):
    # same seed, same items in the same order as the capture: identical outputs AND identical
    # numbers of draws, i.e. the caller's std::mt19937 ends every call in the reference's state
    g = golden("scatter_mixed")
    k = int(g["k"])
    items, exp = g["items"], g["expected"]
    # the capture dropped the few items that ran past K draws; replaying a prefix needs the
    # generator in step, so stop at the first gap
    put(tmp_path / "s.bin", items[:400])
    run("scatter", recipe("mixed"), 99, k, tmp_path / "s.bin", tmp_path / "o.bin")
    got = get(tmp_path / "o.bin", 12 + k)
    n = 0
    while n < 400 and np.array_equal(got[n], exp[n]):
        n += 1
    assert n >= 100, n     # bit-equal, draws included, until the first dropped item shifts the stream
    assert np.array_equal(got[:n, 11], exp[:n, 11])


def test_camera_get_ray_consumes_the_shared_generator(tmp_path):
    g = golden("camera")
    k = int(g["k"])
    put(tmp_path / "st.bin", g["st"][:300])
    for cam, exp in zip(g["cameras"], g["expected"]):
        run("camera", 5, k, *[repr(float(x)) for x in cam], tmp_path / "st.bin", tmp_path / "o.bin")
        assert np.array_equal(get(tmp_path / "o.bin", 8 + k), exp[:300])


def test_texture_value_and_perlin_noise(tmp_path):
    g = golden("texture")
    put(tmp_path / "p.bin", g["points"][-260:])
    run("texture", "chequer", tmp_path / "p.bin", tmp_path / "o.bin")
    assert np.array_equal(get(tmp_path / "o.bin", 3), g["chequer"][-260:])
    run("texture", "perlin", tmp_path / "p.bin", tmp_path / "o.bin")
    assert np.array_equal(get(tmp_path / "o.bin", 3), g["perlin"][-260:])
    run("texture", "noise", tmp_path / "p.bin", tmp_path / "o.bin")
    assert np.array_equal(get(tmp_path / "o.bin", 3), g["perlin"][-260:])


def test_aabb_hit(tmp_path):
    g = golden("aabb")
    put(tmp_path / "c.bin", g["cases"][:500])
    run("aabb", tmp_path / "c.bin", tmp_path / "o.bin")
    assert np.array_equal(get(tmp_path / "o.bin", 1)[:, 0].astype(np.uint8), g["hit"][:500])


def test_errors_and_builders():
    out = run("errors")
    assert "caught: no bounding box in bvh_node constructor" in out
    assert "empty: 0" in out
    assert "mean gamma: 0.70710678118654757 0.70710678118654757 0.70710678118654757" in out
    assert "two_perlin: 3200 paths" in out and "build_world: 3200 paths" in out
    # render_options::devices = {0} is the single-GPU frame; {0, 0} splits the samples and traces the same rays
    line = [l for l in out.splitlines() if l.startswith("devices:")][0]
    assert "one listed equal" in line
    rays = [int(x) for x in line.replace(",", " ").split() if x.isdigit()]
    assert rays[0] == rays[1] and float(line.split()[-1]) < 1e-5


def test_frame_driver_writes_the_reference_ppm_and_its_binary_twin(tmp_path, rt):
    # the command-line driver stands where the reference's main() stands: P3 text on request,
    # "Render duration" on stderr (main.cpp:143-150); --binary writes the same values as P6
    exe = os.path.join(ROOT, "ray-tracing-in-one-weekend_amd", "csrc", "raytracer")
    common = [exe, "--scene", recipe("book2"), "--nx", "50", "--ny", "26", "--ns", "8", "--seed", "4"]
    r = subprocess.run(common + ["--out", str(tmp_path / "a.ppm")], check=True, capture_output=True, text=True, timeout=300)
    assert "Render duration:" in r.stderr
    subprocess.run(common + ["--out", str(tmp_path / "b.ppm"), "--binary"], check=True, capture_output=True, timeout=300)
    text = (tmp_path / "a.ppm").read_text().split()
    assert text[:4] == ["P3", "50", "26", "255"]
    p3 = np.array(text[4:], dtype=np.int64).reshape(26, 50, 3)
    raw = (tmp_path / "b.ppm").read_bytes()
    header = b"P6\n50 26\n255\n"
    assert raw.startswith(header) and len(raw) == len(header) + 50 * 26 * 3
    p6 = np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(26, 50, 3)
    assert np.array_equal(p3, p6)
    # .png: the same bytes again, in a container any viewer opens
    subprocess.run(common + ["--out", str(tmp_path / "c.png")], check=True, capture_output=True, timeout=300)
    assert np.array_equal(decode_png((tmp_path / "c.png").read_bytes()), p6)
    # --devices 0: the list of GPUs with one entry writes the same file
    subprocess.run(common + ["--out", str(tmp_path / "d.ppm"), "--devices", "0"], check=True, capture_output=True, timeout=300)
    assert (tmp_path / "d.ppm").read_text() == (tmp_path / "a.ppm").read_text()
    # and they are the resolve of the library's own sums for the same seed
    cam15 = np.array(golden("frames")["camera"])
    cam15[10] = 50 / 26
    w = rt.World(recipe("book2"))
    sums, _ = w.render(rt.camera(cam15), 50, 26, 0, 8, 4, rt.RT_ACCUM_F32)
    w.close()
    expect = (255.99 * np.sqrt(sums.astype(np.float64) / 8)).astype(np.int64)
    assert np.abs(expect - p3).max() <= 1 and (expect != p3).mean() < 0.01


def test_single_precision_node_test_is_conservative():
    # The inner nodes of the device tree are tested in single precision; everything exact rests on
    # that test never rejecting a box the reference's double-precision test accepts.  The checker
    # aims rays at faces, edges and corners of random boxes, nudged by ulps, at many scales.
    exe = os.path.join(ROOT, "ray-tracing-in-one-weekend_amd", "csrc", "conservative_check")
    r = subprocess.run([exe, "400000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 violations" in r.stdout


def test_division_by_a_shared_reciprocal_is_the_ieee_division():
    # The render kernel forms u = (x + xi) / nx from the refined reciprocal of nx, computed once per wave.
    # The checker compares that with the hardware's full IEEE division for every image size up to 8192
    # (and a spread of larger ones), 65 536 numerators each.
    exe = os.path.join(ROOT, "ray-tracing-in-one-weekend_amd", "csrc", "division_check")
    r = subprocess.run([exe, "65536"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " mismatches 0" in r.stdout


def test_unit_sphere_sampler_equals_the_oracles_point_for_point(orc, tmp_path):
    # The per-path streams sample the unit sphere without rejection: octahedral map in single precision, a
    # hardware square root corrected to the rounded one, IEEE division.  The oracle restates it in C; the two
    # must agree in every bit of every coordinate (frames agree to 1e-11 only if they do).
    exe = os.path.join(ROOT, "ray-tracing-in-one-weekend_amd", "csrc", "sampler_dump")
    out = tmp_path / "points.bin"
    lanes, n = 2048, 96
    for seed, pixel0, sample in [(7, 1000, 3), (2 ** 63 + 5, 4000000000, 4095)]:
        subprocess.run([exe, str(seed), str(pixel0), str(sample), str(lanes), str(n), str(out)], check=True, timeout=300)
        got = np.fromfile(out).reshape(lanes, n, 3)
        for i in range(0, lanes, 7):
            assert np.array_equal(got[i], orc.stream_unit_sphere(seed, (pixel0 + i) % 2 ** 32, sample, n)), (seed, i)
