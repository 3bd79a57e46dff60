"""CPU-side checks of the product's host layer and C-ABI library: it loads, exports every
symbol include/mrt.h declares, and the C++ host constructors agree bit-for-bit with the
oracle's independent restatement.  No GPU, no compute calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import _oracle as O
from raytracing_amd import capi, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = capi.libmrt()
    declared = _declared_symbols()
    assert declared == sorted(capi.MRT_SYMBOLS), "capi.MRT_SYMBOLS out of sync with include/mrt.h"
    for name in declared:
        assert hasattr(L, name), f"libmrt.so does not export {name}"


def test_status_strings_and_argument_validation_without_gpu():
    L = capi.libmrt()
    assert L.mrt_status_string(0) == b"ok"
    assert L.mrt_status_string(-3) == b"no scene uploaded"
    assert L.mrt_device_count(None) == -1
    r = C.c_int32(-1)
    bad = capi.Partition(2, 2, 8)  # rank out of range
    assert L.mrt_local_rows(1080, C.byref(bad), C.byref(r)) == -1
    bad = capi.Partition(0, 2, 12)  # strip not a multiple of 8
    assert L.mrt_local_rows(1080, C.byref(bad), C.byref(r)) == -1


@pytest.mark.parametrize("H,n,strip", [(1080, 1, 8), (1080, 2, 8), (1080, 8, 8), (1083, 3, 8), (40, 8, 8), (2160, 4, 16)])
def test_strip_partition_covers_every_row_once(H, n, strip):
    seen = np.zeros(H, dtype=int)
    for r in range(n):
        rows = capi.local_row_index(H, r, n, strip)
        assert capi.local_rows(H, r, n, strip) == len(rows)
        seen[rows] += 1
    assert (seen == 1).all()


def test_viewport_matches_oracle_bitwise():
    for aspect in (2.0, 16.0 / 9.0, 4.0 / 3.0):
        assert np.array_equal(capi.default_viewport(aspect), O.default_viewport(aspect))
    a = capi.viewport((1, 2, 3), (-4, 0.5, -9), (0.1, 1, 0), 37.5, 1.25)
    b = O.viewport((1, 2, 3), (-4, 0.5, -9), (0.1, 1, 0), 37.5, 1.25)
    assert np.array_equal(a, b)


def test_triangle_constructor_matches_oracle_bitwise():
    spec = scenes.light_stress(16, 64)
    t16 = capi.triangle_precompute(spec.tri_verts)
    for i in range(spec.tri_verts.shape[0]):
        assert np.array_equal(t16[:, i], O.triangle_precompute(*spec.tri_verts[i]))
    # the reference's own test triangles, including the degenerate one through the origin
    for tri in ([(0, 1, -1), (-1, -1, -1), (1, -1, -1)], [(0, -1, 0), (1, 1, -1), (-1, 1, 1)]):
        assert np.array_equal(capi.triangle_precompute(np.float32([tri]))[:, 0], O.triangle_precompute(*tri))


def test_default_scene_flattening():
    flat, cam = scenes.default_scene()
    assert (flat.num_objects, flat.num_spheres, flat.num_triangles, flat.num_lights) == (10, 4, 6, 1)
    assert list(flat.sphere_obj) == [0, 2, 4, 9] and list(flat.tri_obj) == [1, 3, 5, 6, 7, 8]  # src/main.cu:210-233
    assert list(flat.mat_kind) == [0, 0, 0, 0, 0, 1, 1, 0, 0, 1]
    assert np.allclose(np.linalg.norm(flat.light_vec[0]), 1.0, atol=1e-6)  # DirectionalLight normalises
    assert np.array_equal(flat.light_vec[0], capi.directional_light_dir(scenes.DEFAULT_LIGHT_DIR))


def test_generators_are_deterministic():
    a, b = scenes.sphere_field(500), scenes.sphere_field(500)
    assert np.array_equal(a.spheres, b.spheres)
    t = scenes.triangle_terrain(20, 10)
    assert t.tri_verts.shape == (400, 3, 3)
    s = scenes.light_stress()
    assert len(s.lights) == 64 and s.num_objects == 512
    assert sorted(list(s.sphere_obj) + list(s.tri_obj)) == list(range(512))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(capi, "_libmrt", None)
    monkeypatch.setattr(capi, "LIBMRT_PATH", "/nonexistent/libmrt.so")
    with pytest.raises(capi.MrtError):
        capi.libmrt()


def test_cpp_host_layer_compiles_standalone(tmp_path):
    """Every host header is self-contained C++17 (no HIP headers on the host side)."""
    host = os.path.join(ROOT, "raytracing_amd", "host")
    src = tmp_path / "all.cpp"
    hdrs = []
    for d in ("math", "core", "pipeline", "container", "hip_utils", "io"):
        hdrs += [f"{d}/{f}" for f in sorted(os.listdir(os.path.join(host, d))) if f.endswith(".hpp")]
    src.write_text("".join(f'#include "{h}"\n' for h in hdrs) + "int main() { return 0; }\n")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", host, "-I", os.path.join(ROOT, "include"),
                           str(src)])


def test_cpp_host_layer_cpu_tests():
    """raytracing_amd/host/test/host_cpu_tests.cpp: Transform / apply_transform, Interval modes, work partition,
    ImageBuffer ownership and flip, default-scene flattening -- the C++ host layer without a GPU."""
    exe = os.path.join(ROOT, "raytracing_amd", "lib", "host_cpu_tests")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracing_amd", "host")], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and ", 0 failed" in out.stdout, out.stdout + out.stderr


def test_triangle_bound_of_a_sliver_with_open_edge_planes_is_unbounded():
    """A sliver whose float edge planes do not close: the pairwise intersection of edge planes 0 and 1 lies
    1e-8 (relative) OUTSIDE edge plane 2, so the region the hit test accepts is an unbounded wedge and the
    oracle accepts points more than 20 units from the geometric triangle (size 0.02).  The bound must be +inf.
    Found by tools/stress_parity.py, seed 516."""
    tri = np.array([[-6.747651, 12.751217, -2.9846556],
                    [-6.737617, 12.738894, -2.9723556],
                    [-6.7375774, 12.738748, -2.9724247]], np.float32)
    t16 = capi.triangle_precompute(tri)[:, 0]
    q = t16.astype(np.float64)
    # a far point of the plane that all three edge tests accept (found by walking along the wedge)
    n = q[:3] / np.linalg.norm(q[:3])
    far_hits = 0
    centre = tri.astype(np.float64).mean(0)
    rng = np.random.default_rng(3)
    for _ in range(4000):
        dirn = rng.normal(size=3); dirn -= n * (n @ dirn); dirn /= np.linalg.norm(dirn)
        pt = centre + dirn * 20.0
        pt -= n * ((q[:3] @ pt - q[3]) / np.linalg.norm(q[:3]))
        origin = (pt + n * 5.0).astype(np.float32)
        ok, _, point, _ = O.triangle16_hit(t16, origin, (pt - origin).astype(np.float32), 0.01, float("inf"))
        far_hits += bool(ok)
    assert far_hits > 0, "the oracle should accept points 20 units from this sliver"
    assert np.isinf(capi.triangle_bound(t16)[3])


def test_triangle_bound_contains_every_accepted_point():
    """mrt_triangle_bound (host only): the ball the culling kernels use must contain every point the exact
    hit test (oracle) accepts.  Rays are shot at points inside and just outside the triangle; whenever the
    oracle reports a hit, the hit point lies inside the ball.  A triangle whose plane passes through the
    world origin has degenerate edge planes (dir x point_inside with a position vector): unbounded, +inf."""
    rng = np.random.default_rng(11)
    n_finite = 0
    for it in range(60):
        a = rng.uniform(-20, 20, 3)
        size = 10.0 ** rng.uniform(-1.5, 1.3)
        tri = np.stack([a, a + rng.normal(size=3) * size, a + rng.normal(size=3) * size]).astype(np.float32)
        t16 = capi.triangle_precompute(tri)[:, 0]
        ball = capi.triangle_bound(t16)
        if not np.isfinite(ball[3]):
            continue
        n_finite += 1
        w = rng.uniform(-0.3, 1.3, (200, 3)); w /= w.sum(1, keepdims=True)
        targets = (w[:, :, None] * tri.astype(np.float64)[None]).sum(1)
        origin = a + rng.normal(size=3) * 30.0
        for tgt in targets:
            d = (tgt - origin).astype(np.float32)
            ok, _, point, _ = O.triangle16_hit(t16, origin.astype(np.float32), d, 0.01, float("inf"))
            if ok:
                dist = np.linalg.norm(point.astype(np.float64) - ball[:3].astype(np.float64))
                assert dist <= ball[3], (it, dist, ball)
    assert n_finite >= 30
    # vertices are inside their own triangle's ball
    tri = np.array([[1, 2, -5], [4, 2, -6], [2, 5, -7]], np.float32)
    ball = capi.triangle_bound(capi.triangle_precompute(tri)[:, 0])
    assert np.isfinite(ball[3]) and all(np.linalg.norm(v - ball[:3]) <= ball[3] for v in tri)
    # plane through the world origin: never culled
    tri0 = np.array([[0, 0, 0], [4, 1, -6], [2, 5, -7]], np.float32)
    assert capi.triangle_bound(capi.triangle_precompute(tri0)[:, 0])[3] == np.inf
