"""Pins the CPU oracle: the reference's own known-answer tests and the survey's full-frame
anchors (tests/golden).  No GPU."""
import json
import os

import numpy as np
import pytest

import _oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


KA = _load("reference_known_answers.json")
ANCHORS = _load("survey_anchors.json")


def _expected_normal(case, t):
    n = case["normal"]
    if n == "from_hit":  # test/spheretest.cu:120-121
        p = np.float32(case["origin"]) + np.float32(t) * np.float32(case["dir"])
        v = p - np.float32(case["center"])
        return v / np.sqrt((v * v).sum())
    if n == "normalized(1,0,1)":
        return np.float32([1, 0, 1]) / np.sqrt(np.float32(2))
    return np.float32(n)


@pytest.mark.parametrize("case", KA["sphere"], ids=lambda c: c["name"])
def test_sphere_known_answers(case):
    lo, hi = KA["interval"]
    ok, t, _, n = O.sphere_hit(case["center"], case["radius"], case["origin"], case["dir"], lo, hi)
    assert ok
    assert abs(float(t) - case["t"]) <= KA["tolerance"]
    assert np.all(np.abs(n - _expected_normal(case, t)) <= KA["tolerance"])


@pytest.mark.parametrize("case", KA["triangle"], ids=lambda c: c["name"])
def test_triangle_known_answers(case):
    lo, hi = KA["interval"]
    ok, t, _, n = O.triangle_hit(case["a"], case["b"], case["c"], case["origin"], case["dir"], lo, hi)
    assert ok
    assert abs(float(t) - case["t"]) <= KA["tolerance"]
    assert np.all(np.abs(n - _expected_normal(case, t)) <= KA["tolerance"])


@pytest.mark.parametrize("frame", ANCHORS["frames"], ids=lambda f: f"{f['width']}x{f['height']}")
def test_survey_frame_anchors(frame):
    W, H = frame["width"], frame["height"]
    r = O.Scene.default().render(O.default_viewport(ANCHORS["aspect"]), W, H, nthreads=4)
    ids, st = r["id"], r["stats"]
    assert int((ids == -1).sum()) == frame["miss"]
    assert [int((ids == i).sum()) for i in range(10)] == frame["id_histogram"]
    assert st["hit_pixels"] == frame["hit_pixels"]
    assert st["shadow_rays"] == frame["shadow_rays"]
    assert st["hit_pixels"] - st["shadow_rays"] == frame["light_skipped"]
    assert st["occluded"] == frame["occluded"]
    assert st["lit"] == frame["lit"]
    assert st["primary_rays"] + st["shadow_rays"] == frame["total_rays"]
    if "mean_rgb" in frame:
        mean = r["rgb"].reshape(-1, 3).mean(axis=0, dtype=np.float64)
        assert np.all(np.abs(mean - np.array(frame["mean_rgb"])) < 1e-6)
        assert abs(float(r["rgb"].max()) - frame["max_channel"]) < 1e-6
        assert np.allclose(r["rgb"][0, 0], frame["pixel_0_0"], rtol=0, atol=1e-7)
        mirror = r["rgb"][(ids == 5) | (ids == 6)]
        # unlit mirror pixels: 0.01*1 + 0.001*1
        assert np.isclose(np.median(mirror), frame["mirror_pixel_value"], atol=1e-7)


def test_threads_do_not_change_the_frame():
    s = O.Scene.default()
    cam = O.default_viewport(4.0 / 3.0)
    a = s.render(cam, 160, 120, nthreads=1)
    b = s.render(cam, 160, 120, nthreads=5)
    for k in ("rgb", "id", "t"):
        assert np.array_equal(a[k], b[k])
    assert a["stats"] == b["stats"]


def test_row_range_matches_full_frame():
    s = O.Scene.default()
    cam = O.default_viewport(2.0)
    full = s.render(cam, 128, 64)
    part = s.render(cam, 128, 64, rows=(16, 40))
    assert np.array_equal(full["id"][16:40], part["id"][16:40])
    assert np.array_equal(full["rgb"][16:40], part["rgb"][16:40])


def test_closest_hit_tie_keeps_lowest_index():
    # two identical spheres: the first one must own every hit pixel (src/main.cu:70 strict <)
    s = O.Scene()
    s.add_sphere((0, 0, -5), 1.0, O.MAT_DIFFUSE, (0.5, 0.5, 0.5))
    s.add_sphere((0, 0, -5), 1.0, O.MAT_DIFFUSE, (0.5, 0.5, 0.5))
    s.add_directional_light((0, -1, -1), (1, 1, 1))
    cam = O.viewport((0, 0, 0), (0, 0, -1), (0, 1, 0), 40.0, 1.0)
    r = s.render(cam, 64, 64)
    assert (r["id"] == 0).sum() > 0 and (r["id"] == 1).sum() == 0


def test_plane_never_hits_but_keeps_its_index():
    s = O.Scene()
    s.add_plane(0.0, (0, 1, 0), O.MAT_DIFFUSE, (1, 1, 1))  # core/plane.cuh:25-29
    s.add_sphere((0, 0, -5), 1.0, O.MAT_DIFFUSE, (0.5, 0.5, 0.5))
    cam = O.viewport((0, 0, 0), (0, 0, -1), (0, 1, 0), 40.0, 1.0)
    r = s.render(cam, 32, 32)
    assert set(np.unique(r["id"])) == {-1, 1}


def test_multisample_mode_runs_and_averages():
    s = O.Scene.default()
    cam = O.default_viewport(2.0)
    one = s.render(cam, 64, 32, spp=1)
    many = s.render(cam, 64, 32, spp=121)
    assert many["stats"]["primary_rays"] == 64 * 32 * 121
    # the top rows are sky for every sub-sample: jitter barely changes the gradient there
    assert (one["id"][-4:] == -1).all()
    assert np.abs(one["rgb"][-4:] - many["rgb"][-4:]).max() < 5e-3


def test_interval_modes():
    """core/interval.cuh:14-42: a root that IS a bound of the interval counts under a closed bound only (the reference's
    sphere known answer: roots t = 1 and t = 3; and a triangle, which has one)."""
    from raytracing_amd import capi
    sph, ro, rd = [0.0, 0.0, -2.0, 1.0], [0.0, 0.0, 0.0], [0.0, 0.0, -1.0]
    for lc in (True, False):
        for uc in (True, False):
            # (an open lower bound at the near root: Sphere::hit goes on to the far root, core/sphere.cuh:52-58)
            ok, t, _, _ = O.prim_hit_interval(capi.GEOM_SPHERE, sph, ro, rd, 1.0, 100.0, lc, uc)
            assert ok and t == (1.0 if lc else 3.0)
            assert O.prim_hit_interval(capi.GEOM_SPHERE, sph, ro, rd, -1.0, 1.0, lc, uc)[0] == uc
            assert O.prim_hit_interval(capi.GEOM_SPHERE, sph, ro, rd, -1.0, 100.0, lc, uc)[0]
            assert O.prim_hit_interval(capi.GEOM_SPHERE, sph, ro, rd, 1.0, 1.0, lc, uc)[0] == (lc and uc)
    ok, t, p, n = O.prim_hit_interval(capi.GEOM_SPHERE, sph, ro, rd, -1.0, 100.0)
    assert O.sphere_hit(sph[:3], sph[3], ro, rd, -1.0, 100.0)[1] == t == 1.0
    tri16 = capi.triangle_precompute(np.float32([[[0, -1, -3], [1, 1, -4], [-1, 1, -2]]]))[:, 0]   # test/triangletest.cu:72: t = 3
    for lc in (True, False):
        for uc in (True, False):
            assert O.prim_hit_interval(capi.GEOM_TRIANGLE, tri16, [0, 0.5, 0], [0, 0, -1], 3.0, 100.0, lc, uc)[0] == lc
            assert O.prim_hit_interval(capi.GEOM_TRIANGLE, tri16, [0, 0.5, 0], [0, 0, -1], -1.0, 3.0, lc, uc)[0] == uc
