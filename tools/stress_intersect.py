"""Stress run for mrt_intersect: random primitives (including slivers, huge / tiny radii, non-finite values) and
random rays (including zero, axis-parallel and non-finite ones) against the oracle's single-primitive tests, bit
for bit.  usage: python tools/stress_intersect.py <first_seed> <count>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_amd import capi
import _oracle as O

first, count = int(sys.argv[1]), int(sys.argv[2])
ctx = capi.Context(0)
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(31000 + seed)
    n = 200
    o = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-2, 3)
    d = rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-3, 3)
    d[::17] = 0.0
    d[1::13, int(rng.integers(0, 3))] = 0.0
    if seed % 9 == 0:
        o[int(rng.integers(0, n)), int(rng.integers(0, 3))] = rng.choice([np.nan, np.inf, -np.inf])
        d[int(rng.integers(0, n)), int(rng.integers(0, 3))] = rng.choice([np.nan, np.inf, -np.inf])
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    lo, hi = [(-1.0, 100.0), (0.01, np.inf), (1e-3, 1.0), (0.0, 0.0), (5.0, 1.0)][seed % 5]
    if seed % 2:
        c = (rng.normal(size=3) * 5).astype(np.float32); r = np.float32(10.0 ** rng.uniform(-4, 4))
        if seed % 31 == 1: r = np.float32(rng.choice([0.0, np.nan, np.inf, -1.0]))
        hit, t, p, nrm = ctx.intersect(capi.GEOM_SPHERE, list(c) + [r], rays, lo, hi)
        ref = [O.sphere_hit(c, r, rays[i, :3], rays[i, 3:], lo, hi) for i in range(n)]
    else:
        a = rng.normal(size=3) * 5
        tri = np.stack([a, a + rng.normal(size=3) * 10.0 ** rng.uniform(-3, 2), a + rng.normal(size=3) * 10.0 ** rng.uniform(-3, 2)])
        if seed % 6 == 0: tri[2] = tri[1] + 1e-5 * rng.normal(size=3)
        if seed % 30 == 0: tri[2] = tri[1]
        tri16 = capi.triangle_precompute(np.float32([tri]))[:, 0]
        hit, t, p, nrm = ctx.intersect(capi.GEOM_TRIANGLE, tri16, rays, lo, hi)
        ref = [O.triangle16_hit(tri16, rays[i, :3], rays[i, 3:], lo, hi) for i in range(n)]
    for i, (ok, ot, op, on) in enumerate(ref):
        same = bool(hit[i]) == bool(ok) and (not ok or (np.array_equal(t[i], np.float32(ot), equal_nan=True) and
                                                        np.array_equal(p[i], op, equal_nan=True) and np.array_equal(nrm[i], on, equal_nan=True)))
        if not same:
            bad += 1
            print("MISMATCH seed", seed, "ray", i, "sphere" if seed % 2 else "triangle", hit[i], ok, t[i], ot, flush=True)
            break
print("intersect stress done:", count, "seeds,", bad, "mismatches", flush=True)
sys.exit(1 if bad else 0)
