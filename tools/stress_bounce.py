"""Stress run for the bounce kernel: random scenes (mixed diffuse / metallic, point and directional lights) at
random depths, GPU against the oracle, bit for bit.
usage: python tools/stress_bounce.py <first_seed> <count> [--shift X] [--wild]
--shift X: the scene moved by X * (0.6, -0.4, 1); --wild: stress_common.make_wild per seed.  Keywords, so that a
misplaced word cannot silently become a shift (round 2 lost a run to `... 0 wild`)."""
import argparse
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from raytracing_amd import capi
import test_gpu_parity as T
import _oracle as O
from stress_common import moved, make_wild

_ap = argparse.ArgumentParser()
_ap.add_argument("first", type=int); _ap.add_argument("count", type=int)
_ap.add_argument("--shift", type=float, default=0.0); _ap.add_argument("--wild", action="store_true")
_a = _ap.parse_args()
first, count, shift, wild = _a.first, _a.count, _a.shift, _a.wild
ctx = capi.Context(0)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    small = seed % 3 != 0
    spec = T._random_small_scene(seed) if small else T._random_scene(seed)
    if seed % 5 == 0:
        spec.mat_kind[:] = capi.MAT_METALLIC
    if wild:
        spec = make_wild(spec, seed)
    spec = moved(spec, shift)
    W, H = int(rng.integers(3, 24)) * 8 + int(rng.integers(0, 8)), int(rng.integers(2, 16)) * 8 + int(rng.integers(0, 8))
    depth = int(rng.integers(1, 9))
    spp = 121 if seed % 11 == 0 else 1
    if spp == 121:
        W, H = 40, 24
    part = (0, 1, 8)
    if seed % 7 == 5:   # one rank's strips of a partition: a pixel's random stream is its index in the FULL frame
        nr = int(rng.integers(2, 9)); part = (int(rng.integers(0, nr)), nr, 8 * int(rng.integers(1, 4)))
    ctx.upload(spec.flatten())
    cam = spec.cam12(W / H)
    ctx.set_cull(0 if seed % 2 else 1)   # small scenes: with / without the small-scene culling data
    g = ctx.render(cam, W, H, spp=spp, max_depth=depth, rank=part[0], num_ranks=part[1], strip_rows=part[2])
    r = O.Scene.from_spec(spec).render(cam, W, H, spp=spp, max_depth=depth, nthreads=16)
    rows = capi.local_row_index(H, *part)
    r = {k: (r[k][rows] if k != "stats" else r[k]) for k in r}
    ok = all(np.array_equal(g[k], r[k], equal_nan=True) for k in ("id", "t", "rgb")) and \
         (part[1] > 1 or all(g["stats"][k] == r["stats"][k] for k in ("primary_rays", "shadow_rays", "hit_pixels", "occluded")))
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "small" if small else "large", W, H, "spp", spp, "depth", depth,
              {k: int((g[k] != r[k]).sum()) for k in ("id", "t", "rgb")}, flush=True)
    if (seed - first) % 50 == 49:
        print("seed", seed, "done, mismatches so far", bad, "%.0fs" % (time.time() - t0), flush=True)
print("bounce stress done:", count, "seeds,", bad, "mismatches", flush=True)
sys.exit(1 if bad else 0)
