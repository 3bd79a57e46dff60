"""Stress run: many random scenes, every culling kernel against the brute-force kernel, bit for bit.
usage: python tools/stress_parity.py <first_seed> <count> [small|large|both|huge] [shift [scale [wild]]]
huge: 200 to 4000 primitives per type (dozens of clusters: hierarchy walks, depth bounds, teams dealing leaves).
wild (any sixth argument): per seed, also one of: extreme field of view, camera very far or very near, radii over
eight decades, degenerate triangles (repeated or collinear vertices), a NaN or an infinity in the geometry.
shift / scale: the whole scene (geometry, point lights, camera) is scaled and then moved by shift * (0.6, -0.4, 1)
-- far from the world origin fp32 positions get coarse, which is what the culling margins have to survive.
Environment: STRESS_LISTS=1 also renders every large / huge scene with each source of primary candidates (MRT_OPT_LISTS 1, 2,
3) and with lists shortened to a few entries (MRT_OPT_LIST_CAP: the overflow paths); MRT_CULL_TEAM=4 renders every large scene with teams of four waves per tile (mrt_cull.hip) dealing leaves,
MRT_CULL_TEAM=5 with teams dealing lights."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from raytracing_amd import capi
import test_gpu_parity as T
from stress_common import moved, make_wild

first, count = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "both"
shift = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
wild = len(sys.argv) > 6


ctx = capi.Context(0)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    for kind in ("small", "large", "huge"):
        if which not in (kind, "both") or (kind == "huge" and which != "huge"):
            continue
        spec = T._random_small_scene(seed) if kind == "small" else (T._random_scene(seed) if kind == "large" else
                                                                      T._random_scene(seed, 200, 4000))
        if wild:
            spec = make_wild(spec, seed)
        spec = moved(spec, shift, scale)
        rng = np.random.default_rng(seed)
        W, H = int(rng.integers(3, 40)) * 8, int(rng.integers(2, 30)) * 8 + int(rng.integers(0, 8))
        if seed % 4 == 3:   # ragged widths, tiny frames
            W, H = int(rng.integers(1, 330)), int(rng.integers(1, 250))
        part = (0, 1, 8)
        if seed % 7 == 5:   # one rank's strips of a partition
            nr = int(rng.integers(2, 9)); part = (int(rng.integers(0, nr)), nr, 8 * int(rng.integers(1, 4)))
        spp = 121 if seed % 23 == 0 else 1
        if spp == 121:
            W, H = 64, 40
        ctx.upload(spec.flatten())
        cam = spec.cam12(W / H)
        ctx.set_cull(0); ref = ctx.render(cam, W, H, spp=spp, rank=part[0], num_ranks=part[1], strip_rows=part[2])
        if os.environ.get("STRESS_ORACLE"):   # also the brute-force kernel against the CPU oracle
            import _oracle as O
            r = O.Scene.from_spec(spec).render(cam, W, H, spp=spp, nthreads=16)
            rows = capi.local_row_index(H, *part)
            r = {k: r[k][rows] for k in ("id", "t", "rgb")}
            ok = all(np.array_equal(ref[k], r[k], equal_nan=True) for k in ("id", "t", "rgb"))
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, kind, "brute force vs ORACLE", W, H, spp,
                      {k: int((~((ref[k] == r[k]) | (np.isnan(ref[k]) & np.isnan(r[k])))).sum()) for k in ("id", "t", "rgb")}, flush=True)
        # mode 5 three times over: from the second repeat of a view the small-scene launcher starts the frame at the row
        # where the geometry begins (mrt_small.hip, small_first_row), and the order must not show
        runs = [(m, 0, 0) for m in ((5, 5, 5, 2) if kind == "small" else (2,))]
        if kind != "small" and os.environ.get("STRESS_LISTS"):
            cap = int(rng.integers(1, 9))
            runs += [(2, 1, 0), (2, 2, 0), (2, 3, 0), (2, 2, cap), (2, 3, cap)]
        for mode, lists, cap in runs:
            ctx.set_lists(lists, cap)
            ctx.set_cull(mode); g = ctx.render(cam, W, H, spp=spp, rank=part[0], num_ranks=part[1], strip_rows=part[2])
            ctx.set_lists(0, 0)
            ok = all(np.array_equal(g[k], ref[k], equal_nan=True) for k in ("id", "t", "rgb")) and \
                 all(g["stats"][k] == ref["stats"][k] for k in ("primary_rays", "shadow_rays", "hit_pixels", "occluded"))
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, kind, "mode", mode, "lists", lists, cap, W, H, spp,
                      {k: int((g[k] != ref[k]).sum()) for k in ("id", "t", "rgb")}, flush=True)
    if (seed - first) % 50 == 49:
        print("seed", seed, "done, mismatches so far", bad, "%.0fs" % (time.time() - t0), flush=True)
print("stress done:", count, "seeds,", bad, "mismatches", flush=True)
sys.exit(1 if bad else 0)
