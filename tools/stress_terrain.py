"""Stress run for the leaf refinement's margins (mrt_cull.hip): height-field terrains seen from low, grazing cameras --
thousands of sub-pixel triangles per tile along the horizon, where t = num / (n.d) is least accurate -- culling kernel
against the brute-force kernel, bit for bit (id, t, rgb).
usage: python tools/stress_terrain.py <first_seed> <count> [shift [scale [hard]]]
hard (any fifth argument): grids up to 600 x 400 cells, cameras down to a thousandth of a unit above the crests, fields of
view down to one degree.
Per seed: grid size, amplitude, camera height and pitch, field of view, frame size and team mode are drawn at random;
shift / scale move the whole scene away from the world origin (coarser fp32 positions)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from raytracing_amd import capi, scenes
from stress_common import moved

first, count = int(sys.argv[1]), int(sys.argv[2])
shift = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
hard = len(sys.argv) > 5
ctx = capi.Context(0)
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    nx, nz = (int(rng.integers(40, 600)), int(rng.integers(40, 400))) if hard else (int(rng.integers(40, 260)), int(rng.integers(40, 200)))
    amp = float(10.0 ** rng.uniform(-2, 0.5))
    tv = scenes.terrain_triangles(nx, nz, amp=amp)
    n = tv.shape[0]
    kind, rgb = scenes._cycle_materials(n)
    height = float(-2.0 + amp + 10.0 ** rng.uniform(-3 if hard else -2, 1.2))          # just above the crests ... 16 units up
    look_from = (float(rng.uniform(-30, 30)), height, float(rng.uniform(-20, 5)))
    look_at = (float(rng.uniform(-40, 40)), float(-2.0 + rng.uniform(-1, 3)), float(rng.uniform(-150, -60)))
    cam = {"look_from": look_from, "look_at": look_at, "up": (0.02, 1.0, 0.01), "vfov": float(10.0 ** rng.uniform(0, 1.8)) if hard else float(rng.uniform(8, 70))}
    lights = [(capi.LIGHT_DIRECTIONAL, tuple(rng.normal(size=3) * (1, 0.3, 1) - (0, 0.6, 0)), (3.0, 2.5, 2.0)),
              (capi.LIGHT_POINT, (float(rng.uniform(-30, 30)), float(rng.uniform(0, 20)), float(rng.uniform(-120, -20))), (0.6, 0.8, 1.0))]
    spec = scenes.SceneSpec(f"terrain{seed}", np.zeros((0, 4), np.float32), [], tv, np.arange(n), kind, rgb, lights, cam)
    spec = moved(spec, shift, scale)
    W, H = int(rng.integers(20, 90)) * 8, int(rng.integers(10, 50)) * 8 + int(rng.integers(0, 8))
    cam12 = spec.cam12(W / H)
    ctx.upload(spec.flatten())
    ctx.set_cull(0)
    ref = ctx.render(cam12, W, H)
    ctx.set_cull(2)
    for team in (1, 4):
        ctx.set_cull_team(team)
        g = ctx.render(cam12, W, H)
        same = all(np.array_equal(g[k].view(np.int32), ref[k].view(np.int32)) for k in ("id", "t", "rgb"))
        if not same or g["stats"]["shadow_rays"] != ref["stats"]["shadow_rays"] or g["stats"]["occluded"] != ref["stats"]["occluded"]:
            bad += 1
            d = int((g["id"] != ref["id"]).sum())
            print(f"MISMATCH seed {seed} team {team} {W}x{H} {n} triangles amp {amp:.3g} height {height:.3g}: {d} ids differ", flush=True)
    ctx.set_cull_team(0)
    if (seed - first) % 10 == 9:
        print(f"  ... seed {seed}, {time.time() - t0:.0f}s, exact tests / pixel {g['stats']['exact_tests'] / (W * H):.1f}", flush=True)
print(f"terrain stress done: {count} seeds, {bad} mismatches")
sys.exit(1 if bad else 0)
