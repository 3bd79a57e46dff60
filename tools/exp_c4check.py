"""Diagnostic: full 1M-triangle terrain, GPU kernels vs CPU oracle at a small frame."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from raytracing_amd import capi, scenes
spec = scenes.triangle_terrain(1000, 500)
W, H = 192, 108
cam = spec.cam12(W / H)
ctx = capi.Context(0); ctx.upload(spec.flatten())
t0 = time.time(); r = O.Scene.from_spec(spec).render(cam, W, H, nthreads=16); print("oracle s", time.time() - t0, r["stats"])
for mode in (2, 4, 0):
    ctx.set_cull(mode); g = ctx.render(cam, W, H)
    print("mode", mode, g["stats"], "id equal", np.array_equal(g["id"], r["id"]), "t equal", np.array_equal(g["t"], r["t"]),
          "rgb equal", np.array_equal(g["rgb"], r["rgb"]), flush=True)
ctx.set_cull(2)
for (W, H) in ((1920, 1080), (3840, 2160)):
    g = ctx.render(spec.cam12(W / H), W, H)
    ids = g["id"]; print(W, H, "hit frac", (ids >= 0).mean(), "rows", np.where((ids >= 0).any(1))[0][[0, -1]], g["stats"])
