import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_amd import capi
import test_gpu_parity as T
import _oracle as O
seed = int(sys.argv[1]); W, H = int(sys.argv[2]), int(sys.argv[3])
spec = T._random_small_scene(seed)
flat = spec.flatten()
print("ns", flat.num_spheres, "nt", flat.num_triangles, "nl", flat.num_lights, spec.lights)
ctx = capi.Context(0); ctx.upload(flat)
cam = spec.cam12(W / H)
ora = O.Scene.from_spec(spec).render(cam, W, H, nthreads=8)
for mode in (0, 5, 2, 3):
    ctx.set_cull(mode); g = ctx.render(cam, W, H)
    d = {k: int((g[k] != ora[k]).sum()) for k in ("id", "t", "rgb")}
    print("mode", mode, "vs oracle", d, {k: g["stats"][k] for k in ("shadow_rays", "occluded", "hit_pixels")})
    if d["id"]:
        ys, xs = np.nonzero(g["id"] != ora["id"])
        print("   id diff rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "gpu ids", np.unique(g["id"][ys, xs])[:10], "ora ids", np.unique(ora["id"][ys, xs])[:10])
    elif d["rgb"]:
        ys, xs = np.nonzero((g["rgb"] != ora["rgb"]).any(-1))
        print("   rgb diff px", list(zip(ys[:8], xs[:8])), "ids", ora["id"][ys[:8], xs[:8]])
print("oracle stats", ora["stats"])
