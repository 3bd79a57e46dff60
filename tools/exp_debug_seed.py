"""Debug helper: one stress seed, brute force vs the culling kernel; prints what differs (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from raytracing_amd import capi
import test_gpu_parity as T
from stress_common import moved
seed, shift = int(sys.argv[1]), float(sys.argv[2])
spec = moved(T._random_scene(seed, 200, 4000), shift, 1.0)
rng = np.random.default_rng(seed)
W, H = int(rng.integers(3, 40)) * 8, int(rng.integers(2, 30)) * 8 + int(rng.integers(0, 8))
if seed % 4 == 3: W, H = int(rng.integers(1, 330)), int(rng.integers(1, 250))
ctx = capi.Context(0); flat = spec.flatten(); ctx.upload(flat); cam = spec.cam12(W / H)
ctx.set_cull(0); ref = ctx.render(cam, W, H)
ctx.set_cull(2); g = ctx.render(cam, W, H)
bad = np.argwhere(g["id"] != ref["id"])
print("frame", W, H, "spheres", flat.num_spheres, "triangles", flat.num_triangles, "mismatching pixels", len(bad))
sph = set(int(x) for x in flat.sphere_obj)
for (r, c) in bad[:12]:
    a, b = int(ref["id"][r, c]), int(g["id"][r, c])
    print((int(r), int(c)), "brute id", a, "sphere" if a in sph else "tri", "t", ref["t"][r, c], "| cull id", b, "sphere" if b in sph else ("tri" if b >= 0 else "-"), "t", g["t"][r, c])
