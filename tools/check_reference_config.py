"""The reference binary's own configuration -- default scene, 2048x1024, 121 samples per pixel, bounce chains to
depth 8, streams seeded with 1984 (src/main.cu:26, :165; multisample_mesh.cuh; curand_state.cuh:22) -- rendered on
the GPU, timed, and compared bit for bit with the oracle on row bands (the oracle renders any row range of the
full frame; a pixel's random stream is its full-frame index).
usage: python tools/check_reference_config.py [rows_per_band=8] [bands=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_amd import capi, scenes
import _oracle as O

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
bands = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W, H = 2048, 1024
flat, _ = scenes.default_scene()
ctx = capi.Context(0); ctx.upload(flat)
cam = capi.default_viewport(W / H)
g = ctx.render(cam, W, H, spp=121, max_depth=8)          # first call also builds the streams
t0 = time.time(); g = ctx.render(cam, W, H, spp=121, max_depth=8); wall = time.time() - t0
st = g["stats"]
rays = st["primary_rays"] + st["shadow_rays"]
print("GPU: kernel %.2f ms (wall %.1f ms): %d primary + %d shadow rays -> %.0f Mrays/s" %
      (st["kernel_ms"], wall * 1e3, st["primary_rays"], st["shadow_rays"], rays / st["kernel_ms"] / 1e3), flush=True)
if st.get("clusters_opened"):
    print("chain steps %d over %d lane slots: %.1f %% of the lanes busy; %.2f steps per primary ray" %
          (st["exact_tests"], st["clusters_opened"], 100.0 * st["exact_tests"] / st["clusters_opened"],
           st["exact_tests"] / st["primary_rays"]), flush=True)
ora = O.Scene.default()
bad = 0
for b in range(bands):
    r0 = (H // bands) * b + (H // bands) // 2
    t0 = time.time()
    r = ora.render(cam, W, H, rows=(r0, r0 + rows), spp=121, max_depth=8, nthreads=min(rows, 16))
    d = {k: int((g[k][r0:r0 + rows] != r[k][r0:r0 + rows]).sum()) for k in ("id", "t", "rgb")}
    bad += sum(d.values())
    print("rows %d..%d: oracle %.1f s, differing values %s" % (r0, r0 + rows, time.time() - t0, d), flush=True)
print("reference configuration:", "bit-identical on all bands" if not bad else "MISMATCH")
sys.exit(1 if bad else 0)
