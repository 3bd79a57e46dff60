"""Diagnostic: the reference's native mode -- default scene, 2048x1024, 121 samples per pixel (MultisampleMesh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import capi, scenes
flat, cam = scenes.default_scene(); W, H = 2048, 1024
cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
ctx = capi.Context(0); ctx.upload(flat)
for mode in (int(a) for a in (sys.argv[1:] or ["1"])):
    ctx.set_cull(mode)
    ctx.render(cam12, W, H, spp=121)
    r = ctx.render(cam12, W, H, spp=121)
    st = r["stats"]; rays = st["primary_rays"] + st["shadow_rays"]
    print("mode", mode, st, "Mrays/s %.0f" % (rays / st["kernel_ms"] / 1e3), flush=True)
