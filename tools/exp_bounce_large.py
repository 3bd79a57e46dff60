"""Kernel time of bounce chains in the large synthetic scenes (depths 0, 1, 8; 1 spp).  usage: python tools/exp_bounce_large.py [C3|C4s|C5] [W H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import capi, scenes
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
spec = {"C3": lambda: scenes.sphere_field(10000), "C4s": lambda: scenes.triangle_terrain(100, 50),
        "C4m": lambda: scenes.triangle_terrain(300, 150), "C5": scenes.light_stress}[cfg]()
ctx = capi.Context(0); ctx.upload(spec.flatten())
cam = spec.cam12(W / H)
for depth in (0, 1, 8):
    g = ctx.render(cam, W, H, max_depth=depth)
    g = ctx.render(cam, W, H, max_depth=depth)
    st = g["stats"]
    print("%s %dx%d depth %d (%s): kernel %.2f ms, %d primary + %d shadow rays" %
          (cfg, W, H, depth, os.environ.get("MRT_CULL", "default"), st["kernel_ms"], st["primary_rays"], st["shadow_rays"]), flush=True)
