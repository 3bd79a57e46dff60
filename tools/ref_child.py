"""Child for profiler runs: the reference binary's configuration (2048x1024, 121 spp, depth 8) a few times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import capi, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H = 2048, 1024
flat, _ = scenes.default_scene()
ctx = capi.Context(0); ctx.upload(flat)
cam = capi.default_viewport(W / H)
for _ in range(n):
    g = ctx.render(cam, W, H, spp=121, max_depth=8)
print("kernel_ms", g["stats"]["kernel_ms"])
