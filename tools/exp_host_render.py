import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from raytracing_amd import capi, scenes
flat, _ = scenes.default_scene()
ctx = capi.Context(0); ctx.upload(flat)
for (W, H) in ((1920, 1080), (3840, 2160)):
    cam = capi.default_viewport(W / H)
    ctx.render(cam, W, H)
    t0 = time.perf_counter(); n = 10
    for _ in range(n): g = ctx.render(cam, W, H)
    dt = (time.perf_counter() - t0) / n
    mb = W * H * 20 / 1e6
    print("%dx%d: mrt_render wall %.2f ms per frame (%.1f MB D2H -> %.1f GB/s), kernel %.3f ms" % (W, H, dt * 1e3, mb, mb / dt / 1e3, g["stats"]["kernel_ms"]))
