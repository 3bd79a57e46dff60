"""Diagnostic: culling statistics (exact tests, opened clusters) per config for the one-level kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import capi, scenes
cfgs = sys.argv[1].split(",")
ctx = capi.Context(0); ctx.set_cull(int(sys.argv[2]) if len(sys.argv) > 2 else 2)
for cfg in cfgs:
    if cfg == "C2":
        flat, cam = scenes.default_scene(); W, H = 1920, 1080
        cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
    else:
        spec = {"C3": lambda: scenes.sphere_field(10000), "C5": scenes.light_stress,
                "C4m": lambda: scenes.triangle_terrain(300, 150), "C4": lambda: scenes.triangle_terrain(1000, 500)}[cfg]()
        W, H = (1920, 1080) if cfg == "C5" else (3840, 2160)
        flat = spec.flatten(); cam12 = spec.cam12(W / H)
    ctx.upload(flat)
    r = ctx.render(cam12, W, H)
    st = r["stats"]; tiles = ((W + 7) // 8) * ((H + 7) // 8)
    print(cfg, st, "per tile: exact %.1f clusters %.1f" % (st["exact_tests"] / tiles, st["clusters_opened"] / tiles), flush=True)
