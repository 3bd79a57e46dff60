"""Diagnostic: exact tests and opened clusters of a config with and without its lights (primary / shadow split).
usage: python3 tools/exp_c4_split.py C3|C4|C5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import capi, scenes
cfg = sys.argv[1]
spec = {"C3": lambda: scenes.sphere_field(10000), "C5": scenes.light_stress, "C4": lambda: scenes.triangle_terrain(1000, 500)}[cfg]()
W, H = (1920, 1080) if cfg == "C5" else (3840, 2160)
tiles = ((W + 7) // 8) * ((H + 7) // 8)
ctx = capi.Context(0)
for lights in ("all", "none"):
    s = spec if lights == "all" else scenes.SceneSpec(spec.name, spec.spheres, spec.sphere_obj, spec.tri_verts, spec.tri_obj,
                                                      spec.mat_kind, spec.mat_rgb, [], spec.camera)
    ctx.upload(s.flatten())
    r = ctx.render(s.cam12(W / H), W, H)
    st = r["stats"]
    print(cfg, "lights", lights, st, "per tile: exact %.1f clusters %.1f" % (st["exact_tests"] / tiles, st["clusters_opened"] / tiles), flush=True)
