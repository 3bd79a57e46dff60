"""Renders N frames of a config, with or without its lights (for counter passes: rocprofv3 --pmc ... -- python3 this).
usage: python3 tools/exp_render_n.py C3|C4|C5 all|none [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import capi, scenes, multigpu

cfg, lights = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
spec = {"C3": lambda: scenes.sphere_field(10000), "C5": scenes.light_stress, "C4": lambda: scenes.triangle_terrain(1000, 500)}[cfg]()
W, H = (1920, 1080) if cfg == "C5" else (3840, 2160)
if lights == "none":
    spec = scenes.SceneSpec(spec.name, spec.spheres, spec.sphere_obj, spec.tri_verts, spec.tri_obj, spec.mat_kind, spec.mat_rgb, [], spec.camera)
fr = multigpu.FrameRenderer(spec.flatten(), spec.cam12(W / H), W, H)
for _ in range(n):
    fr.render_local()
torch.cuda.synchronize()
fr.close()
