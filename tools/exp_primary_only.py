"""Experiment: how much of a config's kernel time is primary visibility?  The same scene with and without its lights
(no light: no shadow rays, no shading beyond the constant term).  usage: python tools/exp_primary_only.py C3 [C5 ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import capi, scenes, multigpu

for cfg in sys.argv[1:]:
    spec = {"C3": lambda: scenes.sphere_field(10000), "C5": scenes.light_stress, "C4": lambda: scenes.triangle_terrain(1000, 500)}[cfg]()
    W, H = (1920, 1080) if cfg == "C5" else (3840, 2160)
    for lights in ("all", "none"):
        s = spec if lights == "all" else scenes.SceneSpec(spec.name, spec.spheres, spec.sphere_obj, spec.tri_verts, spec.tri_obj,
                                                          spec.mat_kind, spec.mat_rgb, [], spec.camera)
        fr = multigpu.FrameRenderer(s.flatten(), s.cam12(W / H), W, H)
        for _ in range(30): fr.render_local()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fr.render_local()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        print(cfg, "lights", lights, "%.1f us" % best, flush=True)
        fr.close()
