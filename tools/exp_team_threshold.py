"""Diagnostic: terrains of several sizes under one wave per tile and under teams dealing leaves (which have the pair lists).
usage: python3 tools/exp_team_threshold.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import capi, scenes, multigpu
which = sys.argv[1] if len(sys.argv) > 1 else "terrains"
specs = [scenes.triangle_terrain(nx, nz) for (nx, nz) in ((50, 25), (100, 50), (200, 100), (300, 150), (500, 250))] if which == "terrains" else \
        [scenes.triangle_terrain(nx, nz) for (nx, nz) in ((350, 175), (600, 300), (700, 350))] + [scenes.sphere_field(n) for n in (100000, 300000)]
for spec in specs:
    for (W, H) in ((1920, 1080), (3840, 2160)):
        out = []
        for team in (1, 4, 0):
            fr = multigpu.FrameRenderer(spec.flatten(), spec.cam12(W / H), W, H)
            fr.ctx.set_cull_team(team)
            for _ in range(5):
                fr.render_local()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                fr.render_local()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / 30 * 1e3)
            fr.close()
        print(f"{spec.name} {W}x{H}: one wave {out[0]:.3f} ms, teams {out[1]:.3f} ms, by the scene {out[2]:.3f} ms", flush=True)
