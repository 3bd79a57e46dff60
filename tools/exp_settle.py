"""Experiment: how the C2 frame time settles over successive groups of launches (clock ramp after idle)."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import capi, scenes, multigpu
flat, cam = scenes.default_scene(); W, H = 1920, 1080
cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
fr = multigpu.FrameRenderer(flat, cam12, W, H)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for _ in range(5): fr.render_local()
torch.cuda.synchronize()
out = []
t00 = time.perf_counter()
for rep in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fr.render_local()
    e1.record(); torch.cuda.synchronize()
    out.append((round((time.perf_counter() - t00) * 1e3, 1), round(e0.elapsed_time(e1) / n * 1e3, 2)))
print(out)
