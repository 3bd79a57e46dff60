import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from raytracing_amd import capi, scenes
import test_gpu_parity as T
ctx = capi.Context(0)
flat, cam = scenes.default_scene()
for (W, H) in ((1920, 1080), (2048, 1024), (640, 480), (3840, 2160)):
    ctx.upload(flat)
    ctx.render(capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H), W, H)
for seed in range(12):
    spec = T._random_small_scene(seed)
    ctx.upload(spec.flatten()); ctx.render(spec.cam12(16 / 9), 1920, 1080)
