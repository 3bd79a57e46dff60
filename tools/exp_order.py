"""Experiment: launch orders of the tile rows of the C2 frame (variant build with -DMRT_SMALL_ORDER_TABLE).
    make -C raytracing_amd/csrc variant NAME=order DEFS=-DMRT_SMALL_ORDER_TABLE
    MRT_LIB=raytracing_amd/lib/variants/libmrt_order.so python3 tools/exp_order.py"""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from raytracing_amd import capi, scenes, multigpu
flat, cam = scenes.default_scene(); W, H = 1920, 1080
cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
fr = multigpu.FrameRenderer(flat, cam12, W, H)
L = capi.libmrt()
L.mrt_diag_set_order.argtypes = [C.c_void_p, C.c_size_t]
n = 135

def timed(order, reps=3, k=400):
    o = np.ascontiguousarray(order, dtype=np.uint16)
    assert sorted(o.tolist()) == list(range(n))
    assert L.mrt_diag_set_order(o.ctypes.data, n) == 0
    best = 1e9
    for _ in range(reps):
        for _ in range(40): fr.render_local()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(k): fr.render_local()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / k * 1e3)
    return round(best, 2)

def rot(r0): return [(y + r0) % n for y in range(n)]
def weave(heavy, sky, lead=0, every=1, tail=0):
    """heavy rows in order with one sky row after every `every` heavy rows, `lead` sky rows first, `tail` kept for the end"""
    sky = list(sky); out = sky[:lead]; sky = sky[lead:]
    keep = sky[len(sky) - tail:] if tail else []; sky = sky[:len(sky) - tail] if tail else sky
    for i, h in enumerate(heavy):
        out.append(h)
        if (i + 1) % every == 0 and sky: out.append(sky.pop(0))
    return out + sky + keep

for _ in range(200): fr.render_local()
torch.cuda.synchronize()
res = {}
res["identity"] = timed(list(range(n)))
res["rot24"] = timed(rot(24))
heavy = list(range(26, 100)); sky = list(range(100, 135)) + list(range(0, 26))
res["band then sky (=rot26)"] = timed(heavy + sky)
for every in (1, 2, 3):
    for tail in (0, 10, 20, 30):
        res[f"weave every {every}, tail {tail}"] = timed(weave(heavy, sky, 0, every, tail))
res["weave every 1, lead 2, tail 10"] = timed(weave(heavy, sky, 2, 1, 10))
# densest rows first: rows 45..99 (70 % candidates), then 26..44, then sky
res["dense first"] = timed(list(range(45, 100)) + list(range(26, 45)) + sky)
res["dense first, reversed band"] = timed(list(range(99, 25, -1)) + sky)
res["identity again"] = timed(list(range(n)))
for k, v in res.items(): print(f"{v:7.2f}  {k}")
