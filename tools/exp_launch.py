"""Experiment: host-side launch cost of one frame (plain launches vs events vs HIP graph)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import capi, scenes, multigpu

flat, cam = scenes.default_scene()
W, H = 1920, 1080
cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
fr = multigpu.FrameRenderer(flat, cam12, W, H)
for _ in range(50):
    fr.render_local()
torch.cuda.synchronize()

def timeit(fn, n=1000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    return t_issue / n * 1e6, t1 / n * 1e6

print("plain launches: issue %.2f us, total %.2f us per frame" % timeit(fr.render_local))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def with_ev():
    e0.record(); fr.render_local(); e1.record()
print("with 2 events : issue %.2f us, total %.2f us per frame" % timeit(with_ev))
torch.cuda.synchronize(); print("event-measured kernel: %.2f us" % (e0.elapsed_time(e1) * 1e3))

# region events
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(1000): fr.render_local()
e.record(); torch.cuda.synchronize()
print("region events: %.2f us per frame" % (s.elapsed_time(e)))

# HIP graph of G frames
for G in (1, 10, 50):
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fr.render_local()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(G): fr.render_local()
    torch.cuda.synchronize()
    n = 1000 // G
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    print("graph of %d frames: %.2f us per frame" % (G, (time.perf_counter() - t0) / (n * G) * 1e6))
fr.close()
