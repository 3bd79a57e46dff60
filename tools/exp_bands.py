"""Diagnostic: kernel time per horizontal band of the frame (bands rendered as ranks of a strip partition)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from raytracing_amd import capi, scenes
cfg = sys.argv[1]; mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
spec = {"C4": lambda: scenes.triangle_terrain(1000, 500), "C3": lambda: scenes.sphere_field(10000)}[cfg]()
W, H = 3840, 2160
cam = spec.cam12(W / H)
ctx = capi.Context(0); ctx.upload(spec.flatten()); ctx.set_cull(mode)
strip = 80; nb = H // strip
buf = torch.zeros(strip * W * 3, device="cuda"); ids = torch.zeros(strip * W, dtype=torch.int32, device="cuda")
tt = torch.zeros(strip * W, device="cuda"); st = torch.zeros(8, dtype=torch.int64, device="cuda")
for b in range(nb):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.render_device(cam, W, H, buf.data_ptr(), ids.data_ptr(), tt.data_ptr(), st.data_ptr(), None, rank=b, num_ranks=nb, strip_rows=strip)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = st.cpu().numpy()
    print("band rows %4d-%4d  %8.3f ms  hits %7d exact/tile %8.1f clusters/tile %6.1f" % (b * strip, b * strip + strip, dt * 1e3, s[2], s[4] / (480 * 10), s[5] / 4800), flush=True)
