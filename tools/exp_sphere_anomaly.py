"""Diagnostic: a sphere field or a terrain at 1080p and 4K under each source of primary candidates (MRT_OPT_LISTS), with counters.
usage: python3 tools/exp_sphere_anomaly.py <spheres> | 0 <nx> <nz>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import capi, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
spec = scenes.sphere_field(n) if n > 0 else scenes.triangle_terrain(int(sys.argv[2]), int(sys.argv[3]))  # (0 nx nz: a terrain)
ctx = capi.Context(0)
ctx.upload(spec.flatten())
for (W, H) in ((1920, 1080), (3840, 2160)):
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    for lists in (0, 1, 2, 3):
        ctx.set_lists(lists, 0)
        ctx.render(spec.cam12(W / H), W, H)
        r = ctx.render(spec.cam12(W / H), W, H)
        st = r["stats"]
        print(f"{W}x{H} lists {lists}: kernel {st['kernel_ms']:.3f} ms, exact / tile {st['exact_tests'] / tiles:.1f}, batches / tile {st['clusters_opened'] / tiles:.1f}", flush=True)
