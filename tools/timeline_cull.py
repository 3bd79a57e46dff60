#!/usr/bin/env python3
"""Per-wave timeline of the tile-culling kernel on a bench config (diagnostic build only).

    make -C raytracing_amd/csrc variant NAME=timeline DEFS=-DMRT_TIMELINE
    MRT_LIB=raytracing_amd/lib/variants/libmrt_timeline.so python3 tools/timeline_cull.py C5 [out.json]

Prints where a frame's time goes: waves resident over time, wave life by tile row, the tail behind the last start."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raytracing_amd import capi, multigpu, scenes

    cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
    kw = {}
    if cfg == "REF":  # the bounce kernel on the reference's own frame (stamps: mrt_bounce.hip)
        flat, cam = scenes.default_scene(); W, H = 2048, 1024
        cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
        kw = {"spp": 121, "max_depth": 8}
    spec, (W, H) = (None, (W, H)) if cfg == "REF" else {"C3": (scenes.sphere_field(10000), (3840, 2160)), "C4": (scenes.triangle_terrain(1000, 500), (3840, 2160)),
                    "C5": (scenes.light_stress(), (1920, 1080))}[cfg]
    fr = multigpu.FrameRenderer(flat, cam12, W, H, **kw) if cfg == "REF" else multigpu.FrameRenderer(spec.flatten(), spec.cam12(W / H), W, H)
    for _ in range(3):
        fr.render_local()
    torch.cuda.synchronize()
    L = capi.libmrt()
    tx, ty = (W + 7) // 8, (H + 7) // 8
    nw = tx * ty
    words = 8 if cfg == "REF" else 4
    buf = np.zeros(nw * words, dtype=np.uint64)
    diag = L.mrt_diag_timeline_bounce if cfg == "REF" else L.mrt_diag_timeline_cull
    diag.argtypes = [C.c_void_p, C.c_size_t]
    assert diag(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(nw, words).astype(np.int64)
    rt0 = t[:, 0].min()
    start, end = (t[:, 0] - rt0) * 10.0 / 1e3, (t[:, 1] - rt0) * 10.0 / 1e3   # us
    life_us = end - start
    cycles = t[:, 2]
    exact, clusters = t[:, 3] & 0xFFFFFFFF, t[:, 3] >> 32
    span = end.max()
    pct = lambda a: [round(float(x), 2) for x in np.percentile(a, [0, 10, 50, 90, 99, 100])]
    edges = np.linspace(0, span, 25)
    prof = [[round(float(e), 1), int(((start <= e) & (end > e)).sum()), int((start <= e).sum())] for e in edges[1:-1]]
    rows = life_us.reshape(ty, tx)
    out = {"config": cfg, "waves": nw, "frame_span_us": round(float(span), 1),
           "start_us pctl[0,10,50,90,99,100]": pct(start), "end_us pctl": pct(end), "life_us pctl": pct(life_us),
           "cycles pctl": pct(cycles), "exact tests per wave pctl": pct(exact), "clusters per wave pctl": pct(clusters),
           "sum_life_us / 7168 slots": round(float(life_us.sum() / 7168), 1),
           "waves with life > 10% of span": int((life_us > 0.1 * span).sum()),
           "longest waves (us, tile_x, tile_y, exact, clusters)": [[round(float(life_us[i]), 1), int(i % tx), int(i // tx), int(exact[i]), int(clusters[i])]
                                                                   for i in np.argsort(-life_us)[:12]],
           "profile[t_us, resident waves, started so far]": prof,
           "mean life_us by tile row (every %d-th)" % max(1, ty // 16): [round(float(x), 1) for x in rows.mean(axis=1)[::max(1, ty // 16)]],
           "max life_us by tile row": [round(float(x), 1) for x in rows.max(axis=1)[::max(1, ty // 16)]]}
    if cfg == "REF":  # where the cycles of the longest waves go (mrt_bounce.hip, TLB_MARK)
        top = np.argsort(-life_us)[:64]
        ph = t[top, 4:8].sum(axis=0).astype(float)
        out["longest 64 waves: share of loop cycles in closest hit | light | push + bounce | fold + loop head"] = [round(float(x), 3) for x in ph / ph.sum()]
        out["longest 64 waves: cycles per iteration"] = round(float(t[top, 4:8].sum() / (t[top, 3] >> 32).sum()), 0)
        allp = t[:, 4:8].sum(axis=0).astype(float)
        out["all waves: share of loop cycles, same four"] = [round(float(x), 3) for x in allp / max(allp.sum(), 1.0)]
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    fr.close()


if __name__ == "__main__":
    main()
