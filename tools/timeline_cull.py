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
    words = 16 if cfg == "REF" else 4
    pipe = cfg == "REF" and os.environ.get("MRT_BOUNCE_PIPE", "1") != "0"  # the two-wave pipeline's stamps (mrt_bounce_pipe.hip)
    buf = np.zeros(nw * words, dtype=np.uint64)
    diag = (L.mrt_diag_timeline_pipe if pipe else L.mrt_diag_timeline_bounce) if cfg == "REF" else L.mrt_diag_timeline_cull
    diag.argtypes = [C.c_void_p, C.c_size_t]
    assert diag(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(nw, words).astype(np.int64)
    if pipe:  # tiles without candidates leave no stamps there: give them the frame's first stamp and no life
        seen = t[:, 0] > 0
        t[~seen, 0] = t[seen, 0].min(); t[~seen, 1] = t[seen, 0].min()
    rt0 = t[:, 0].min()
    start, end = (t[:, 0] - rt0) * 10.0 / 1e3, (t[:, 1] - rt0) * 10.0 / 1e3   # us
    life_us = end - start
    weight = (t[:, 2] >> 56) & 0xFF
    cycles = t[:, 2] & ((1 << 56) - 1)
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
    if pipe:
        iters = (t[:, 3] >> 32)
        top = np.argsort(-life_us)[:64]
        sh = t[top, 4:9].sum(axis=0).astype(float)
        names = ["wait for the tracer + read the record", "sky", "shadow rays + light", "push", "fold"]
        out["longest 64 tiles: shader cycles per iteration"] = {n: round(float(sh[i] / iters[top].sum()), 0) for i, n in enumerate(names)}
        out["longest 64 tiles: tracer cycles per iteration (all of it)"] = round(float(cycles[top].sum() / iters[top].sum()), 0)
        byit = np.argsort(-iters)[:24]
        out["most iterations (iterations, life_us, start_us, tile_x, tile_y)"] = [[int(iters[i]), round(float(life_us[i]), 1), round(float(start[i]), 1), int(i % tx), int(i // tx)] for i in byit]
    elif cfg == "REF":  # where the cycles of the longest waves go (mrt_bounce.hip, TLB_MARK)
        top = np.argsort(-life_us)[:64]
        ph = t[top, 4:8].sum(axis=0).astype(float)  # (with the finer stamps in, these four no longer mean what they say)
        out["longest 64 waves: share of loop cycles in closest hit | light | push + bounce | fold + loop head"] = [round(float(x), 3) for x in ph / ph.sum()]
        out["longest 64 waves: cycles per iteration"] = round(float(t[top, 4:8].sum() / (t[top, 3] >> 32).sum()), 0)
        iters = (t[:, 3] >> 32)
        byit = np.argsort(-iters)[:48]
        out["most iterations (iterations, life_us, start_us, tile_x, tile_y)"] = [[int(iters[i]), round(float(life_us[i]), 1), round(float(start[i]), 1), int(i % tx), int(i // tx)] for i in byit]
        out["iterations histogram (edges 0,100,..,1200)"] = [int(x) for x in np.histogram(iters, bins=np.arange(0, 1301, 100))[0]]
        out["us per iteration by iterations decile of the tiles with > 300"] = [round(float(x), 2) for x in np.percentile((life_us / np.maximum(iters, 1))[iters > 300], [0, 25, 50, 75, 100])] if (iters > 300).any() else []
        out["iterations by predicted weight: [weight, tiles, min, median, max]"] = [
            [int(w), int((weight == w).sum())] + [int(x) for x in np.percentile(iters[weight == w], [0, 50, 100])] for w in np.unique(weight)]
        fine = t[top, 4:16].sum(axis=0).astype(float)
        names = ["sky+select (after merge)", "shading", "push+rng+bounce tail", "fold+loop head", "loop head -> before prim loop", "prim loop", "hit record", "light setup+bundle", "shadow cull+loops+merge", "push+rng"]
        out["longest 64 waves: cycles per iteration by stamp"] = {n: round(float(fine[i] / (t[top, 3] >> 32).sum()), 0) for i, n in enumerate(names)}
        allp = t[:, 4:8].sum(axis=0).astype(float)
        out["all waves: share of loop cycles, same four"] = [round(float(x), 3) for x in allp / max(allp.sum(), 1.0)]
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    fr.close()


if __name__ == "__main__":
    main()
