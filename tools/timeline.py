#!/usr/bin/env python3
"""Per-wave timeline of the small-scene kernel on the C2 frame (diagnostic build only).

    make -C raytracing_amd/csrc variant NAME=timeline DEFS=-DMRT_TIMELINE
    MRT_LIB=raytracing_amd/lib/variants/libmrt_timeline.so python3 tools/timeline.py [out.json]

The diagnostic kernel stamps s_memtime at wave entry, after the culling prologue, before the stores and at exit,
plus HW_ID / XCC_ID; this script renders a few frames, reads the stamps of the last one and prints where the frame's
time goes: launch ramp, per-phase wave time for sky-only and candidate tiles, waves per SIMD, tail."""
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

    flat, cam = scenes.default_scene()
    W, H = 1920, 1080
    cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
    fr = multigpu.FrameRenderer(flat, cam12, W, H)
    for _ in range(20):
        fr.render_local()
    torch.cuda.synchronize()
    fr.render_local()
    torch.cuda.synchronize()
    L = capi.libmrt()
    nw = (W // 16) * ((H + 7) // 8)
    buf = np.zeros(nw * 16, dtype=np.uint64)
    L.mrt_diag_timeline.argtypes = [C.c_void_p, C.c_size_t]
    assert L.mrt_diag_timeline(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(nw, 16).astype(np.int64)
    # durations from the shader clock (per-CU counters, not comparable across CUs); placement in time from the
    # 100 MHz real-time clock, which all CUs share
    pro, body, sto, life = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    rt0 = t[:, 4].min()
    start, end = (t[:, 4] - rt0) * 10.0, (t[:, 6] - rt0) * 10.0     # ns
    span = end.max()
    clk = np.median(life / np.maximum(end - start, 10.0))           # cycles per ns
    sky = t[:, 7] == 0
    hw = t[:, 5] & 0xFFFFFFFF
    xcc = (t[:, 5] >> 32) & 0xF
    simd_key = xcc * 100000 + ((hw >> 13) & 7) * 10000 + ((hw >> 12) & 1) * 1000 + ((hw >> 8) & 15) * 10 + ((hw >> 4) & 3)
    keys, inv, counts = np.unique(simd_key, return_inverse=True, return_counts=True)
    per_simd_end = np.zeros(len(keys)); np.maximum.at(per_simd_end, inv, end)
    per_simd_life = np.zeros(len(keys)); np.add.at(per_simd_life, inv, life)
    q = lambda a: [int(x) for x in np.percentile(a, [0, 10, 50, 90, 99, 100])] if len(a) else [0] * 6
    prof = []
    for k in range(24):
        mid = (k + 0.5) * span / 24
        prof.append([round(mid / 1e3, 2), int(((start <= mid) & (end > mid)).sum()), int((start <= mid).sum())])
    out = {
        "waves": int(nw), "sky_waves": int(sky.sum()), "frame_span_us(first start to last end)": round(span / 1e3, 2),
        "median_clock_GHz": round(float(clk), 3),
        "start_us pctl[0,10,50,90,99,100]": [round(x / 1e3, 2) for x in np.percentile(start, [0, 10, 50, 90, 99, 100])],
        "end_us pctl[0,10,50,90,99,100]": [round(x / 1e3, 2) for x in np.percentile(end, [0, 10, 50, 90, 99, 100])],
        "profile[t_us, resident waves, started so far]": prof,
        "cycles sky": {"prologue": q(pro[sky]), "body": q(body[sky]), "stores": q(sto[sky]), "life": q(life[sky])},
        "cycles hit": {"prologue": q(pro[~sky]), "body": q(body[~sky]), "stores": q(sto[~sky]), "life": q(life[~sky])},
        "sum_Mcycles": {"prologue": round(pro.sum() / 1e6, 1), "body_sky": round(body[sky].sum() / 1e6, 1),
                        "body_hit": round(body[~sky].sum() / 1e6, 1), "stores": round(sto.sum() / 1e6, 1)},
        "simds_seen": int(len(keys)), "waves_per_simd[min,med,max]": [int(counts.min()), int(np.median(counts)), int(counts.max())],
        "simd_finish_us pctl": [round(x / 1e3, 2) for x in np.percentile(per_simd_end, [0, 10, 50, 90, 99, 100])],
        "simd_sum_of_wave_life_cycles pctl": q(per_simd_life),
    }
    # inside the tile body of waves with candidates (diagnostic stamps 8..14 of mrt_small.hip)
    hitm = (~sky) & (t[:, 14] > 0) & (t[:, 13] > 0)
    seg = [("rays", 1, 8), ("closest hits", 8, 9), ("hit records", 9, 10), ("bundle", 10, 11), ("shadow candidates", 11, 12),
           ("shadow rays", 12, 13), ("shading", 13, 14), ("combine to stores", 14, 2), ("stores", 2, 3)]
    out["hit tile segments cycles pctl"] = {n: q((t[:, b] - t[:, a])[hitm]) for n, a, b in seg}
    out["hit tiles with shadow work"] = int(hitm.sum())
    rows = life.reshape((H + 7) // 8, W // 16).mean(axis=1)
    out["mean_life_by_tile_row_every_8th"] = [int(x) for x in rows[::8]]
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
        np.save(os.path.splitext(sys.argv[1])[0] + "_raw.npy", t)


if __name__ == "__main__":
    main()
