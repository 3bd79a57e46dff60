"""Experiment: kernel time of the C2 frame under different env-selected kernel variants."""
import sys, os, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from raytracing_amd import capi, scenes, multigpu
    cfg = sys.argv[2]
    if cfg == "C2":
        flat, cam = scenes.default_scene(); W, H = 1920, 1080
        cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
    else:
        if cfg == "REF":
            flat, cam = scenes.default_scene(); W, H = 2048, 1024
            cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
        spec = {"REF": lambda: None, "C3": lambda: scenes.sphere_field(10000), "C5": scenes.light_stress,
                "C4s": lambda: scenes.triangle_terrain(100, 50), "C4": lambda: scenes.triangle_terrain(1000, 500),
                "C4m": lambda: scenes.triangle_terrain(300, 150)}[cfg]()
        if cfg != "REF":
            W, H = (3840, 2160) if cfg in ("C3", "C4", "C4m") else (1920, 1080)
            flat = spec.flatten(); cam12 = spec.cam12(W / H)
    fr = multigpu.FrameRenderer(flat, cam12, W, H, **({"spp": 121, "max_depth": 8} if cfg == "REF" else {}))
    n = 400 if cfg == "C2" else (40 if cfg in ("C3", "C5") else 10)
    best = None
    for rep in range(3):   # HIP events around n back-to-back launches; best of three
        for _ in range(max(2, n // 10)): fr.render_local()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0.record()
        for _ in range(n): fr.render_local()
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e6
        ev = e0.elapsed_time(e1) / n * 1e3
        best = ev if best is None else min(best, ev)
    print(json.dumps({"us_per_frame_events_best": round(best, 3), "us_per_frame_wall_last": round(wall, 3)}))
    sys.exit(0)
cfgs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["C2"]
variants = [v for v in (sys.argv[2].split(";") if len(sys.argv) > 2 else [""])]
for cfg in cfgs:
    for v in variants:
        env = dict(os.environ)
        for kv in v.split(","):
            if kv: k, val = kv.split("="); env[k] = val
        out = subprocess.run([sys.executable, __file__, "child", cfg], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(cfg, v or "default", line[-1] if line else out.stderr[-300:], flush=True)
