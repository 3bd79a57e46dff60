"""Experiment: the bounce kernel's teams on the reference's own frame (2048x1024, 121 spp, depth 8).
usage: python tools/exp_bounce_team.py            sweeps MRT_BOUNCE_TEAM_MIN / MRT_BOUNCE_TEAM_CAP in child processes
       python tools/exp_bounce_team.py child      one measurement with the environment as it is
Every child prints the kernel time (HIP events inside mrt_render: the three ordering kernels included) and a hash of
the three planes: the hashes must all be equal."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    from raytracing_amd import capi, scenes
    W, H = 2048, 1024
    flat, _ = scenes.default_scene()
    ctx = capi.Context(0); ctx.upload(flat)
    cam = capi.default_viewport(W / H)
    ms = []
    for i in range(6):
        g = ctx.render(cam, W, H, spp=121, max_depth=8)
        ms.append(g["stats"]["kernel_ms"])
    h = hashlib.sha256(g["rgb"].tobytes() + g["id"].tobytes() + g["t"].tobytes()).hexdigest()[:12]
    print("min %.3f ms  median %.3f ms  hash %s  shadow %d" % (min(ms[1:]), sorted(ms[1:])[2], h, g["stats"]["shadow_rays"]), flush=True)
    sys.exit(0)

for tmin, cap in [(10, 0), (10, 256), (10, 1024), (10, 4096), (9, 1024), (8, 1024), (8, 4096), (6, 2048), (4, 4096)]:
    env = dict(os.environ, MRT_BOUNCE_TEAM_MIN=str(tmin), MRT_BOUNCE_TEAM_CAP=str(cap))
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=120)
    print("team_min %2d cap %4d: %s" % (tmin, cap, p.stdout.strip() or p.stderr[-300:]), flush=True)
    if p.returncode != 0:
        break
