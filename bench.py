#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (primary + shadow) of the render hot path.

    python bench.py --gpus N --steps K --warmup W [--config all|C2|C3|C4|C5|REF|C2,C3,...]

A "step" is one frame of a workload: every rank renders its strips of the frame with the HIP back end (scene already
resident in HBM); for N > 1 every rank but 0 then sends its strips to rank 0 over RCCL (one message per peer) and
rank 0 puts them in place.  The HEADLINE workload (the top-level fields of the JSON line) is the first config named:
by default BASELINE.json configs[1], the reference's default scene (src/main.cu:183-237) at 1920x1080, 1 spp, primary
+ shadow rays (C2).  With no --config every workload is measured in the same run and reported in the `configs`
array of the same line: C2, C3 (10k spheres, 4K), C4 (1M triangles, 4K), C5 (64 point lights, 1080p) and REF -- not
a BASELINE config: what the reference binary itself renders, the default scene at 2048x1024, 121 samples per pixel,
bounce chains to depth 8 (src/main.cu:26, :165).

Timing.  After W warm-up frames and an untimed settle phase (GPU clocks), a config is timed as R BLOCKS of exactly K
frames, each block bracketed by a barrier + torch.cuda.synchronize() on both sides and taken as the MAX over ranks;
R is chosen so that a config gets at least ~50 ms of GPU time (a 20-frame block of a 17 us frame is 0.35 ms: one such
block measures the closing synchronisation as much as the frames).  `value` and `ms_per_step` come from the MEDIAN
block; min / max are reported beside them.

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks
(python -m torch.distributed.run ... bench.py, as the driver would) before it touches a GPU, relays rank 0's JSON
line and exits with the children's status.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      -- the render kernel against the HBM roof: algorithmic bytes per launch / its HIP-event time
  executed_work -- what the kernel executed per launch according to the committed rocprofv3 counter passes
                   (profiles/counters.json): vector instructions, their issue time, waves per SIMD, HBM bytes; each
                   entry carries the hash of the kernel sources it was taken from and `stale` says whether that is
                   the tree this run was built from
  first_view    -- small scenes: the same frames with the launcher's view memory off (MRT_OPT_VIEW_CACHE 0): what a
                   camera that moves every frame gets
  end_to_end    -- blocking mrt_render into page-locked host memory (launch + sync + frame in host memory, which is
                   what src/main.cu:239-247 does), N = 1 only, outside the timed region
  cpu_baseline  -- the CPU oracle ("port") timed on this host, bounded sample (rank 0, N = 1 only)
  configs       -- one entry per workload measured in this run (the headline's included)
  frame_check   -- N > 1: the gathered frame against rank 0 rendering all of it alone, bit for bit; a difference makes
                   the run exit with status 3
"""
import argparse
import hashlib
import json
import math
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (packed FMA); parity forbids FMA: a quarter of it with plain mul / add
BYTES_PER_PIXEL = 20         # RGB f32 + id i32 + t f32
SIMDS = 1024                 # 256 CUs x 4
VALU_ISSUE_NS = 2.0 / 2.4    # MI355X_MICROARCH.md: a wave64 vector instruction issues over 2 cycles at 2.4 GHz
ALL_CONFIGS = ("C2", "C3", "C4", "C5", "REF")
MIN_GPU_MS = 50.0            # GPU time a config gets in its timed blocks, at least
MAX_BLOCKS = 200


def build_workload(name):
    """-> (flat scene, cam12, W, H, flops_per_ray, scene_bytes, oracle scene factory, label, spp, max_depth)"""
    from raytracing_amd import capi, scenes

    spp, max_depth = 1, 0
    if name == "REF":
        flat, cam = scenes.default_scene()
        W, H, spp, max_depth = 2048, 1024, 121, 8
        cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
        label = "the reference binary's own frame: default main.cu scene, 2048x1024, 121 spp, bounce chains to depth 8"
        spec = None
    elif name == "C2":
        flat, cam = scenes.default_scene()
        W, H = 1920, 1080
        cam12 = capi.viewport(cam["look_from"], cam["look_at"], cam["up"], cam["vfov"], W / H)
        label = "default main.cu scene (4 spheres + 6 triangles, 1 directional light), 1920x1080, 1 spp, depth 0"
        spec = None
    else:
        if name == "C3":
            spec, (W, H) = scenes.sphere_field(10000), (3840, 2160)
            label = "10k random spheres, 3840x2160, 1 spp, depth 0"
        elif name == "C4":
            spec, (W, H) = scenes.triangle_terrain(1000, 500), (3840, 2160)
            label = "1M-triangle terrain, 3840x2160, 1 spp, depth 0"
        elif name == "C5":
            spec, (W, H) = scenes.light_stress(), (1920, 1080)
            label = "256 spheres + 256 triangles, 64 point lights, 1920x1080, 1 spp, depth 0"
        else:
            raise SystemExit(f"unknown config {name}")
        flat = spec.flatten()
        cam12 = spec.cam12(W / H)
    flops = 23 * flat.num_spheres + 36 * flat.num_triangles
    scene_bytes = 16 * flat.num_spheres + 64 * flat.num_triangles
    return flat, cam12, W, H, flops, scene_bytes, spec, label, spp, max_depth


def source_hash():
    """Hash of the sources libmrt.so is built from (csrc/*.hip, *.h, *.cpp, the Makefile and include/mrt.h): what a
    committed counter pass is tied to (tools/summarize_profile.py writes it into profiles/counters.json)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracing_amd", "csrc")
    names = sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile")
    for f in [os.path.join(d, n) for n in names] + [os.path.join(ROOT, "include", "mrt.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_counters(config, brute):
    """Per-launch rocprofv3 counters of the render kernel from the committed passes (profiles/counters.json, written
    by tools/profile_config.sh + tools/summarize_profile.py on the GPU box: SQ counters, FETCH_SIZE and WRITE_SIZE
    each in a --pmc run of its own of THIS command).  Not measured by this run: the JSON says where they come from
    and whether the kernel sources have changed since (`stale`)."""
    try:
        with open(os.path.join(ROOT, "profiles", "counters.json")) as f:
            return json.load(f).get(config + ("_brute" if brute else ""))
    except (OSError, ValueError):
        return None


def cpu_baseline(name, spec, flat, cam12, W, H, spp=1, max_depth=0, quick=False):
    """Time the CPU oracle (a scalar port of the reference path) on this host: bounded sample of the same
    workload.  Brute force costs rays x primitives x (1 + lights), so the sample frame is shrunk (same scene,
    same camera, fewer pixels) until one frame is about 1e9 primitive tests per thread.  quick: the threaded leg
    only, a quarter of that and at most ~3 s (the extra configs of a default run)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O  # test infrastructure, used here only as the reported CPU baseline

    scene = O.Scene.default() if spec is None else O.Scene.from_spec(spec)
    threads = min(16, os.cpu_count() or 1)  # the GPU box's CPU share for one GPU is 16
    per_ray = max(1, flat.num_spheres + flat.num_triangles) * (1 + flat.num_lights) * spp
    legs = (("threads", threads, 2.5e8 * threads, 3.0),) if quick else \
           (("1thread", 1, 1.0e9, 6.0), ("threads", threads, 1.0e9 * threads, 10.0))
    out = {}
    for label, nt, budget_tests, budget_s in legs:
        scale = 1
        while (W // scale) * (H // scale) * per_ray > budget_tests and min(W, H) // (scale * 2) >= 8:
            scale *= 2
        sw, sh = W // scale, H // scale
        frames, t0, rays = 0, time.perf_counter(), 0
        while True:
            r = scene.render(cam12, sw, sh, spp=spp, max_depth=max_depth, nthreads=nt, want=("rgb", "id", "t"))
            rays += r["stats"]["primary_rays"] + r["stats"]["shadow_rays"]
            frames += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s or frames >= 200:
                break
        out[label] = (rays / dt / 1e6, frames, dt, sw, sh)
    vn = out["threads"]
    res = {
        "value": round(vn[0], 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"{vn[1]} frame(s) of the same scene and camera at {vn[3]}x{vn[4]} in {vn[2]:.1f}s on {threads} threads "
                  f"(rows split statically)",
    }
    if "1thread" in out:
        v1 = out["1thread"]
        res["sample"] += f"; single thread: {v1[0]:.4f} Mrays/s over {v1[1]} frame(s) at {v1[3]}x{v1[4]}"
        res["value_1thread"] = round(v1[0], 4)
    return res


def end_to_end(fr, cam12, W, H, spp, max_depth, rays_per_frame, frames):
    """Blocking mrt_render of the whole frame into page-locked host memory (mrt_host_alloc): launch + sync + the
    frame readable by the host, as src/main.cu:239-247 leaves it.  Includes the 20 B/pixel copy over PCIe."""
    import numpy as np
    from raytracing_amd import capi

    hb = capi.HostBuffer(W * H * 20)
    rgb = hb.view(np.float32, (H, W, 3))
    ids = hb.view(np.int32, (H, W), offset=W * H * 12)
    t = hb.view(np.float32, (H, W), offset=W * H * 16)
    fr.ctx.render_into(cam12, W, H, rgb, ids, t, spp=spp, max_depth=max_depth)  # first call sizes the staging buffer
    t0 = time.perf_counter()
    for _ in range(frames):
        fr.ctx.render_into(cam12, W, H, rgb, ids, t, spp=spp, max_depth=max_depth)
    ms = (time.perf_counter() - t0) / frames * 1e3
    del rgb, ids, t
    hb.free()
    return {"ms": round(ms, 4), "Mrays/s": round(rays_per_frame / ms / 1e3, 1), "buffer": "page-locked",
            "frames": frames, "what": "blocking mrt_render: kernel + device-to-host copy of rgb, id and t + sync"}


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """bench.py --gpus N called directly: start N ranks the way the driver does, before any GPU call here (a process
    that has initialised the GPU must not be replaced, and need not be: this parent only relays).  Returns the
    children's exit status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    elif proc.stdout:
        sys.stderr.write(proc.stdout[-4000:])
    if proc.returncode == 0 and line is None:
        print("bench.py: the ranks exited cleanly but printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


class Env:
    """What every config of a run shares: the process group, the device, the flags."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.args = torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("MRT_BENCH_BACKEND", "nccl")  # "gloo": rehearsal on a box with fewer GPUs than ranks
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: there is no CPU fallback of the product path")
        if self.backend != "nccl":
            local_rank = local_rank % torch.cuda.device_count()  # rehearsal: ranks share the GPUs there are
        elif local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {self.rank}: {self.world} ranks over RCCL need {self.world} GPUs, this node shows "
                             f"{torch.cuda.device_count()} (MRT_BENCH_BACKEND=gloo rehearses the path on fewer)")
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
            else:
                dist.init_process_group(self.backend)
        self.src_hash = source_hash()

    def progress(self, msg):  # stderr only: stdout carries the one JSON line
        if self.rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def sync(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()


def make_comm(env, fr):
    """The exchange step's carrier: mrt_gather_strips (RCCL behind the C-ABI) on a communicator made from an id that
    rank 0 creates and torch.distributed hands round.  If any rank cannot create it (or MRT_BENCH_GATHER=torch, or a
    gloo rehearsal with ranks sharing a GPU) every rank uses torch.distributed's point-to-point calls instead."""
    torch, dist = env.torch, env.dist
    if env.world == 1:
        return "none"
    gather = "torch.distributed point-to-point (" + env.backend + ")"
    want = os.environ.get("MRT_BENCH_GATHER", "capi")  # "torch": skip; "capi-force": try even in a gloo rehearsal
    if (env.backend == "nccl" and want == "capi") or want == "capi-force":
        from raytracing_amd import capi
        uid = [None]
        ok = 1
        try:
            if env.rank == 0:
                uid[0] = capi.comm_unique_id()
        except capi.MrtError as e:
            env.progress(f"mrt_comm_unique_id failed: {e}")
        dist.broadcast_object_list(uid, src=0)
        try:
            if uid[0] is None:
                raise capi.MrtError("no communicator id")
            fr.init_comm(uid[0])
        except capi.MrtError as e:
            ok = 0
            print(f"[bench] rank {env.rank}: mrt_comm_create failed: {e}", file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32, device=fr.dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            gather = "mrt_gather_strips (RCCL behind the C-ABI: ncclSend per peer, grouped ncclRecv on rank 0)"
        else:
            fr.drop_comm()
    return gather


def timed_blocks(env, fr, steps, nblocks, local_only):
    """nblocks blocks of exactly `steps` frames, each between two barrier + synchronise pairs.  -> (wall seconds per
    block, MAX over ranks; HIP-event ms per frame per block or None).  local_only: frames are fr.render_local() with
    one HIP-event pair on the launch stream (torch's current stream) around each block -- its time over `steps` is
    the kernel's average launch-to-launch duration (no per-step events: a pair costs 2-3 us of a 17 us frame)."""
    torch, dist = env.torch, env.dist
    wall, kern = [], []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nblocks)] \
        if local_only else None
    for b in range(nblocks):
        env.sync()
        t0 = time.perf_counter()
        if local_only:
            ev[b][0].record()
            for _ in range(steps):
                fr.render_local()
            ev[b][1].record()
        else:
            for _ in range(steps):
                fr.step()
        env.sync()
        wall.append(time.perf_counter() - t0)
    if local_only:
        kern = [e0.elapsed_time(e1) / steps for e0, e1 in ev]
    w = torch.tensor(wall, dtype=torch.float64, device=fr.dev)
    if env.world > 1:
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
    return [float(x) for x in w.cpu()], (kern if local_only else None)


def measure(env, name, headline):
    """One config, all of it: -> the config's entry (rank 0) or None."""
    from raytracing_amd import capi, multigpu

    torch, dist, args = env.torch, env.dist, env.args
    world, rank = env.world, env.rank
    env.progress(f"building workload {name}")
    flat, cam12, W, H, flops_ray, scene_bytes, spec, label, spp, max_depth = build_workload(name)
    steps = args.steps if args.steps is not None else 100
    warmup = args.warmup if args.warmup is not None else 5
    planes = {"all": capi.PLANES_ALL, "rgb": capi.PLANE_RGB}[args.planes]

    # N > 1: how many shares of the frame rank 0 takes (multigpu.suggest_root_share).  "auto": rank 0 times the whole
    # frame on its GPU alone and the ranks agree on the share that balances its render against what it has to receive.
    root_share, share_how = 1, None
    if world > 1:
        k = torch.zeros(1, dtype=torch.int64, device=torch.device(f"cuda:{env.local_rank}"))
        if args.root_share != "auto":
            k[0] = max(1, int(args.root_share))
            share_how = "--root-share"
        elif rank == 0:
            probe = multigpu.FrameRenderer(flat, cam12, W, H, device=env.local_rank, spp=spp, max_depth=max_depth)
            probe.ctx.set_cull(0 if args.brute_force else 1)
            for _ in range(5):
                probe.render_local()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                probe.render_local()
            e1.record()
            torch.cuda.synchronize()
            one_gpu_ms = e0.elapsed_time(e1) / 10
            probe.close()
            k[0] = multigpu.suggest_root_share(one_gpu_ms, W * H * 4 * multigpu.plane_dwords(planes), world,
                                               link_gbps=args.link_gbps)
        dist.broadcast(k, 0)
        root_share = int(k.item())
        if share_how is None:
            share_how = (f"auto: the whole frame on one GPU against the bytes rank 0 receives at {args.link_gbps} GB/s per "
                         "link (multigpu.suggest_root_share)")
    fr = multigpu.FrameRenderer(flat, cam12, W, H, rank=rank, num_ranks=world, device=env.local_rank, spp=spp,
                                max_depth=max_depth, planes=planes, pipeline=args.pipeline, root_share=root_share)
    fr.ctx.set_cull(0 if args.brute_force else 1)
    gather = make_comm(env, fr)
    bytes_pp = BYTES_PER_PIXEL
    if max_depth > 0:
        bytes_pp += 20  # the pixel's random stream: five xorshift words read per frame (mrt_rng.hip)

    env.progress("scene uploaded; counting rays")
    # rays per frame (deterministic for a scene + camera): counted once, outside the timed region
    mine = fr.count_rays()
    cnt = torch.tensor([mine["primary_rays"], mine["shadow_rays"], fr.local_rows * W], dtype=torch.int64,
                       device=fr.dev)
    if world > 1:
        dist.all_reduce(cnt)
    primary, shadow, _ = (int(x) for x in cnt.cpu())
    rays_per_frame = primary + shadow

    env.progress(f"{rays_per_frame} rays per frame; warmup {warmup}, blocks of {steps} steps")
    for _ in range(warmup):
        fr.step()
    env.sync()
    # The GPU raises its clocks only under sustained load: after idle, the default frame takes 18.0 us and settles at
    # 16.2 us once launches have been coming for ~12 ms (tools/exp_settle.py).  `warmup` frames of 17 us are over
    # before that, so untimed frames go on until SETTLE_MS have passed since the first: a steady-state rate is what the
    # metric means.
    settle_ms = float(os.environ.get("MRT_BENCH_SETTLE_MS", "60"))
    t_settle = time.perf_counter()
    fr.step()
    env.sync()
    one = torch.tensor([time.perf_counter() - t_settle], dtype=torch.float64, device=fr.dev)
    if world > 1:
        dist.all_reduce(one, op=dist.ReduceOp.MAX)  # every rank runs the same number of frames
    settle_frames = int(max(0, min(8192, settle_ms * 1e-3 / max(float(one.item()), 1e-6))))
    t_loop = time.perf_counter()
    for _ in range(settle_frames):  # back to back, no synchronisation in between
        fr.step()
    env.sync()
    settle_spent_ms = (time.perf_counter() - t_settle) * 1e3
    # frame time estimate -> number of blocks (the same on every rank)
    est = torch.tensor([(time.perf_counter() - t_loop) / max(settle_frames, 1) if settle_frames else float(one.item())],
                       dtype=torch.float64, device=fr.dev)
    if world > 1:
        dist.all_reduce(est, op=dist.ReduceOp.MAX)
    nblocks = args.blocks if args.blocks else \
        int(max(3, min(MAX_BLOCKS, math.ceil(MIN_GPU_MS * 1e-3 / max(steps * float(est.item()), 1e-9)))))

    # The timed region: nblocks x exactly `steps` frames.
    wall, kern = timed_blocks(env, fr, steps, nblocks, local_only=(world == 1))
    if world > 1:  # the render kernel alone, timed afterwards the same way on render-only frames
        ksteps = min(steps, 50)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(ksteps):
            fr.render_local()
        e1.record()
        torch.cuda.synchronize()
        kern_ms = e0.elapsed_time(e1) / ksteps
        kern_how = f"one HIP-event pair around {ksteps} back-to-back render-only launches on the launch stream"
    else:
        kern_ms = statistics.median(kern)
        kern_how = (f"one HIP-event pair around each block of {steps} back-to-back launches on the launch stream; "
                    f"median of {nblocks} blocks (min {min(kern):.5f}, max {max(kern):.5f})")
    med = statistics.median(wall)

    # small scenes, N = 1: what the FIRST render of a view costs (and every render of a camera that moves): the
    # launcher's view memory off, same frames, same timing
    first_view = None
    if world == 1 and not args.brute_force and max_depth == 0 and flat.num_spheres + flat.num_triangles <= 64:
        fr.ctx.set_view_cache(False)
        for _ in range(20):
            fr.render_local()
        w2, k2 = timed_blocks(env, fr, steps, nblocks, local_only=True)
        fr.ctx.set_view_cache(True)
        m2 = statistics.median(w2)
        first_view = {"kernel_ms": round(statistics.median(k2), 5), "ms_per_step": round(m2 / steps * 1e3, 5),
                      "value": round(rays_per_frame * steps / m2 / 1e6, 3),
                      "what": "MRT_OPT_VIEW_CACHE 0: every frame is launched as the first of its view, as for a camera "
                              "that moves every frame -- the launcher works out the tile row the launch starts with "
                              "from an estimate that costs the host microseconds (small_first_row_fast), on every "
                              "frame; the headline renders ONE view again and again and keeps the row (DESIGN.md 5.5)"}

    # N > 1: is the frame that arrived on rank 0 the frame one GPU renders?  Outside the timed region: one more
    # distributed frame, then rank 0 renders all of it alone and compares the planes that travelled bit for bit.
    frame_check = None
    if world > 1:
        fr.step()
        env.sync()
        if rank == 0:
            solo = multigpu.FrameRenderer(flat, cam12, W, H, device=env.local_rank, spp=spp, max_depth=max_depth)
            solo.ctx.set_cull(0 if args.brute_force else 1)
            solo.render_local()
            torch.cuda.synchronize()
            pairs = [("rgb", fr.rgb, solo.rgb), ("id", fr.ids, solo.ids), ("t", fr.tt, solo.tt)]
            same = {k: bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) for k, a, b in pairs if a is not None}
            solo.close()
            frame_check = {"what": f"the frame gathered from {world} ranks against rank 0 rendering all of it alone, "
                                   "bit for bit", **{k + "_equal": v for k, v in same.items()},
                           "ok": all(same.values())}
            if not all(same.values()):
                print(f"[bench] FRAME CHECK FAILED ({name}): {frame_check}", file=sys.stderr, flush=True)
        env.sync()
    if args.dump_frame and headline:
        frame = fr.frame_host()
        if rank == 0 and frame is not None:
            import numpy as np
            np.savez(args.dump_frame, **frame)

    out = None
    if rank == 0:
        ms_per_step = med / steps * 1e3
        value = rays_per_frame * steps / med / 1e6
        # dominant kernel = the render kernel; per launch this rank processes its own strips
        my_pixels = fr.local_rows * W
        my_rays = mine["primary_rays"] + mine["shadow_rays"]
        alg_bytes = my_pixels * bytes_pp + scene_bytes
        ach_gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
        small = flat.num_spheres + flat.num_triangles <= 64
        if max_depth > 0:
            pipe = small and os.environ.get("MRT_BOUNCE_PIPE", "1") != "0"
            kernel_name = ("render_bounce_pipe_kernel" if pipe else "render_bounce_small_kernel") if small else "render_bounce_kernel"
            kernel_desc = (kernel_name + ": 8x8-pixel tiles, every lane walks its pixel's samples and bounce chains as its "
                           "own state machine (XORWOW stream per pixel); per-tile candidates for primary rays, shadow "
                           "bundles, brute force further down the chains"
                           + ("; two waves per tile: one traces (closest hits, bounces, draws), one shades (shadow rays, "
                              "lights, level stack), an LDS ring between them" if pipe else ""))
        elif args.brute_force:
            kernel_name, kernel_desc = "render_kernel", "render_kernel (brute force: every ray x every primitive)"
        elif small:
            kernel_name = "render_small_kernel"
            kernel_desc = ("render_small_kernel: 16x8-pixel tiles, 2 pixels per lane, per-tile culling (screen rectangles "
                           "for spheres, screen-space edge functions for triangles) + exact tests")
        else:
            kernel_name = "render_cull_kernel"
            kernel_desc = ("render_cull_kernel: per-tile conservative culling over the kd-ordered primitive "
                           "hierarchy + exact tests")
        pmc = committed_counters(name, args.brute_force) if world == 1 else None
        stale = None
        if pmc:
            stale = pmc.get("source_hash") != env.src_hash  # an entry without a hash predates the provenance stamp
        traffic = None
        if pmc and "FETCH_SIZE_KB" in pmc and "WRITE_SIZE_KB" in pmc:
            # gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts half the bytes of wide reads; both in KiB
            traffic = int((2.0 * pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024)
        metric = ("Mrays/s (primary+shadow) at 1920x1080" if name in ("C2", "C5") else
                  "Mrays/s (primary+shadow at every level) at 2048x1024, 121 spp, depth 8" if name == "REF" else
                  "Mrays/s (primary+shadow) at 3840x2160")
        multi = fr.describe()
        out = {
            "name": name, "metric": metric,
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "warmup_settle": {"frames": settle_frames + 1, "ms": round(settle_spent_ms, 2),
                              "why": "untimed frames after the warm-up until the GPU clocks have settled (tools/exp_settle.py)"},
            "ms_per_step": round(ms_per_step, 5),
            "blocks": {"n": nblocks, "steps_each": steps, "which": "median",
                       "ms_per_step_min": round(min(wall) / steps * 1e3, 5),
                       "ms_per_step_max": round(max(wall) / steps * 1e3, 5),
                       "gpu_ms_total": round(sum(wall) * 1e3, 2),
                       "why": f"each block is exactly {steps} frames between barrier + synchronise pairs (MAX over "
                              f"ranks); enough blocks for >= {MIN_GPU_MS:.0f} ms of GPU time"},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}: {label}", "width": W, "height": H, "spp": spp, "max_depth": max_depth,
                       "primary_rays": primary, "shadow_rays": shadow,
                       "parallelism": f"{world} GPU(s): strips of {fr.strip_rows} rows round-robin, scene replicated" +
                                      ("" if world == 1 else ", one message per peer to rank 0 (" + multi["planes_gathered"]
                                       + "), rank 0 renders in place and scatters the rest; carrier: " + gather),
                       "outputs": "rgb f32 + id i32 + t f32" if world == 1 else
                                  multi["planes_gathered"] + " rendered and gathered on rank 0",
                       "kernel": kernel_desc},
            "roofline": {"bound": "hbm", "achieved": round(ach_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_gbs / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": pmc.get("source") if (pmc and traffic is not None) else None,
                         "traffic_stale": stale if traffic is not None else None,
                         "kernel": kernel_name, "kernel_ms": round(kern_ms, 5), "kernel_ms_how": kern_how,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "binding_resource": "vector-instruction issue, not HBM (see executed_work.valu_issue_measured_costs; DESIGN.md 5.5)"},
        }
        if world > 1:
            out["multi_gpu"] = {**multi, "root_share_how": share_how, "exchange_overlaps_next_render": multi["pipeline_depth"] > 1,
                                "carrier": gather}
        if pmc and "SQ_INSTS_VALU" in pmc:
            # what the kernel executed, not what a brute-force kernel would have: committed counter passes
            valu = pmc["SQ_INSTS_VALU"]
            issue_ms = valu / SIMDS * VALU_ISSUE_NS * 1e-6
            ex = {"source": pmc.get("source"), "source_hash": pmc.get("source_hash"), "git_rev": pmc.get("git_rev"),
                  "stale": stale, "tree_hash": env.src_hash,
                  "valu_insts_per_launch": int(valu),
                  "valu_issue_ms": round(issue_ms, 5),
                  "valu_issue_frac_of_kernel_time": round(issue_ms / kern_ms, 4),
                  "valu_issue_model": "SQ_INSTS_VALU / 1024 SIMDs x 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)"}
            mixk = ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
                    "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT")
            if all(k in pmc for k in mixk):
                # The 2-cycle model above prices every vector instruction as a plain multiply.  tools/ubench_mix.hip
                # measured what the kinds really cost per SIMD with 8 waves resident (ns per instruction): v_mul / v_add /
                # integer add 1.15, v_fma 1.68, v_rcp / v_sqrt / v_rsq and v_readlane 3.45, conversions 1.8, compares,
                # selects, min / max, DPP moves and the pieces of IEEE division 1.85-2.0, plain moves 0.98.  With the
                # kernel's own mix (counter pass sq3) that gives its issue time; "other" (everything the typed counters
                # do not name: moves, compares, selects, min / max, lane reads and writes, DPP, division pieces) is priced
                # at 1.8 and, as a floor, at 0.98 as if it were all moves.
                add, mul, fma, trans, i32, cvt = (pmc[k] for k in mixk)
                other = max(0.0, valu - (add + mul + fma + trans + i32 + cvt))
                typed_ns = (add + mul) * 1.15 + fma * 1.68 + trans * 3.45 + i32 * 1.2 + cvt * 1.8
                est_ms, low_ms = (typed_ns + other * 1.8) / SIMDS * 1e-6, (typed_ns + other * 0.98) / SIMDS * 1e-6
                ex["valu_issue_measured_costs"] = {
                    "ms": round(est_ms, 5), "frac_of_kernel_time": round(est_ms / kern_ms, 4),
                    "floor_ms": round(low_ms, 5), "floor_frac_of_kernel_time": round(low_ms / kern_ms, 4),
                    "mix": {"add_f32": int(add), "mul_f32": int(mul), "fma_f32": int(fma), "trans_f32": int(trans),
                            "int32": int(i32), "cvt": int(cvt), "other": int(other)},
                    "how": "per-kind instruction counts (SQ_INSTS_VALU_*) x the per-kind issue costs measured by "
                           "tools/ubench_mix.hip (profiles/r03_ubench_mix.txt), over 1024 SIMDs"}
            if pmc.get("rocprof_kernel_avg_ns"):
                ex["rocprof_kernel_avg_ms"] = round(pmc["rocprof_kernel_avg_ns"] * 1e-6, 5)
            if "SQ_WAVES" in pmc:
                ex["waves_per_launch"] = int(pmc["SQ_WAVES"])
            if pmc.get("GRBM_GUI_ACTIVE") and pmc.get("SQ_WAVE_CYCLES"):
                # SQ_WAVE_CYCLES: quad-cycles summed over waves; GRBM_GUI_ACTIVE: busy cycles summed over the 8 XCDs
                # (MI355X_MICROARCH.md) -> average resident waves per SIMD while the kernel runs
                ex["avg_waves_per_simd"] = round(4.0 * pmc["SQ_WAVE_CYCLES"] / (pmc["GRBM_GUI_ACTIVE"] / 8.0) / SIMDS, 2)
            if pmc.get("vgpr"):
                ex["vgpr"], ex["occupancy_waves_per_simd"] = pmc["vgpr"], pmc.get("occupancy")
            if pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("SQ_WAVE_CYCLES"):
                ex["valu_active_frac_of_wave_cycles"] = round(pmc["SQ_ACTIVE_INST_VALU"] / pmc["SQ_WAVE_CYCLES"], 4)
            out["executed_work"] = ex
        if args.brute_force:
            # only the brute-force kernel executes the algorithmic flop count (23 per sphere test, 36 per triangle test)
            alg_flops = my_rays * flops_ray
            ach_tf = alg_flops / (kern_ms * 1e-3) / 1e12
            out["roofline_valu"] = {"bound": "fp32 vector ALU (no MFMA: no dense contraction)",
                                    "achieved": round(ach_tf, 4), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(ach_tf / VALU_PEAK_TFLOPS, 5),
                                    "algorithmic_flops_per_ray": flops_ray, "algorithmic_flops_per_launch": alg_flops}
        if first_view is not None:
            out["first_view"] = first_view
            out["first_view_ms"] = first_view["kernel_ms"]
        if frame_check is not None:
            out["frame_check"] = frame_check
        if world == 1:
            env.progress(f"{name}: GPU done ({ms_per_step:.4f} ms/step median of {nblocks} blocks, kernel {kern_ms:.4f} ms)"
                         "; end-to-end frames")
            out["end_to_end"] = end_to_end(fr, cam12, W, H, spp, max_depth, rays_per_frame,
                                           3 if name in ("C4", "REF") else 20)
        if world == 1 and not args.no_cpu_baseline:
            env.progress("timing the CPU oracle (bounded sample" + (", ~20 s)" if headline else ", ~3 s)"))
            out["cpu_baseline"] = cpu_baseline(name, spec, flat, cam12, W, H, spp, max_depth, quick=not headline)
    fr.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="frames per timed block (default 100)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="all",
                    help="all (default: C2 as the headline, then C3, C4, C5, REF) | one config | a comma list; the "
                         "first one named is the headline")
    ap.add_argument("--blocks", type=int, default=0, help="timed blocks per config (default: enough for 50 ms of GPU time)")
    ap.add_argument("--planes", choices=("all", "rgb"), default="all",
                    help="N > 1: what the ranks render for and send to rank 0 (rgb: the picture alone, 12 of 20 B/pixel)")
    ap.add_argument("--pipeline", type=int, default=2, choices=(1, 2),
                    help="N > 1: frames in flight (2: the exchange of a frame overlaps the next frame's render)")
    ap.add_argument("--root-share", default="auto",
                    help="N > 1: shares of the frame rank 0 renders of N - 1 + that many (1: every rank the same; auto: "
                         "balanced against the exchange)")
    ap.add_argument("--link-gbps", type=float, default=45.0, help="achievable GB/s of one xGMI link, for --root-share auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--brute-force", action="store_true", help="use the every-ray-every-primitive kernel")
    ap.add_argument("--dump-frame", default=None, help="rank 0 saves the headline's last frame (rgb, id, t) to this .npz")
    args = ap.parse_args()
    names = list(ALL_CONFIGS) if args.config == "all" else [c.strip() for c in args.config.split(",") if c.strip()]
    for c in names:
        if c not in ALL_CONFIGS:
            raise SystemExit(f"unknown config {c}")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    env = Env(args)
    args.gpus = env.world
    entries = []
    for i, name in enumerate(names):
        e = measure(env, name, headline=(i == 0))
        if e is not None:
            entries.append(e)
    status = 0
    if env.rank == 0:
        head = dict(entries[0])
        del head["name"]
        # the other workloads of this run, compact: what the judge needs per config (the headline's entry repeats
        # the top-level numbers so that the array is complete by itself)
        keep = ("name", "metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "blocks", "config", "roofline",
                "executed_work", "first_view_ms", "frame_check", "multi_gpu", "end_to_end", "cpu_baseline")
        head["configs"] = [{k: e[k] for k in keep if k in e} | {"kernel_ms": e["roofline"]["kernel_ms"]} for e in entries]
        bad = [e["name"] for e in entries if e.get("frame_check") and not e["frame_check"]["ok"]]
        if bad:
            head["frame_check_failed"] = bad
            status = 3
        print(json.dumps(head), flush=True)
    if env.world > 1:
        flag = env.torch.tensor([status], dtype=env.torch.int32, device=f"cuda:{env.local_rank}")
        env.dist.broadcast(flag, src=0)
        status = int(flag.item())
        env.dist.barrier()
        env.dist.destroy_process_group()
    sys.exit(status)


if __name__ == "__main__":
    main()
