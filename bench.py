#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (primary + shadow) of the render hot path.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5|REF]

A "step" is one frame of the configured workload: every rank renders its strips of the frame
with the HIP back end (scene already resident in HBM), and for N > 1 the strips are gathered
to rank 0 over RCCL and assembled.  Default workload = BASELINE.json configs[1]: the
reference's default scene (src/main.cu:183-237) at 1920x1080, 1 spp, primary + shadow rays.
REF (not a BASELINE config) is what the reference binary itself renders: the same scene at 2048x1024,
121 samples per pixel, bounce chains to depth 8 (src/main.cu:26, :165).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      -- dominant kernel (render_kernel) vs the HBM roof, algorithmic bytes/launch
  roofline_valu -- the same launches vs the fp32 vector-ALU roof, algorithmic flops/launch
  cpu_baseline  -- the CPU oracle ("port") timed on this host, bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak (packed FMA); no-FMA packed ceiling is half, plain a quarter
BYTES_PER_PIXEL = 20         # RGB f32 + id i32 + t f32


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


def measured_traffic(config, alg_bytes):
    """HBM bytes per launch from the committed PMC passes (profiles/hbm_traffic.json, produced by
    tools/pmc_bench.sh: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs of this command).
    gfx950 correction from MI355X_MICROARCH.md: FETCH_SIZE counts half the bytes of wide reads -> x2;
    both counters are in KiB."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f).get(config)
        if not t:
            return None
        return int((2.0 * t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024)
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(name, spec, flat, cam12, W, H, spp=1, max_depth=0):
    """Time the CPU oracle (a scalar port of the reference path) on this host: bounded sample of the same
    workload.  Brute force costs rays x primitives x (1 + lights), so the sample frame is shrunk (same scene,
    same camera, fewer pixels) until one frame is about 1e9 primitive tests per thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O  # test infrastructure, used here only as the reported CPU baseline

    scene = O.Scene.default() if spec is None else O.Scene.from_spec(spec)
    threads = min(16, os.cpu_count() or 1)  # the GPU box's CPU share for one GPU is 16
    per_ray = max(1, flat.num_spheres + flat.num_triangles) * (1 + flat.num_lights) * spp
    out = {}
    for label, nt, budget_tests, budget_s in (("1thread", 1, 1.0e9, 6.0), ("threads", threads, 1.0e9 * threads, 10.0)):
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
    v1, vn = out["1thread"], out["threads"]
    return {
        "value": round(vn[0], 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": f"{vn[1]} frame(s) of the same scene and camera at {vn[3]}x{vn[4]} in {vn[2]:.1f}s on {threads} threads "
                  f"(rows split statically); single thread: {v1[0]:.4f} Mrays/s over {v1[1]} frame(s) at {v1[3]}x{v1[4]}",
        "value_1thread": round(v1[0], 4),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ids", action="store_true", help="write RGB only (12 B/pixel)")
    ap.add_argument("--brute-force", action="store_true", help="use the every-ray-every-primitive kernel")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from raytracing_amd import multigpu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback of the product path")
    if os.environ.get("MRT_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % torch.cuda.device_count()  # rehearsal: ranks share the GPUs there are
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MRT_BENCH_BACKEND", "nccl")  # "gloo": rehearsal on a box with fewer GPUs than ranks
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    def progress(msg):  # stderr only: stdout carries the one JSON line
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    progress(f"building workload {args.config}")
    flat, cam12, W, H, flops_ray, scene_bytes, spec, label, spp, max_depth = build_workload(args.config)
    heavy = args.config in ("C3", "C4")
    heavy = heavy or args.config == "REF"
    steps = args.steps if args.steps is not None else (3 if args.config == "C4" else (20 if heavy else 500))
    warmup = args.warmup if args.warmup is not None else (1 if heavy else 50)

    fr = multigpu.FrameRenderer(flat, cam12, W, H, rank=rank, num_ranks=world, device=local_rank,
                                with_ids=not args.no_ids, spp=spp, max_depth=max_depth)
    fr.ctx.set_cull(0 if args.brute_force else 1)
    bytes_pp = 12 if args.no_ids else BYTES_PER_PIXEL
    if max_depth > 0:
        bytes_pp += 20  # the pixel's random stream: five xorshift words read per frame (mrt_rng.hip)

    progress("scene uploaded; counting rays")
    # rays per frame (deterministic for a scene + camera): counted once, outside the timed region
    mine = fr.count_rays()
    cnt = torch.tensor([mine["primary_rays"], mine["shadow_rays"], fr.local_rows * W], dtype=torch.int64,
                       device=fr.dev)
    if world > 1:
        dist.all_reduce(cnt)
    primary, shadow, _ = (int(x) for x in cnt.cpu())
    rays_per_frame = primary + shadow

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    progress(f"{rays_per_frame} rays per frame; warmup {warmup}, steps {steps}")
    for _ in range(warmup):
        fr.step()
    sync()
    # HIP events on the stream the kernel is launched on (torch's current stream).  At N=1 with many steps the
    # events bracket groups of 8 back-to-back launches (a frame lasts ~20 us; an event pair around a single
    # launch adds its own microsecond or two) and the group time is divided by 8; otherwise they bracket the
    # render kernel of every step (the gather/assembly follow the second event).
    group = 8 if (world == 1 and steps >= 64) else 1
    ev = {k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for k in range(0, steps - group + 1, group)}
    t0 = time.perf_counter()
    for k in range(steps):
        e = ev.get(k - k % group)
        if e and k % group == 0:
            e[0].record()
        fr.render_local()
        if e and k % group == group - 1:
            e[1].record()
        if world > 1:
            multigpu.gather_strips(fr.rgb, world, rank, dst=0, out=fr.gathered)
            if rank == 0:
                fr.ctx.assemble_strips(fr.gathered.data_ptr(), fr.gathered.shape[1], fr.frame.data_ptr(), W, H, 3,
                                       world, fr.strip_rows, stream=torch.cuda.current_stream().cuda_stream)
    sync()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=fr.dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    kern_ms = sum(a.elapsed_time(b) for a, b in ev.values()) / (len(ev) * group)

    if rank == 0:
        ms_per_step = elapsed / steps * 1e3
        value = rays_per_frame * steps / elapsed / 1e6
        # dominant kernel = render_kernel; per launch this rank processes its own strips
        my_pixels = fr.local_rows * W
        my_rays = mine["primary_rays"] + mine["shadow_rays"]
        alg_bytes = my_pixels * bytes_pp + scene_bytes
        alg_flops = my_rays * flops_ray
        ach_gbs = alg_bytes / (kern_ms * 1e-3) / 1e9
        ach_tf = alg_flops / (kern_ms * 1e-3) / 1e12
        small = flat.num_spheres + flat.num_triangles <= 64
        if max_depth > 0:
            kernel_name = "render_bounce_small_kernel" if small else "render_bounce_kernel"
            kernel_desc = ("render_bounce_small_kernel: 8x8-pixel tiles, every lane walks its pixel's samples and bounce "
                           "chains as its own state machine (XORWOW stream per pixel); per-tile candidates for primary "
                           "rays, shadow bundles, brute force further down the chains")
        elif not fr.ctx.cull:
            kernel_name, kernel_desc = "render_kernel", "render_kernel (brute force: every ray x every primitive)"
        elif small:
            kernel_name = "render_small_kernel"
            kernel_desc = ("render_small_kernel: 16x8-pixel tiles, 2 pixels per lane, per-tile culling (cone test for "
                           "spheres, screen-space edge functions for triangles) + exact tests")
        else:
            kernel_name = "render_cull_kernel"
            kernel_desc = ("render_cull_kernel: per-tile conservative culling over the kd-ordered primitive "
                           "hierarchy + exact tests")
        out = {
            "metric": "Mrays/s (primary+shadow) at 1920x1080" if args.config in ("C2", "C5") else
                      "Mrays/s (primary+shadow at every level) at 2048x1024, 121 spp, depth 8" if args.config == "REF" else
                      "Mrays/s (primary+shadow) at 3840x2160",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {label}", "width": W, "height": H, "spp": spp, "max_depth": max_depth,
                       "primary_rays": primary, "shadow_rays": shadow,
                       "parallelism": f"{world} GPU(s): 8-row strips round-robin, scene replicated" +
                                      ("" if world == 1 else ", RCCL gather to rank 0 + strip assembly"),
                       "outputs": "rgb f32 + id i32 + t f32" if not args.no_ids else "rgb f32",
                       "kernel": kernel_desc},
            "roofline": {"bound": "hbm", "achieved": round(ach_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_gbs / HBM_PEAK_GBS, 5),
                         "traffic": measured_traffic(args.config, alg_bytes) if world == 1 else None,
                         "kernel": kernel_name, "kernel_ms": round(kern_ms, 5),
                         "algorithmic_bytes_per_launch": alg_bytes},
            "roofline_valu": {"bound": "fp32 vector ALU (no MFMA: no dense contraction)",
                              "achieved": round(ach_tf, 4), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(ach_tf / VALU_PEAK_TFLOPS, 5),
                              "frac_of_no_fma_packed_ceiling": round(ach_tf / (VALU_PEAK_TFLOPS / 2), 5),
                              "algorithmic_flops_per_ray": flops_ray, "algorithmic_flops_per_launch": alg_flops},
        }
        if world == 1 and not args.no_cpu_baseline:
            progress(f"GPU done ({ms_per_step:.4f} ms/step); timing the CPU oracle (bounded sample, ~20 s)")
            out["cpu_baseline"] = cpu_baseline(args.config, spec, flat, cam12, W, H, spp, max_depth)
        print(json.dumps(out), flush=True)

    fr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
