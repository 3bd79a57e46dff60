"""The N > 1 bench path started the way the driver starts it, on the one GPU a test box has: `python bench.py
--gpus N` with no WORLD_SIZE launches its own ranks; MRT_BENCH_BACKEND=gloo lets them share GPU 0 (RCCL wants one
GPU per rank: the transfers go through the host here, everything else -- partition, frame_layout 1 on rank 0,
packed buffers, staging offsets, mrt_scatter_strips -- is the product path).  Checks the JSON line and that the frame
gathered on rank 0 is the 1-GPU frame, bit for bit, in all three planes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, n, config, extra=(), gather=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MRT_BENCH_BACKEND"] = "gloo"
    if gather:
        env["MRT_BENCH_GATHER"] = gather
    out = tmp_path / f"frame_{config}_{n}.npz"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--config",
           config, "--no-cpu-baseline", "--dump-frame", str(out), *extra]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), np.load(out), p.stderr


@pytest.mark.timeout(400)
@pytest.mark.parametrize("config", ["C2", "C5"])
def test_bench_direct_invocation_with_two_and_four_ranks(tmp_path, config):
    one, f1, _ = _bench(tmp_path, 1, config)
    assert one["n_gpus"] == 1 and one["roofline"]["kernel_ms"] > 0 and one["end_to_end"]["ms"] > 0
    assert [c["name"] for c in one["configs"]] == [config]
    if config == "C2":  # small scene: the first-view time beside the repeated-view headline
        assert one["first_view_ms"] > 0 and one["first_view"]["value"] > 0
    for n in (2, 4):
        js, fn, _ = _bench(tmp_path, n, config)
        assert js["n_gpus"] == n and js["steps"] == 3 and js["value"] > 0
        assert js["config"]["primary_rays"] == one["config"]["primary_rays"]
        assert js["config"]["shadow_rays"] == one["config"]["shadow_rays"]
        assert js["config"]["outputs"].startswith("rgb+id+t")
        assert js["multi_gpu"]["pipeline_depth"] == 2 and js["multi_gpu"]["exchange_overlaps_next_render"]
        chk = js["frame_check"]  # the bench's own 1-vs-N comparison, made on rank 0 after the timed region
        assert chk["ok"] and chk["rgb_equal"] and chk["id_equal"] and chk["t_equal"], chk
        for k in ("rgb", "id", "t"):
            assert np.array_equal(fn[k], f1[k]), (n, k)


@pytest.mark.timeout(400)
def test_bench_every_config_in_one_run_at_two_ranks(tmp_path):
    """The default run measures every workload and reports them in `configs`; here at N = 2 over gloo with the three
    cheap ones (the 4K scenes and the 121-spp frame take the oracle-free GPU path too, but minutes through the host
    transfers of a rehearsal).  Every entry carries its own timing, roofline and frame check."""
    js, _, _ = _bench(tmp_path, 2, "C2,C5,C3", extra=("--blocks", "3"))
    assert [c["name"] for c in js["configs"]] == ["C2", "C5", "C3"]
    assert js["value"] == js["configs"][0]["value"] and "frame_check_failed" not in js
    for c in js["configs"]:
        assert c["value"] > 0 and c["ms_per_step"] > 0 and c["kernel_ms"] > 0 and c["blocks"]["n"] == 3
        assert c["blocks"]["ms_per_step_min"] <= c["ms_per_step"] <= c["blocks"]["ms_per_step_max"]
        assert c["roofline"]["frac"] > 0 and c["frame_check"]["ok"], c
        assert c["n_gpus"] == 2 and c["multi_gpu"]["planes_gathered"] == "rgb+id+t"


@pytest.mark.timeout(400)
@pytest.mark.parametrize("pipeline", [1, 2])
def test_bench_picture_only_gather_and_both_pipeline_depths(tmp_path, pipeline):
    """--planes rgb: the other ranks render and send the picture alone (12 of 20 bytes per pixel); --pipeline 1 | 2:
    exchange after the render on one stream, or on a side stream beside the next frame's render.  The picture that
    arrives is the 1-GPU picture in every combination."""
    one, f1, _ = _bench(tmp_path, 1, "C2")
    js, fn, _ = _bench(tmp_path, 3, "C2", extra=("--planes", "rgb", "--pipeline", str(pipeline), "--steps", "5"))
    assert js["multi_gpu"]["planes_gathered"] == "rgb" and js["multi_gpu"]["bytes_per_pixel_gathered"] == 12
    assert js["multi_gpu"]["pipeline_depth"] == pipeline
    assert js["frame_check"]["ok"] and "id_equal" not in js["frame_check"]
    assert np.array_equal(fn["rgb"], f1["rgb"]) and "id" not in fn.files


@pytest.mark.timeout(400)
@pytest.mark.parametrize("n,share", [(2, "1"), (3, "5"), (2, "63"), (4, "auto")])
def test_bench_weighted_partition(tmp_path, n, share):
    """--root-share K: rank 0 renders K of N - 1 + K shares of the frame in place (one render, mrt_partition.rank_span),
    the others one share each; auto picks K from the one-GPU frame time against the bytes to gather.  The frame that
    arrives is the 1-GPU frame whatever the shares."""
    one, f1, _ = _bench(tmp_path, 1, "C2")
    js, fn, _ = _bench(tmp_path, n, "C2", extra=("--root-share", share))
    mg = js["multi_gpu"]
    if share != "auto":
        assert mg["weighted_partition"] == (share != "1") and mg["root_share"] == f"{share} of {n - 1 + int(share)} shares"
    else:
        assert "auto" in mg["root_share_how"]
    assert js["frame_check"]["ok"], js["frame_check"]
    for k in ("rgb", "id", "t"):
        assert np.array_equal(fn[k], f1[k]), (n, share, k)


@pytest.mark.timeout(300)
def test_bench_falls_back_when_the_ranks_cannot_form_a_communicator(tmp_path):
    """The communicator set-up of the N > 1 bench on one GPU: rank 0 makes the id, torch.distributed hands it round,
    every rank calls mrt_comm_create -- which RCCL refuses here (two ranks on one device) -- and all ranks agree to
    carry the exchange over torch.distributed instead.  The frame is still the 1-GPU frame."""
    one, f1, _ = _bench(tmp_path, 1, "C2")
    js, fn, err = _bench(tmp_path, 2, "C2", gather="capi-force")
    assert "mrt_comm_create failed" in err
    assert "torch.distributed point-to-point" in js["config"]["parallelism"]
    for k in ("rgb", "id", "t"):
        assert np.array_equal(fn[k], f1[k]), k
