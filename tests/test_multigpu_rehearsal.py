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
    for n in (2, 4):
        js, fn, _ = _bench(tmp_path, n, config)
        assert js["n_gpus"] == n and js["steps"] == 3 and js["value"] > 0
        assert js["config"]["primary_rays"] == one["config"]["primary_rays"]
        assert js["config"]["shadow_rays"] == one["config"]["shadow_rays"]
        assert "all three gathered" in js["config"]["outputs"]
        chk = js["frame_check"]  # the bench's own 1-vs-N comparison, made on rank 0 after the timed region
        assert chk["rgb_equal"] and chk["id_equal"] and chk["t_equal"], chk
        for k in ("rgb", "id", "t"):
            assert np.array_equal(fn[k], f1[k]), (n, k)


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
