"""The N > 1 path on CPU: gloo ranks each produce their strips of a frame (filled from the oracle, standing in for
the GPU render) in the product's layouts -- rank 0 in place in the whole frame, the others packed rgb | id | t --
exchange them with the product's exchange_packed, and rank 0 puts the received strips in place with the numpy
mirror of mrt_scatter_strips (the GPU suite checks that mirror against the kernel).  Checks the partition, the
staging layout, the exchange and the scatter that bench.py --gpus N relies on; and that bench.py --gpus N started
directly launches its own ranks and hands their failure back."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O
from raytracing_amd import capi, multigpu

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

W, H = 96, 52  # 52 rows = 6.5 strips of 8: uneven counts and a ragged last strip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, strip, full, result_path, root_share=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, shares, span = multigpu.share_partition(rank, world, root_share)
        rows = capi.local_row_index(H, first, shares, strip, span)
        assert len(rows) == capi.local_rows(H, first, shares, strip, span)   # the C-ABI's count, the host mirror's rows
        offsets, total = multigpu.staging_offsets(H, W, world, strip, skip_rank=0, root_share=root_share)
        if rank == 0:
            frame = {"rgb": np.zeros((H, W, 3), np.float32), "id": np.full((H, W), -9, np.int32),
                     "t": np.zeros((H, W), np.float32)}
            for k in frame:
                frame[k][rows] = full[k][rows]          # rendered in place (frame_layout 1)
            staging = torch.zeros(total, dtype=torch.float32)
            multigpu.exchange_packed(None, staging, offsets, rank, world, dst=0)
            multigpu.scatter_strips_host(staging.numpy(), offsets, frame["rgb"], frame["id"], frame["t"], world, strip,
                                         skip_rank=0, root_share=root_share)
            np.savez(result_path, **frame)
        else:
            n = len(rows) * W
            packed = torch.zeros(n * multigpu.PLANE_DWORDS, dtype=torch.float32)
            assert packed.numel() == offsets[rank][1]
            p = packed.numpy()
            p[:n * 3] = full["rgb"][rows].reshape(-1)
            p[n * 3:n * 4] = full["id"][rows].reshape(-1).view(np.float32)
            p[n * 4:] = full["t"][rows].reshape(-1)
            multigpu.exchange_packed(packed, None, offsets, rank, world, dst=0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,strip", [(2, 8), (3, 8), (8, 8), (3, 16)])
def test_exchange_and_scatter_over_gloo(tmp_path, world, strip):
    r = O.Scene.default().render(O.default_viewport(W / H), W, H)
    full = {k: r[k] for k in ("rgb", "id", "t")}
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), strip, full, out), nprocs=world, join=True)
    got = np.load(out)
    for k in full:
        assert np.array_equal(got[k], full[k]), k


@pytest.mark.parametrize("world,strip,root_share", [(2, 8, 3), (3, 8, 2), (4, 8, 4), (3, 16, 2)])
def test_weighted_exchange_and_scatter_over_gloo(tmp_path, world, strip, root_share):
    """Rank 0 takes root_share of world - 1 + root_share shares of the frame (mrt_partition.rank_span, rendered in place
    in one go), every other rank one: same exchange, same scatter, same frame."""
    r = O.Scene.default().render(O.default_viewport(W / H), W, H)
    full = {k: r[k] for k in ("rgb", "id", "t")}
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), strip, full, out, root_share), nprocs=world, join=True)
    got = np.load(out)
    for k in full:
        assert np.array_equal(got[k], full[k]), k


def test_root_share_partition_and_suggestion():
    # every row belongs to exactly one rank, rank 0 holds its shares' strips in ascending order
    for Hh, n, strip, k in ((1080, 8, 8, 57), (52, 3, 8, 2), (2160, 4, 16, 5), (270, 2, 8, 63)):
        seen = np.zeros(Hh, np.int32)
        for r in range(n):
            first, shares, span = multigpu.share_partition(r, n, k)
            rows = capi.local_row_index(Hh, first, shares, strip, span)
            assert len(rows) == capi.local_rows(Hh, first, shares, strip, span) and np.all(np.diff(rows) > 0)
            seen[rows] += 1
        assert np.all(seen == 1)
        assert sum(multigpu.rank_rows(Hh, n, strip, k)) == Hh
    # the default frame: 16 us on one GPU, 41.5 MB to gather -- rank 0 should take nearly everything; a frame of 3 ms
    # and the same bytes should be dealt evenly
    assert multigpu.suggest_root_share(0.016, 41.5e6, 8) >= 40
    assert multigpu.suggest_root_share(0.016, 41.5e6, 2) >= 40
    assert multigpu.suggest_root_share(3.3, 42e6, 8) == 1
    assert multigpu.suggest_root_share(1.0, 1e6, 1) == 1


def test_staging_layout_and_strip_choice():
    for Hh, n, strip in ((1080, 8, 8), (52, 3, 8), (2160, 4, 40), (8, 8, 8), (270, 2, 16)):
        offsets, total = multigpu.staging_offsets(Hh, 64, n, strip, skip_rank=0)
        rows = multigpu.rank_rows(Hh, n, strip)
        assert sum(rows) == Hh and offsets[0] == (0, 0)
        assert total == (Hh - rows[0]) * 64 * multigpu.PLANE_DWORDS
        for r in range(1, n):
            assert offsets[r] == (sum(rows[1:r]) * 64 * 5, rows[r] * 64 * 5)
    for Hh, n in ((1080, 1), (1080, 2), (1080, 8), (2160, 8), (64, 8)):
        s = multigpu.default_strip_rows(Hh, n)
        assert s >= 8 and s % 8 == 0 and (n > 1 or s == 8)
        if n > 1 and Hh >= 8 * n * 6:
            assert 6 <= Hh / (s * n) <= 12    # six strips per rank or a few more (strips are multiples of 8 rows)


def test_bench_started_directly_launches_its_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` with no WORLD_SIZE must start two ranks itself (VERDICT r1 #3a).  Here there is no
    GPU, so both ranks stop at "needs a GPU": the parent must come back with a non-zero status and that message,
    not with "launch with torch.distributed.run"."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU suite runs the 2-rank rehearsal instead")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode != 0
    assert "needs a GPU" in out.stderr + out.stdout
    assert "launch with torch.distributed.run" not in out.stderr + out.stdout
