"""The N > 1 path on CPU: two gloo ranks each produce their strips of a frame (filled from the
oracle, standing in for the GPU render), gather them to rank 0 with the product's
gather_strips, and rank 0 interleaves them back with the strip rule.  Checks the partition,
padding, gather order and assembly that bench.py --gpus N relies on."""
import os
import socket

import numpy as np
import pytest

import _oracle as O
from raytracing_amd import capi, multigpu

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

W, H = 96, 52  # 52 rows = 6.5 strips: uneven counts and a ragged last strip


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, full_rgb, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = capi.local_row_index(H, rank, world)
        pad = multigpu.padded_rows(H, world)
        local = torch.zeros(pad * W * 3, dtype=torch.float32)
        local[: len(rows) * W * 3] = torch.from_numpy(full_rgb[rows].reshape(-1))
        gathered = multigpu.gather_strips(local, world, rank, dst=0)
        if rank == 0:
            frame = np.zeros((H, W, 3), dtype=np.float32)
            g = gathered.numpy().reshape(world, pad, W, 3)
            for r in range(world):
                rr = capi.local_row_index(H, r, world)
                frame[rr] = g[r, : len(rr)]
            np.save(result_path, frame)
        else:
            assert gathered is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_and_assemble_over_gloo(tmp_path, world):
    full = O.Scene.default().render(O.default_viewport(W / H), W, H)["rgb"]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), full, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), full)


def test_padded_rows_is_the_maximum_over_ranks():
    for Hh, n in ((1080, 8), (52, 3), (2160, 4), (8, 8)):
        assert multigpu.padded_rows(Hh, n) == max(capi.local_rows(Hh, r, n) for r in range(n))
