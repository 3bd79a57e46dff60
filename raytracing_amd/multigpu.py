"""Multi-GPU frame rendering: one process per GPU (torch.distributed, backend "nccl" = RCCL).

The scene is replicated; the frame is dealt to ranks in strips of 8 rows (strip s -> rank
s % N, include/mrt.h mrt_partition); every rank renders its strips into a compact local buffer;
the one exchange step is a gather of those buffers to rank 0 over xGMI, followed by a small
assembly kernel that interleaves the strips back into a row-major frame.  No reference
counterpart: the reference is single-GPU.

torch is used for device memory, streams and the collective only.
"""
import numpy as np

from . import capi


def padded_rows(height, num_ranks, strip_rows=8):
    """Rows of the largest local buffer (rank 0 always holds the most)."""
    return capi.local_rows(height, 0, num_ranks, strip_rows)


def gather_strips(local, num_ranks, rank, dst=0, group=None, out=None):
    """Gather equally-sized (padded) per-rank buffers to `dst`.

    local: 1-D tensor of padded_rows*W*C elements, the rank's strips back to back.
    Returns a [num_ranks, len(local)] tensor on dst (None elsewhere).  Works for any backend
    (nccl on GPUs, gloo on CPU tensors)."""
    import torch
    import torch.distributed as dist

    if num_ranks == 1:
        return local.view(1, -1)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the N > 1 path on a box without N GPUs (bench.py MRT_BENCH_BACKEND=gloo): stage through
        # the host; the product path is the RCCL gather below
        host = gather_strips(local.cpu(), num_ranks, rank, dst, group)
        if rank != dst:
            return None
        if out is None:
            out = torch.empty((num_ranks, local.numel()), dtype=local.dtype, device=local.device)
        out.copy_(host)
        return out
    if rank == dst:
        if out is None:
            out = torch.empty((num_ranks, local.numel()), dtype=local.dtype, device=local.device)
        dist.gather(local, gather_list=[out[r] for r in range(num_ranks)], dst=dst, group=group)
        return out
    dist.gather(local, gather_list=None, dst=dst, group=group)
    return None


class FrameRenderer:
    """Renders frames of one scene on this rank's GPU and, for N > 1, gathers them to rank 0."""

    def __init__(self, flat_scene, cam12, width, height, rank=0, num_ranks=1, device=0, strip_rows=8, spp=1,
                 with_ids=False, max_depth=0):
        import torch

        self.torch = torch
        self.rank, self.num_ranks, self.strip_rows = rank, num_ranks, strip_rows
        self.W, self.H, self.spp, self.max_depth = width, height, spp, max_depth
        self.cam12 = np.ascontiguousarray(cam12, dtype=np.float32)
        self.dev = torch.device(f"cuda:{device}")
        torch.cuda.set_device(self.dev)
        self.ctx = capi.Context(device)
        self.ctx.upload(flat_scene)
        self.local_rows = capi.local_rows(height, rank, num_ranks, strip_rows)
        self.pad_rows = padded_rows(height, num_ranks, strip_rows)
        n = self.pad_rows * width
        self.rgb = torch.zeros(n * 3, dtype=torch.float32, device=self.dev)
        self.ids = torch.zeros(n, dtype=torch.int32, device=self.dev) if with_ids else None
        self.tt = torch.zeros(n, dtype=torch.float32, device=self.dev) if with_ids else None
        self.stats = torch.zeros(8, dtype=torch.int64, device=self.dev)
        self.gathered = None
        self.frame = None
        self._call = None
        if rank == 0 and num_ranks > 1:
            self.gathered = torch.empty((num_ranks, n * 3), dtype=torch.float32, device=self.dev)
            self.frame = torch.empty((height, width, 3), dtype=torch.float32, device=self.dev)

    def render_local(self, want_stats=False):
        """Enqueue this rank's strips on torch's current stream (asynchronous)."""
        import ctypes as C
        if self._call is None:  # argument structs are built once: a frame is tens of microseconds
            cam = capi.Camera.from12(self.cam12)
            par = capi.Context._params(self.W, self.H, self.spp, self.max_depth, self.rank, self.num_ranks, self.strip_rows)
            self._call = (cam, par, C.byref(cam), C.byref(par), C.c_void_p(self.rgb.data_ptr()),
                          C.c_void_p(self.ids.data_ptr()) if self.ids is not None else None,
                          C.c_void_p(self.tt.data_ptr()) if self.tt is not None else None,
                          C.c_void_p(self.stats.data_ptr()))
        _, _, cam_p, par_p, rgb_p, id_p, t_p, st_p = self._call
        stream = C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)
        rc = self.ctx._L.mrt_render_device(self.ctx._h, cam_p, par_p, rgb_p, id_p, t_p, st_p if want_stats else None,
                                           stream)
        self.ctx._check(rc, "mrt_render_device")

    def step(self):
        """One frame: render local strips, gather to rank 0, assemble.  Asynchronous."""
        self.render_local()
        if self.num_ranks > 1:
            gather_strips(self.rgb, self.num_ranks, self.rank, dst=0, out=self.gathered)
            if self.rank == 0:
                stream = self.torch.cuda.current_stream(self.dev).cuda_stream
                self.ctx.assemble_strips(self.gathered.data_ptr(), self.gathered.shape[1], self.frame.data_ptr(),
                                         self.W, self.H, 3, self.num_ranks, self.strip_rows, stream=stream)

    def count_rays(self):
        """Blocking: primary/shadow/hit/occluded counts of this rank's strips."""
        self.render_local(want_stats=True)
        self.torch.cuda.synchronize(self.dev)
        s = self.stats.cpu().numpy()
        return {"primary_rays": int(s[0]), "shadow_rays": int(s[1]), "hit_pixels": int(s[2]), "occluded": int(s[3]),
                "exact_tests": int(s[4]), "clusters_opened": int(s[5])}

    def frame_host(self):
        """Rank 0: the assembled frame as a numpy array [H, W, 3]."""
        self.torch.cuda.synchronize(self.dev)
        if self.num_ranks == 1:
            return self.rgb.view(self.pad_rows, self.W, 3)[: self.H].cpu().numpy()
        return self.frame.cpu().numpy() if self.rank == 0 else None

    def close(self):
        self.ctx.close()
