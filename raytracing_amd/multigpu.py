"""Multi-GPU frame rendering: one process per GPU (torch.distributed, backend "nccl" = RCCL).

The scene is replicated; the frame is dealt to ranks in strips of rows (strip s -> rank s % N, include/mrt.h
mrt_partition).  Rank 0 renders its own strips straight into the final frame (frame_layout 1); every other rank
renders its strips back to back into ONE packed buffer -- rgb | id | t planes of its rows -- and sends that buffer
to rank 0: one message per peer, each over the peer's own xGMI link, all of them received in one group.  Rank 0 then
copies the peers' strips from the staging area to their places in the three frame planes (mrt_scatter_strips: one
pass over the (N-1)/N of the frame that came from elsewhere; its own rows are never copied).  There is no other
exchange step and no reference counterpart: the reference is single-GPU.

The exchange step has two carriers: mrt_gather_strips -- RCCL behind the C-ABI, on a communicator made from an id
the ranks pass around (FrameRenderer.init_comm) -- and torch.distributed's point-to-point calls (RCCL too, under
backend "nccl"; gloo for the CPU tests and rehearsals).  Same messages, same staging layout, same scatter kernel.
torch is otherwise used for device memory and streams only.
"""
import numpy as np

from . import capi

PLANE_DWORDS = 5  # per pixel: rgb 3 x f32, id i32, t f32


def rank_rows(height, num_ranks, strip_rows=8):
    """Rows every rank renders."""
    return [capi.local_rows(height, r, num_ranks, strip_rows) for r in range(num_ranks)]


def default_strip_rows(height, num_ranks):
    """8 rows (one tile row) on one GPU; with N GPUs about six strips per rank -- enough to deal sky and geometry
    rows evenly, few enough that a strip is a large contiguous block."""
    if num_ranks <= 1:
        return 8
    return max(8, (height // (num_ranks * 6)) // 8 * 8)


def staging_offsets(height, width, num_ranks, strip_rows=8, skip_rank=0):
    """(offset, length) in 4-byte elements of every rank's packed buffer in rank 0's staging area (the layout
    mrt_scatter_strips reads); skip_rank has length 0."""
    out, off = [], 0
    for r, rows in enumerate(rank_rows(height, num_ranks, strip_rows)):
        n = 0 if r == skip_rank else rows * width * PLANE_DWORDS
        out.append((off, n))
        off += n
    return out, off


def exchange_packed(packed, staging, offsets, rank, num_ranks, dst=0, group=None):
    """The one exchange step: every rank but dst sends its packed buffer, dst receives them all at their offsets
    in `staging` (one grouped batch of point-to-point transfers).  Works for nccl on GPU tensors and gloo on CPU
    tensors; GPU tensors under gloo (rehearsal on a box with fewer GPUs than ranks) are staged through the host."""
    import torch
    import torch.distributed as dist

    if num_ranks == 1:
        return
    backend = dist.get_backend(group)
    via_host = backend == "gloo" and (packed.is_cuda if rank != dst else staging.is_cuda)
    if rank == dst:
        buf = torch.empty(staging.shape, dtype=staging.dtype) if via_host else staging
        ops = [dist.P2POp(dist.irecv, buf[o:o + n], r, group) for r, (o, n) in enumerate(offsets) if r != dst and n]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if via_host:
            staging.copy_(buf)
    else:
        src = packed.cpu() if via_host else packed
        if src.numel():
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, dst, group)]):
                w.wait()


def scatter_strips_host(staging, offsets, rgb, ids, tt, num_ranks, strip_rows=8, skip_rank=0):
    """numpy mirror of mrt_scatter_strips (CPU tests of the N > 1 path): staging is a flat float32 array, the
    frame planes are [H, W, 3] float32, [H, W] int32, [H, W] float32."""
    H, W = ids.shape
    for r in range(num_ranks):
        if r == skip_rank:
            continue
        rows = capi.local_row_index(H, r, num_ranks, strip_rows)
        n, (o, _) = len(rows), offsets[r]
        rgb[rows] = staging[o:o + n * W * 3].reshape(n, W, 3)
        ids[rows] = staging[o + n * W * 3:o + n * W * 4].view(np.int32).reshape(n, W)
        tt[rows] = staging[o + n * W * 4:o + n * W * 5].reshape(n, W)


class FrameRenderer:
    """Renders frames of one scene on this rank's GPU and, for N > 1, collects them on rank 0."""

    def __init__(self, flat_scene, cam12, width, height, rank=0, num_ranks=1, device=0, strip_rows=None, spp=1,
                 max_depth=0):
        import torch

        self.torch = torch
        self.rank, self.num_ranks = rank, num_ranks
        self.strip_rows = strip_rows or default_strip_rows(height, num_ranks)
        self.W, self.H, self.spp, self.max_depth = width, height, spp, max_depth
        self.cam12 = np.ascontiguousarray(cam12, dtype=np.float32)
        self.dev = torch.device(f"cuda:{device}")
        torch.cuda.set_device(self.dev)
        self.ctx = capi.Context(device)
        self.ctx.upload(flat_scene)
        self.local_rows = capi.local_rows(height, rank, num_ranks, self.strip_rows)
        self.stats = torch.zeros(8, dtype=torch.int64, device=self.dev)
        self.offsets, total = staging_offsets(height, width, num_ranks, self.strip_rows, skip_rank=0)
        self.packed = self.staging = None
        if rank == 0:
            # whole frame; with N > 1 only rank 0's own rows are rendered here, the rest arrives
            n = height * width
            self.frame = torch.zeros(n * PLANE_DWORDS, dtype=torch.float32, device=self.dev)
            self.rgb, self.ids, self.tt = self.frame[:n * 3], self.frame[n * 3:n * 4].view(torch.int32), self.frame[n * 4:]
            self.layout = 1 if num_ranks > 1 else 0
            if num_ranks > 1:
                self.staging = torch.empty(total, dtype=torch.float32, device=self.dev)
        else:
            n = self.local_rows * width
            self.packed = torch.zeros(n * PLANE_DWORDS, dtype=torch.float32, device=self.dev)
            self.rgb, self.ids, self.tt = self.packed[:n * 3], self.packed[n * 3:n * 4].view(torch.int32), self.packed[n * 4:]
            self.layout = 0
        self._call = None
        self.comm = None

    def init_comm(self, unique_id):
        """Collective over the ranks: from now on the exchange step goes through mrt_gather_strips (RCCL behind the
        C-ABI).  unique_id: the 128 bytes of capi.comm_unique_id() made on rank 0."""
        self.comm = capi.Comm(self.ctx, unique_id, self.rank, self.num_ranks)

    def drop_comm(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    def render_local(self, want_stats=False):
        """Enqueue this rank's strips on torch's current stream (asynchronous)."""
        import ctypes as C
        if self._call is None:  # argument structs are built once: a frame is tens of microseconds
            cam = capi.Camera.from12(self.cam12)
            par = capi.Context._params(self.W, self.H, self.spp, self.max_depth, self.rank, self.num_ranks,
                                       self.strip_rows, self.layout)
            self._call = (cam, par, C.byref(cam), C.byref(par), C.c_void_p(self.rgb.data_ptr()),
                          C.c_void_p(self.ids.data_ptr()), C.c_void_p(self.tt.data_ptr()),
                          C.c_void_p(self.stats.data_ptr()))
        _, _, cam_p, par_p, rgb_p, id_p, t_p, st_p = self._call
        stream = C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)
        rc = self.ctx._L.mrt_render_device(self.ctx._h, cam_p, par_p, rgb_p, id_p, t_p, st_p if want_stats else None,
                                           stream)
        self.ctx._check(rc, "mrt_render_device")

    def collect(self):
        """The exchange step + rank 0's scatter (asynchronous on the current stream)."""
        if self.num_ranks == 1:
            return
        if self.comm is not None:
            stream = self.torch.cuda.current_stream(self.dev).cuda_stream
            if self.rank == 0:
                self.ctx.gather_strips(self.comm, None, self.staging.data_ptr(), self.rgb.data_ptr(), self.ids.data_ptr(),
                                       self.tt.data_ptr(), self.W, self.H, self.strip_rows, root_in_place=True,
                                       stream=stream)
            else:
                self.ctx.gather_strips(self.comm, self.packed.data_ptr(), None, None, None, None, self.W, self.H,
                                       self.strip_rows, root_in_place=True, stream=stream)
            return
        exchange_packed(self.packed, self.staging, self.offsets, self.rank, self.num_ranks, dst=0)
        if self.rank == 0:
            stream = self.torch.cuda.current_stream(self.dev).cuda_stream
            self.ctx.scatter_strips(self.staging.data_ptr(), self.rgb.data_ptr(), self.ids.data_ptr(), self.tt.data_ptr(),
                                    self.W, self.H, self.num_ranks, self.strip_rows, skip_rank=0, stream=stream)

    def step(self):
        """One frame: render local strips, send them to rank 0, rank 0 puts them in place.  Asynchronous."""
        self.render_local()
        self.collect()

    def count_rays(self):
        """Blocking: primary/shadow/hit/occluded counts of this rank's strips."""
        self.render_local(want_stats=True)
        self.torch.cuda.synchronize(self.dev)
        s = self.stats.cpu().numpy()
        return {"primary_rays": int(s[0]), "shadow_rays": int(s[1]), "hit_pixels": int(s[2]), "occluded": int(s[3]),
                "exact_tests": int(s[4]), "clusters_opened": int(s[5])}

    def frame_host(self):
        """Rank 0: the assembled frame as numpy arrays {rgb [H, W, 3], id [H, W], t [H, W]}; None elsewhere."""
        self.torch.cuda.synchronize(self.dev)
        if self.rank != 0:
            return None
        return {"rgb": self.rgb.view(self.H, self.W, 3).cpu().numpy(), "id": self.ids.view(self.H, self.W).cpu().numpy(),
                "t": self.tt.view(self.H, self.W).cpu().numpy()}

    def close(self):
        self.drop_comm()
        self.ctx.close()
