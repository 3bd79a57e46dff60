"""Multi-GPU frame rendering: one process per GPU (torch.distributed, backend "nccl" = RCCL).

The scene is replicated; the frame is dealt to ranks in strips of rows (strip s -> rank s % N, include/mrt.h
mrt_partition).  Rank 0 renders its own strips straight into the final frame (frame_layout 1); every other rank
renders its strips back to back into ONE packed buffer -- rgb | id | t planes of its rows (or only those of them the caller wants gathered) -- and sends that
buffer to rank 0: one message per peer, each over the peer's own xGMI link, all of them received in one group.  Rank 0 then
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

PLANE_DWORDS = 5  # per pixel with every plane: rgb 3 x f32, id i32, t f32


def plane_dwords(planes=capi.PLANES_ALL):
    """4-byte elements per pixel of a packed buffer that holds `planes` (capi.PLANE_* mask; rgb always)."""
    return 3 + (1 if planes & capi.PLANE_ID else 0) + (1 if planes & capi.PLANE_T else 0)


def share_partition(rank, num_ranks, root_share=1):
    """(first share, shares, span) of a real rank when the frame is dealt in num_ranks - 1 + root_share shares and rank 0
    renders the first root_share of them (include/mrt.h: mrt_gather_strips_root); root_share 1 = the plain partition."""
    shares = num_ranks - 1 + root_share
    return (0, shares, root_share) if rank == 0 else (root_share + rank - 1, shares, 1)


def rank_rows(height, num_ranks, strip_rows=8, root_share=1):
    """Rows every rank renders."""
    out = []
    for r in range(num_ranks):
        first, shares, span = share_partition(r, num_ranks, root_share)
        out.append(capi.local_rows(height, first, shares, strip_rows, span))
    return out


def suggest_root_share(render_ms_one_gpu, frame_bytes, num_ranks, link_gbps=45.0, max_shares=64):
    """How many shares rank 0 should take when frames follow each other (exchange beside the next render): with k of
    V = N - 1 + k shares, rank 0 renders k / V of the frame (render_ms_one_gpu * k / V) while it receives the other
    (N - 1) / V of frame_bytes over its N - 1 links, each peer's 1 / V over that peer's own link at link_gbps
    (xGMI: ~45-50 GB/s achievable per link) -- the frame costs the larger of the two; the smallest k that minimises it."""
    if num_ranks <= 1:
        return 1
    best, best_t = 1, None
    for k in range(1, max_shares - (num_ranks - 1) + 1):
        v = num_ranks - 1 + k
        t = max(render_ms_one_gpu * k / v, frame_bytes / v / (link_gbps * 1e6))
        if best_t is None or t < best_t * 0.98:
            best, best_t = k, t
    return best


def default_strip_rows(height, num_ranks):
    """8 rows (one tile row) on one GPU; with N GPUs about six strips per rank -- enough to deal sky and geometry
    rows evenly, few enough that a strip is a large contiguous block."""
    if num_ranks <= 1:
        return 8
    return max(8, (height // (num_ranks * 6)) // 8 * 8)


def staging_offsets(height, width, num_ranks, strip_rows=8, skip_rank=0, planes=capi.PLANES_ALL, root_share=1):
    """(offset, length) in 4-byte elements of every rank's packed buffer in rank 0's staging area (the layout
    mrt_scatter_strips reads); skip_rank has length 0."""
    out, off = [], 0
    for r, rows in enumerate(rank_rows(height, num_ranks, strip_rows, root_share)):
        n = 0 if r == skip_rank else rows * width * plane_dwords(planes)
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


def scatter_strips_host(staging, offsets, rgb, ids, tt, num_ranks, strip_rows=8, skip_rank=0, planes=capi.PLANES_ALL,
                        root_share=1):
    """numpy mirror of mrt_scatter_strips(_planes / _span) (CPU tests of the N > 1 path): staging is a flat float32
    array, the frame planes are [H, W, 3] float32, [H, W] int32, [H, W] float32; a plane that does not travel is left
    alone.  num_ranks: REAL ranks; with root_share > 1 rank r's rows are those of its share (share_partition)."""
    H, W = rgb.shape[:2]
    for r in range(num_ranks):
        if r == skip_rank:
            continue
        first, shares, span = share_partition(r, num_ranks, root_share)
        rows = capi.local_row_index(H, first, shares, strip_rows, span)
        n, (o, _) = len(rows), offsets[r]
        rgb[rows] = staging[o:o + n * W * 3].reshape(n, W, 3)
        o += n * W * 3
        if planes & capi.PLANE_ID:
            ids[rows] = staging[o:o + n * W].view(np.int32).reshape(n, W)
            o += n * W
        if planes & capi.PLANE_T:
            tt[rows] = staging[o:o + n * W].reshape(n, W)


class FrameRenderer:
    """Renders frames of one scene on this rank's GPU and, for N > 1, collects them on rank 0.

    planes: what travels to rank 0 and what the ranks render for it -- rgb always, id and t on request (the picture
    alone is 12 of the 20 bytes per pixel; at N = 1 all three planes are always written).
    pipeline: frames in flight at N > 1.  1: render, then exchange, on one stream.  2 (default): two sets of buffers;
    the exchange (+ rank 0's scatter) of frame k runs on a side stream while frame k + 1 renders, with events between
    the two streams, so a stream of frames costs max(render, exchange) per frame, not the sum.  self.rgb / ids / tt
    are the planes of the frame the last step() produced (complete once both streams have been synchronised).
    root_share: shares of the frame rank 0 renders when it is dealt in num_ranks - 1 + root_share shares (1: every
    rank the same; more where the exchange, not the render, bounds the frame: suggest_root_share)."""

    def __init__(self, flat_scene, cam12, width, height, rank=0, num_ranks=1, device=0, strip_rows=None, spp=1,
                 max_depth=0, planes=capi.PLANES_ALL, pipeline=2, root_share=1):
        import torch

        self.torch = torch
        self.rank, self.num_ranks = rank, num_ranks
        self.root_share = 1 if num_ranks == 1 else max(1, min(int(root_share), 64 - (num_ranks - 1)))
        self.part = share_partition(rank, num_ranks, self.root_share)   # (first share, shares, span)
        self.strip_rows = strip_rows or default_strip_rows(height, self.part[1])
        self.W, self.H, self.spp, self.max_depth = width, height, spp, max_depth
        self.planes = capi.PLANES_ALL if num_ranks == 1 else (int(planes) | capi.PLANE_RGB)
        self.depth = 1 if num_ranks == 1 else max(1, min(2, int(pipeline)))
        self.cam12 = np.ascontiguousarray(cam12, dtype=np.float32)
        self.dev = torch.device(f"cuda:{device}")
        torch.cuda.set_device(self.dev)
        self.ctx = capi.Context(device)
        self.ctx.upload(flat_scene)
        self.local_rows = capi.local_rows(height, self.part[0], self.part[1], self.strip_rows, self.part[2])
        self.stats = torch.zeros(8, dtype=torch.int64, device=self.dev)
        self.offsets, total = staging_offsets(height, width, num_ranks, self.strip_rows, skip_rank=0, planes=self.planes,
                                              root_share=self.root_share)
        has_id, has_t = bool(self.planes & capi.PLANE_ID), bool(self.planes & capi.PLANE_T)
        self.slots = []  # per frame in flight: dict(buf, rgb, ids, tt, staging)
        for _ in range(self.depth):
            if rank == 0:
                # whole frame; with N > 1 only rank 0's own rows are rendered here, the rest arrives
                n = height * width
                buf = torch.zeros(n * PLANE_DWORDS, dtype=torch.float32, device=self.dev)
                sl = {"buf": buf, "rgb": buf[:n * 3], "ids": buf[n * 3:n * 4].view(torch.int32), "tt": buf[n * 4:],
                      "staging": torch.empty(total, dtype=torch.float32, device=self.dev) if num_ranks > 1 else None}
                if not has_id:
                    sl["ids"] = None
                if not has_t:
                    sl["tt"] = None
            else:
                # this rank's rows back to back: rgb | id (if it travels) | t (if it travels)
                n = self.local_rows * width
                buf = torch.zeros(n * plane_dwords(self.planes), dtype=torch.float32, device=self.dev)
                o = n * 3
                sl = {"buf": buf, "rgb": buf[:o], "ids": None, "tt": None, "staging": None}
                if has_id:
                    sl["ids"] = buf[o:o + n].view(torch.int32)
                    o += n
                if has_t:
                    sl["tt"] = buf[o:o + n]
            self.slots.append(sl)
        self.layout = 1 if (rank == 0 and num_ranks > 1) else 0
        self.frames = 0      # frames started
        self._calls = {}
        self.comm = None
        self.xstream = None
        if self.depth > 1:
            self.xstream = torch.cuda.Stream(device=self.dev)
            self.ev_rendered = [torch.cuda.Event() for _ in range(self.depth)]
            self.ev_free = [torch.cuda.Event() for _ in range(self.depth)]
            self._used = [False] * self.depth

    # the planes of the frame the last step() / render_local() produced
    def _last(self):
        return self.slots[(self.frames - 1) % self.depth if self.frames else 0]

    rgb = property(lambda self: self._last()["rgb"])
    ids = property(lambda self: self._last()["ids"])
    tt = property(lambda self: self._last()["tt"])
    packed = property(lambda self: None if self.rank == 0 else self._last()["buf"])
    staging = property(lambda self: self._last()["staging"])

    def describe(self):
        """What is switched on, for the bench line."""
        names = ["rgb"] + (["id"] if self.planes & capi.PLANE_ID else []) + (["t"] if self.planes & capi.PLANE_T else [])
        return {"pipeline_depth": self.depth, "planes_gathered": "+".join(names),
                "bytes_per_pixel_gathered": 4 * plane_dwords(self.planes), "weighted_partition": self.root_share > 1,
                "root_share": f"{self.root_share} of {self.part[1]} shares" if self.num_ranks > 1 else None}

    def init_comm(self, unique_id):
        """Collective over the ranks: from now on the exchange step goes through mrt_gather_strips (RCCL behind the
        C-ABI).  unique_id: the 128 bytes of capi.comm_unique_id() made on rank 0."""
        self.comm = capi.Comm(self.ctx, unique_id, self.rank, self.num_ranks)

    def drop_comm(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    def _render_slot(self, s, want_stats=False):
        import ctypes as C
        if s not in self._calls:  # argument structs are built once: a frame is tens of microseconds
            sl = self.slots[s]
            cam = capi.Camera.from12(self.cam12)
            par = capi.Context._params(self.W, self.H, self.spp, self.max_depth, self.part[0], self.part[1],
                                       self.strip_rows, self.layout, self.part[2])
            ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
            self._calls[s] = (cam, par, C.byref(cam), C.byref(par), ptr(sl["rgb"]), ptr(sl["ids"]), ptr(sl["tt"]),
                              C.c_void_p(self.stats.data_ptr()))
        _, _, cam_p, par_p, rgb_p, id_p, t_p, st_p = self._calls[s]
        stream = C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)
        rc = self.ctx._L.mrt_render_device(self.ctx._h, cam_p, par_p, rgb_p, id_p, t_p, st_p if want_stats else None,
                                           stream)
        self.ctx._check(rc, "mrt_render_device")

    def render_local(self, want_stats=False):
        """Enqueue this rank's strips on torch's current stream (asynchronous), into the next set of buffers."""
        s = self.frames % self.depth
        self.frames += 1
        self._render_slot(s, want_stats)
        return s

    def collect(self, s=None):
        """The exchange step + rank 0's scatter of the frame in buffer set s (asynchronous on the current stream)."""
        if self.num_ranks == 1:
            return
        sl = self.slots[(self.frames - 1) % self.depth if s is None else s]
        ptr = lambda x: x.data_ptr() if x is not None else None
        stream = self.torch.cuda.current_stream(self.dev).cuda_stream
        if self.comm is not None:
            if self.rank == 0:
                self.ctx.gather_strips(self.comm, None, ptr(sl["staging"]), ptr(sl["rgb"]), ptr(sl["ids"]), ptr(sl["tt"]),
                                       self.W, self.H, self.strip_rows, root_in_place=True, stream=stream,
                                       planes=self.planes, root_span=self.root_share)
            else:
                self.ctx.gather_strips(self.comm, ptr(sl["buf"]), None, None, None, None, self.W, self.H,
                                       self.strip_rows, root_in_place=True, stream=stream, planes=self.planes,
                                       root_span=self.root_share)
            return
        exchange_packed(sl["buf"] if self.rank else None, sl["staging"], self.offsets, self.rank, self.num_ranks, dst=0)
        if self.rank == 0:
            self.ctx.scatter_strips(ptr(sl["staging"]), ptr(sl["rgb"]), ptr(sl["ids"]), ptr(sl["tt"]), self.W, self.H,
                                    self.part[1], self.strip_rows, skip_rank=0, stream=stream, planes=self.planes,
                                    skip_span=self.root_share)

    def step(self):
        """One frame: render local strips, send them to rank 0, rank 0 puts them in place.  Asynchronous; with
        pipeline 2 the exchange goes to the side stream and the next step() renders beside it."""
        if self.depth == 1:
            self.render_local()
            self.collect()
            return
        torch = self.torch
        cur = torch.cuda.current_stream(self.dev)
        s = self.frames % self.depth
        if self._used[s]:
            cur.wait_event(self.ev_free[s])     # the exchange that last read (or filled) this set of buffers is over
        self.render_local()
        self.ev_rendered[s].record(cur)
        self.xstream.wait_event(self.ev_rendered[s])
        with torch.cuda.stream(self.xstream):
            self.collect(s)
        self.ev_free[s].record(self.xstream)
        self._used[s] = True

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
        out = {"rgb": self.rgb.view(self.H, self.W, 3).cpu().numpy()}
        if self.ids is not None:
            out["id"] = self.ids.view(self.H, self.W).cpu().numpy()
        if self.tt is not None:
            out["t"] = self.tt.view(self.H, self.W).cpu().numpy()
        return out

    def close(self):
        self.torch.cuda.synchronize(self.dev)
        self.drop_comm()
        self.ctx.close()
