"""The multi-GPU entry points of the C-ABI (mrt_comm_*, mrt_gather_strips, mrt_render_multi) on the ONE GPU a test
box has.  What can run there runs through RCCL itself: a communicator of one rank, whose packed buffer travels as a
send to itself and a receive (root_in_place = 0), then mrt_scatter_strips -- the same calls, offsets and kernels the
N-rank frame uses.  More than one rank needs more than one GPU (RCCL refuses two ranks on one device); the N > 1
layout logic is covered by the strip tests of test_gpu_parity.py and the gloo tests of test_distributed_cpu.py."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import capi, scenes

pytestmark = pytest.mark.gpu


def _frame_buffers(torch, W, H):
    n = W * H
    frame = torch.full((n * 5,), -3.0, dtype=torch.float32, device="cuda:0")
    return frame, frame[:n * 3], frame[n * 3:n * 4].view(torch.int32), frame[n * 4:]


@pytest.mark.parametrize("W,H,strip", [(320, 200, 8), (250, 130, 16)])
def test_gather_strips_single_rank_through_rccl(gpu_ctx, W, H, strip):
    torch = pytest.importorskip("torch")
    flat, _ = scenes.default_scene()
    gpu_ctx.upload(flat)
    cam = capi.default_viewport(W / H)
    full = gpu_ctx.render(cam, W, H)
    comm = capi.Comm(gpu_ctx, capi.comm_unique_id(), 0, 1)
    try:
        n = W * H
        _, rgb, ids, tt = _frame_buffers(torch, W, H)
        packed = torch.zeros(n * 5, dtype=torch.float32, device="cuda:0")
        staging = torch.zeros(n * 5, dtype=torch.float32, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):  # the second frame reuses the connection the first one set up
            gpu_ctx.render_device(cam, W, H, packed.data_ptr(), id_ptr=packed[n * 3:].data_ptr(),
                                  t_ptr=packed[n * 4:].data_ptr(), stream=stream, strip_rows=strip)
            gpu_ctx.gather_strips(comm, packed.data_ptr(), staging.data_ptr(), rgb.data_ptr(), ids.data_ptr(),
                                  tt.data_ptr(), W, H, strip_rows=strip, root_in_place=False, stream=stream)
            torch.cuda.synchronize()
            assert np.array_equal(rgb.view(H, W, 3).cpu().numpy(), full["rgb"])
            assert np.array_equal(ids.view(H, W).cpu().numpy(), full["id"])
            assert np.array_equal(tt.view(H, W).cpu().numpy(), full["t"])
            rgb.fill_(-3.0)
        # rank 0 rendered in place and nobody else sends: nothing to do, and no staging area is needed
        gpu_ctx.render_device(cam, W, H, rgb.data_ptr(), id_ptr=ids.data_ptr(), t_ptr=tt.data_ptr(), stream=stream,
                              strip_rows=strip, frame_layout=1)
        gpu_ctx.gather_strips(comm, None, None, rgb.data_ptr(), ids.data_ptr(), tt.data_ptr(), W, H, strip_rows=strip,
                              root_in_place=True, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(rgb.view(H, W, 3).cpu().numpy(), full["rgb"])
    finally:
        comm.close()


@pytest.mark.parametrize("planes", [capi.PLANE_RGB, capi.PLANE_RGB | capi.PLANE_T, capi.PLANE_RGB | capi.PLANE_ID])
def test_gather_strips_planes_single_rank_through_rccl(gpu_ctx, planes):
    """Only some planes travel (mrt_gather_strips_planes): the packed buffer holds rgb, then id / t if gathered; the
    planes that do not travel keep what the frame buffers held."""
    torch = pytest.importorskip("torch")
    W, H, strip = 200, 136, 16
    flat, _ = scenes.default_scene()
    gpu_ctx.upload(flat)
    cam = capi.default_viewport(W / H)
    full = gpu_ctx.render(cam, W, H)
    comm = capi.Comm(gpu_ctx, capi.comm_unique_id(), 0, 1)
    try:
        n = W * H
        _, rgb, ids, tt = _frame_buffers(torch, W, H)
        ids.fill_(-7)
        has_id, has_t = bool(planes & capi.PLANE_ID), bool(planes & capi.PLANE_T)
        nd = 3 + has_id + has_t
        packed = torch.zeros(n * nd, dtype=torch.float32, device="cuda:0")
        staging = torch.zeros(n * nd, dtype=torch.float32, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        gpu_ctx.render_device(cam, W, H, packed.data_ptr(), id_ptr=packed[n * 3:].data_ptr() if has_id else None,
                              t_ptr=packed[n * (3 + has_id):].data_ptr() if has_t else None, stream=stream,
                              strip_rows=strip)
        gpu_ctx.gather_strips(comm, packed.data_ptr(), staging.data_ptr(), rgb.data_ptr(), ids.data_ptr(),
                              tt.data_ptr(), W, H, strip_rows=strip, root_in_place=False, stream=stream, planes=planes)
        torch.cuda.synchronize()
        assert np.array_equal(rgb.view(H, W, 3).cpu().numpy(), full["rgb"])
        got_id, got_t = ids.view(H, W).cpu().numpy(), tt.view(H, W).cpu().numpy()
        assert np.array_equal(got_id, full["id"]) if has_id else bool((got_id == -7).all())
        assert np.array_equal(got_t, full["t"]) if has_t else bool((got_t == -3.0).all())
        L = capi.libmrt()
        assert L.mrt_gather_strips_planes(gpu_ctx._h, comm._h, None, None, C.c_void_p(256), None, None, 64, 64, 8, 1, 6,
                                          None) == -1  # rgb always travels
        assert L.mrt_scatter_strips_planes(gpu_ctx._h, C.c_void_p(256), C.c_void_p(256), None, None, 64, 64, 2, 8, 0, 8,
                                           None) == -1
    finally:
        comm.close()


def test_render_multi_picture_only(gpu_ctx):
    """mrt_render_multi with NULL id / t: nothing but the picture is rendered for and copied to the caller."""
    flat, _ = scenes.default_scene()
    gpu_ctx.upload(flat)
    cam = capi.default_viewport(2.0)
    full = gpu_ctx.render(cam, 256, 128)
    out = capi.render_multi([gpu_ctx], cam, 256, 128, planes=capi.PLANE_RGB)
    assert np.array_equal(out["rgb"], full["rgb"]) and bool((out["id"] == -2).all())


def test_comm_outlives_its_context():
    """Either destroy order is allowed (include/mrt.h): the communicator keeps the GPU ordinal itself."""
    ctx = capi.Context(0)
    comm = capi.Comm(ctx, capi.comm_unique_id(), 0, 1)
    ctx.close()
    comm.close()


def test_gather_strips_argument_checks(gpu_ctx):
    L = capi.libmrt()
    comm = capi.Comm(gpu_ctx, capi.comm_unique_id(), 0, 1)
    other = capi.Context(0)
    try:
        # a communicator belongs to its context
        assert L.mrt_gather_strips(other._h, comm._h, None, None, None, None, None, 64, 64, 8, 1, None) == -1
        # rank 0 needs frame buffers; strips are multiples of 8 rows
        assert L.mrt_gather_strips(gpu_ctx._h, comm._h, None, None, None, None, None, 64, 64, 8, 1, None) == -1
        assert L.mrt_gather_strips(gpu_ctx._h, comm._h, None, None, C.c_void_p(256), None, None, 64, 64, 12, 1, None) == -1
        h = C.c_void_p()
        uid = (C.c_uint8 * 128)()
        assert L.mrt_comm_create(gpu_ctx._h, uid, 3, 2, C.byref(h)) == -1   # rank out of range
        assert L.mrt_comm_create(gpu_ctx._h, uid, 0, 65, C.byref(h)) == -1  # more than 64 ranks
        # two contexts on one GPU cannot form a local group
        hs = (C.c_void_p * 2)(gpu_ctx._h, other._h)
        out = (C.c_void_p * 2)()
        assert L.mrt_comm_create_local(hs, 2, out) == -1
        assert b"one GPU" in L.mrt_last_error(gpu_ctx._h)
    finally:
        other.close()
        comm.close()


@pytest.mark.parametrize("scene", ["default", "spheres", "default_d2"])
def test_render_multi_one_context_matches_render(gpu_ctx, scene):
    W, H = 200, 120
    depth = 0
    if scene == "spheres":
        spec = scenes.sphere_field(300)
        flat, cam = spec.flatten(), spec.cam12(W / H)
    else:
        flat, _ = scenes.default_scene()
        cam = capi.default_viewport(W / H)
        depth = 2 if scene == "default_d2" else 0
    gpu_ctx.upload(flat)
    full = gpu_ctx.render(cam, W, H, max_depth=depth)
    # without communicators, and with the group of one that mrt_comm_create_local makes
    out = capi.render_multi([gpu_ctx], cam, W, H, max_depth=depth)
    L = capi.libmrt()
    hs, comms = (C.c_void_p * 1)(gpu_ctx._h), (C.c_void_p * 1)()
    assert L.mrt_comm_create_local(hs, 1, comms) == 0
    try:
        out2 = capi.render_multi([gpu_ctx], cam, W, H, max_depth=depth, comms=comms, strip_rows=16)
    finally:
        L.mrt_comm_destroy(comms[0])
    for o in (out, out2):
        for k in ("rgb", "id", "t"):
            assert np.array_equal(o[k], full[k]), k
        for k in ("primary_rays", "shadow_rays", "hit_pixels", "occluded"):
            assert o["stats"][k] == full["stats"][k]
    # the frame goes on living in the context's buffer: a plain render afterwards must not be disturbed by it
    again = gpu_ctx.render(cam, W, H, max_depth=depth)
    assert np.array_equal(again["rgb"], full["rgb"])


def test_gather_strips_root_single_rank(gpu_ctx):
    """mrt_gather_strips_root on a communicator of one rank: the root renders all root_span shares in place in one call
    (mrt_partition.rank_span) and there is nobody to receive from; and its argument checks."""
    torch = pytest.importorskip("torch")
    W, H, strip, span = 200, 136, 8, 3
    flat, _ = scenes.default_scene()
    gpu_ctx.upload(flat)
    cam = capi.default_viewport(W / H)
    full = gpu_ctx.render(cam, W, H)
    comm = capi.Comm(gpu_ctx, capi.comm_unique_id(), 0, 1)
    try:
        _, rgb, ids, tt = _frame_buffers(torch, W, H)
        stream = torch.cuda.current_stream().cuda_stream
        gpu_ctx.render_device(cam, W, H, rgb.data_ptr(), id_ptr=ids.data_ptr(), t_ptr=tt.data_ptr(), stream=stream,
                              rank=0, num_ranks=span, strip_rows=strip, frame_layout=1, rank_span=span)
        gpu_ctx.gather_strips(comm, None, None, rgb.data_ptr(), ids.data_ptr(), tt.data_ptr(), W, H, strip_rows=strip,
                              root_in_place=True, stream=stream, root_span=span)
        torch.cuda.synchronize()
        assert np.array_equal(rgb.view(H, W, 3).cpu().numpy(), full["rgb"])
        assert np.array_equal(ids.view(H, W).cpu().numpy(), full["id"])
        assert np.array_equal(tt.view(H, W).cpu().numpy(), full["t"])
        with pytest.raises(capi.MrtError):   # more than 64 shares
            gpu_ctx.gather_strips(comm, None, None, rgb.data_ptr(), ids.data_ptr(), tt.data_ptr(), W, H, strip_rows=strip,
                                  root_in_place=True, stream=stream, root_span=65)
        with pytest.raises(capi.MrtError):   # a span that runs past the partition
            gpu_ctx.render_device(cam, W, H, rgb.data_ptr(), stream=stream, rank=2, num_ranks=3, strip_rows=strip,
                                  rank_span=2)
    finally:
        comm.close()
