"""mrt_render_device_static, the entry point that allocates nothing (include/mrt.h): its output against
mrt_render_device's, through the C-ABI, bit for bit -- two calls in a row, the second with a smaller input, so the
static array is reused with the first call's contents still in it.  Collected last (the file name sorts last): a
failure here must not hide the other GPU tests from `pytest -x`."""
import numpy as np
import pytest

from raytracing_amd import capi, scenes

pytestmark = pytest.mark.gpu

KEYS = ("primary_rays", "shadow_rays", "hit_pixels", "occluded")


def _render(ctx, torch, cam, W, H, static, spp=1, stream=None, **part):
    dev = torch.device("cuda:0")
    rgb = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
    ids = torch.full((H, W), -7, dtype=torch.int32, device=dev)
    tt = torch.full((H, W), -1.0, dtype=torch.float32, device=dev)
    cnt = torch.full((8,), 99, dtype=torch.int64, device=dev)
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    ctx.render_device(cam, W, H, rgb.data_ptr(), id_ptr=ids.data_ptr(), t_ptr=tt.data_ptr(), stats_ptr=cnt.data_ptr(),
                      stream=st, spp=spp, static=static, **part)
    torch.cuda.synchronize()
    # counters 0..3: primary rays, shadow rays, hit pixels, occluded rays.  (4, 5 count exact tests and batches of candidates:
    # diagnostics of HOW the frame was culled, and the allocation-free entry point culls without the per-frame lists.)
    return rgb.cpu().numpy(), ids.cpu().numpy(), tt.cpu().numpy(), cnt.cpu().numpy()[:4]


def _same(a, b, what):
    for x, y, k in zip(a, b, ("rgb", "id", "t", "counters")):
        assert np.array_equal(x, y), f"{what}: {k} differs between mrt_render_device_static and mrt_render_device"


@pytest.mark.parametrize("scene", ["default", "spheres_3000", "default_brute", "terrain_teams"])
def test_static_entry_point_matches_render_device(scene):
    torch = pytest.importorskip("torch")
    ctx = capi.Context(0)
    try:
        if scene == "spheres_3000":          # tile-culling kernel: counters only
            spec = scenes.sphere_field(3000)
            flat, cam_of = spec.flatten(), spec.cam12
        elif scene == "terrain_teams":       # teams dealing leaves: with scratch, lists of (pixel, triangle) pairs from the
            spec = scenes.triangle_terrain(120, 80)  # per-frame triangle set-up; without, every survivor for all pixels
            flat, cam_of = spec.flatten(), spec.cam12
            ctx.set_cull_team(4)
        else:                                # small-scene kernel: counters + u / v tables
            flat, cam_of = scenes.default_scene()[0], capi.default_viewport
        ctx.upload(flat)
        if scene == "default_brute":
            ctx.set_cull(0)
        # two calls in a row, the second smaller (narrower AND lower: stale table entries beyond it must not matter)
        for (W, H, spp) in ((640, 360, 1), (250, 130, 1), (96, 48, 121), (33, 17, 1)):
            cam = cam_of(W / H)
            want = _render(ctx, torch, cam, W, H, static=False, spp=spp)
            got = _render(ctx, torch, cam, W, H, static=True, spp=spp)
            _same(got, want, f"{scene} {W}x{H} spp {spp}")
            assert want[3][0] == W * H * spp and (want[1] >= 0).sum() > 0
        # strips of a partition and frame_layout 1, on a stream of the caller's
        side = torch.cuda.Stream()
        W, H = 320, 200
        cam = cam_of(W / H)
        for part in ({"rank": 1, "num_ranks": 3}, {"rank": 0, "num_ranks": 2, "strip_rows": 16, "frame_layout": 1}):
            want = _render(ctx, torch, cam, W, H, static=False, **part)
            got = _render(ctx, torch, cam, W, H, static=True, stream=side.cuda_stream, **part)
            _same(got, want, f"{scene} {part}")
    finally:
        ctx.close()


def test_static_entry_point_limits_and_two_contexts():
    torch = pytest.importorskip("torch")
    a, b = capi.Context(0), capi.Context(0)
    try:
        a.upload(scenes.default_scene()[0])
        spec = scenes.sphere_field(500)
        b.upload(spec.flatten())
        dev = torch.device("cuda:0")
        buf = torch.zeros(64 * 32 * 3, dtype=torch.float32, device=dev)
        cam = capi.default_viewport(2.0)
        with pytest.raises(capi.MrtError, match="max_depth 0 only"):
            a.render_device(cam, 64, 32, buf.data_ptr(), max_depth=2, static=True)
        with pytest.raises(capi.MrtError, match="16384"):
            a.render_device(cam, 16392, 8, buf.data_ptr(), static=True)
        with pytest.raises(capi.MrtError, match="16384"):
            a.render_device(cam, 8, 20000, buf.data_ptr(), static=True)
        # two contexts share the GPU's one array: back to back (not overlapping) they must not disturb each other
        wa = _render(a, torch, cam, 200, 100, static=False)
        wb = _render(b, torch, spec.cam12(2.0), 256, 128, static=False)
        for _ in range(2):
            _same(_render(a, torch, cam, 200, 100, static=True), wa, "context a")
            _same(_render(b, torch, spec.cam12(2.0), 256, 128, static=True), wb, "context b")
    finally:
        a.close(); b.close()
