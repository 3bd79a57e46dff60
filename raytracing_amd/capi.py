"""ctypes binding of libmrt.so (include/mrt.h) and libmrt_host.so (host/host_capi.cpp).

No fallback of any kind: if a library is missing or a call fails, this raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
# MRT_LIB: an experimental build of the same library (make -C raytracing_amd/csrc variant ...), for tools/ only
LIBMRT_PATH = os.environ.get("MRT_LIB") or os.path.join(LIB_DIR, "libmrt.so")
LIBHOST_PATH = os.path.join(LIB_DIR, "libmrt_host.so")

GEOM_SPHERE, GEOM_PLANE, GEOM_TRIANGLE = 0, 1, 2
MAT_DIFFUSE, MAT_METALLIC = 0, 1
LIGHT_DIRECTIONAL, LIGHT_POINT = 0, 1
PLANE_RGB, PLANE_ID, PLANE_T, PLANES_ALL = 1, 2, 4, 7

# every symbol include/mrt.h declares (tests check the library exports all of them)
MRT_SYMBOLS = [
    "mrt_device_count", "mrt_ctx_create", "mrt_ctx_destroy", "mrt_ctx_set_option", "mrt_last_error", "mrt_status_string",
    "mrt_triangle_bound", "mrt_rng_state", "mrt_host_alloc", "mrt_host_free", "mrt_scene_upload", "mrt_local_rows", "mrt_render", "mrt_render_device", "mrt_render_device_static", "mrt_scatter_strips",
    "mrt_scatter_strips_planes", "mrt_scatter_strips_span", "mrt_gather_strips_planes", "mrt_gather_strips_root", "mrt_intersect", "mrt_intersect_interval", "mrt_comm_unique_id", "mrt_comm_create", "mrt_comm_create_local", "mrt_comm_destroy",
    "mrt_gather_strips", "mrt_render_multi",
]

_f = C.POINTER(C.c_float)
_i32 = C.POINTER(C.c_int32)
_u8 = C.POINTER(C.c_uint8)


class MrtError(RuntimeError):
    pass


class SceneDesc(C.Structure):
    _fields_ = [
        ("num_objects", C.c_int32), ("mat_kind", _u8), ("mat_rgb", _f),
        ("num_spheres", C.c_int32), ("sphere_cx", _f), ("sphere_cy", _f), ("sphere_cz", _f), ("sphere_r", _f),
        ("sphere_obj", _i32),
        ("num_triangles", C.c_int32), ("tri", _f), ("tri_obj", _i32),
        ("num_lights", C.c_int32), ("light_kind", _u8), ("light_vec", _f), ("light_rgb", _f),
    ]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("horizontal", C.c_float * 3), ("vertical", C.c_float * 3),
                ("up_left", C.c_float * 3)]

    @classmethod
    def from12(cls, cam12):
        a = [float(x) for x in np.asarray(cam12, dtype=np.float32).reshape(12)]
        c = cls()
        c.origin[:] = a[0:3]
        c.horizontal[:] = a[3:6]
        c.vertical[:] = a[6:9]
        c.up_left[:] = a[9:12]
        return c


class Partition(C.Structure):
    _fields_ = [("rank", C.c_int32), ("num_ranks", C.c_int32), ("strip_rows", C.c_int32), ("frame_layout", C.c_int32),
                ("rank_span", C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("part", Partition)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("hit_pixels", C.c_uint64),
                ("occluded", C.c_uint64), ("exact_tests", C.c_uint64), ("clusters_opened", C.c_uint64),
                ("kernel_ms", C.c_float)]

    def as_dict(self):
        return {"primary_rays": int(self.primary_rays), "shadow_rays": int(self.shadow_rays),
                "hit_pixels": int(self.hit_pixels), "occluded": int(self.occluded), "exact_tests": int(self.exact_tests),
                "clusters_opened": int(self.clusters_opened), "kernel_ms": float(self.kernel_ms)}


_libmrt = None
_libhost = None


def libmrt():
    """Load libmrt.so.  Raises if it was not built (no fallback)."""
    global _libmrt
    if _libmrt is None:
        if not os.path.exists(LIBMRT_PATH):
            raise MrtError(f"{LIBMRT_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIBMRT_PATH)
        L.mrt_device_count.argtypes = [C.POINTER(C.c_int)]
        L.mrt_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mrt_ctx_destroy.argtypes = [C.c_void_p]
        L.mrt_ctx_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mrt_last_error.argtypes = [C.c_void_p]
        L.mrt_last_error.restype = C.c_char_p
        L.mrt_status_string.argtypes = [C.c_int]
        L.mrt_status_string.restype = C.c_char_p
        L.mrt_scene_upload.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.mrt_triangle_bound.argtypes = [_f, _f]
        L.mrt_rng_state.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
        L.mrt_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.mrt_host_free.argtypes = [C.c_void_p]
        L.mrt_local_rows.argtypes = [C.c_int32, C.POINTER(Partition), _i32]
        L.mrt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.POINTER(Stats)]
        L.mrt_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mrt_render_device_static.argtypes = L.mrt_render_device.argtypes
        L.mrt_scatter_strips.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.mrt_scatter_strips_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.mrt_gather_strips_planes.argtypes = [C.c_void_p] * 7 + [C.c_int32] * 5 + [C.c_void_p]
        L.mrt_scatter_strips_span.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 7 + [C.c_void_p]
        L.mrt_gather_strips_root.argtypes = [C.c_void_p] * 7 + [C.c_int32] * 5 + [C.c_void_p]
        L.mrt_intersect.argtypes = [C.c_void_p, C.c_int, _f, _f, C.c_float, C.c_float, C.c_int64, _u8, _f, _f, _f]
        L.mrt_intersect_interval.argtypes = [C.c_void_p, C.c_int, _f, _f, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int64, _u8,
                                             _f, _f, _f]
        L.mrt_comm_unique_id.argtypes = [C.c_void_p]
        L.mrt_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        L.mrt_comm_create_local.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]
        L.mrt_comm_destroy.argtypes = [C.c_void_p]
        L.mrt_gather_strips.argtypes = [C.c_void_p] * 7 + [C.c_int32] * 4 + [C.c_void_p]
        L.mrt_render_multi.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.POINTER(Camera),
                                       C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
        _libmrt = L
    return _libmrt


def libhost():
    """Load libmrt_host.so (C exports of the C++ host layer)."""
    global _libhost
    if _libhost is None:
        if not os.path.exists(LIBHOST_PATH):
            raise MrtError(f"{LIBHOST_PATH} is missing: build it with `make -C raytracing_amd/host`")
        L = C.CDLL(LIBHOST_PATH)
        L.rth_viewport.argtypes = [_f, _f, _f, C.c_float, C.c_float, _f]
        L.rth_triangle_precompute.argtypes = [_f, C.c_int64, _f]
        L.rth_directional_light_dir.argtypes = [_f, _f]
        L.rth_default_scene_create.restype = C.c_void_p
        L.rth_flat_scene_destroy.argtypes = [C.c_void_p]
        L.rth_flat_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        L.rth_msmesh_offsets.argtypes = [C.c_float, C.c_float, _f]
        L.rth_sphere_can_hit.argtypes = [_f, C.c_float, _f, _f]
        L.rth_triangle_can_hit.argtypes = [_f, _f, _f]
        L.rth_default_scene_transformed_create.argtypes = [_f, _f]
        L.rth_default_scene_transformed_create.restype = C.c_void_p
        _libhost = L
    return _libhost


def _fp(a):
    return a.ctypes.data_as(_f)


# ------------------------------------------------------------------ host-layer helpers

def viewport(look_from, look_at, up, vfov_deg, aspect):
    """rt::Viewport constructor -> 12 floats (origin, horizontal, vertical, up_left)."""
    out = np.zeros(12, dtype=np.float32)
    a = [np.asarray(v, dtype=np.float32) for v in (look_from, look_at, up)]
    libhost().rth_viewport(_fp(a[0]), _fp(a[1]), _fp(a[2]), float(vfov_deg), float(aspect), _fp(out))
    return out


def default_viewport(aspect=2.0):
    """config::make_viewport() of the reference (src/main.cu:34-41)."""
    return viewport((0.0, 4.0, 6.0), (0.0, 0.0, -20.0), (0.0, 1.0, 0.0), 20.0, aspect)


def triangle_precompute(verts):
    """rt::Triangle constructor, batched.  verts [n,3,3] -> [16,n] streams."""
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    n = v.shape[0]
    out = np.zeros((16, n), dtype=np.float32)
    if n:
        libhost().rth_triangle_precompute(_fp(v), n, _fp(out))
    return out


def triangle_bound(tri16):
    """Bounding ball (x, y, z, r) the culling kernels use for a precomputed triangle (host-only diagnostic)."""
    q = np.ascontiguousarray(tri16, dtype=np.float32).reshape(16)
    out = np.zeros(4, dtype=np.float32)
    rc = libmrt().mrt_triangle_bound(_fp(q), _fp(out))
    if rc != 0:
        raise MrtError(f"mrt_triangle_bound: {rc}")
    return out


def rng_state(seed, subsequence):
    """XORWOW state after curand_init(seed, subsequence, 0) as the library computes it (host-only diagnostic):
    uint32[6] = xorshift words v[0..4], Weyl counter d."""
    out = np.zeros(6, dtype=np.uint32)
    rc = libmrt().mrt_rng_state(int(seed), int(subsequence), out.ctypes.data)
    if rc != 0:
        raise MrtError(f"mrt_rng_state: {rc}")
    return out


class HostBuffer:
    """Page-locked host memory from mrt_host_alloc, viewed as numpy arrays; free() or the garbage collector
    returns it."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        ptr = C.c_void_p()
        rc = libmrt().mrt_host_alloc(self.nbytes, C.byref(ptr))
        if rc != 0 or not ptr.value:
            raise MrtError(f"mrt_host_alloc({nbytes}): {rc}")
        self.ptr = ptr.value

    def view(self, dtype, shape, offset=0):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert offset + n <= self.nbytes
        buf = (C.c_char * n).from_address(self.ptr + offset)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            libmrt().mrt_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sphere_can_hit(center, radius, origin, direction):
    """rt::Sphere::can_hit through rt::SceneObject (host C++)."""
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (center, origin, direction)]
    return bool(libhost().rth_sphere_can_hit(_fp(a[0]), float(radius), _fp(a[1]), _fp(a[2])))


def triangle_can_hit(verts, origin, direction):
    """rt::Triangle::can_hit through rt::SceneObject (host C++); verts [3, 3]."""
    a = [np.ascontiguousarray(v, dtype=np.float32).reshape(-1) for v in (verts, origin, direction)]
    return bool(libhost().rth_triangle_can_hit(_fp(a[0]), _fp(a[1]), _fp(a[2])))


def directional_light_dir(d):
    a = np.asarray(d, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    libhost().rth_directional_light_dir(_fp(a), _fp(out))
    return out


class FlatScene:
    """SoA scene in the layout of mrt_scene_desc (numpy arrays kept alive here)."""

    def __init__(self, mat_kind, mat_rgb, spheres, sphere_obj, tri16, tri_obj, light_kind, light_vec, light_rgb):
        self.mat_kind = np.ascontiguousarray(mat_kind, dtype=np.uint8)
        self.mat_rgb = np.ascontiguousarray(mat_rgb, dtype=np.float32).reshape(-1, 3)
        sp = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
        self.sphere_cx = np.ascontiguousarray(sp[:, 0])
        self.sphere_cy = np.ascontiguousarray(sp[:, 1])
        self.sphere_cz = np.ascontiguousarray(sp[:, 2])
        self.sphere_r = np.ascontiguousarray(sp[:, 3])
        self.sphere_obj = np.ascontiguousarray(sphere_obj, dtype=np.int32)
        self.tri = np.ascontiguousarray(tri16, dtype=np.float32).reshape(16, -1)
        self.tri_obj = np.ascontiguousarray(tri_obj, dtype=np.int32)
        self.light_kind = np.ascontiguousarray(light_kind, dtype=np.uint8)
        self.light_vec = np.ascontiguousarray(light_vec, dtype=np.float32).reshape(-1, 3)
        self.light_rgb = np.ascontiguousarray(light_rgb, dtype=np.float32).reshape(-1, 3)

    @property
    def num_objects(self):
        return len(self.mat_kind)

    @property
    def num_spheres(self):
        return len(self.sphere_obj)

    @property
    def num_triangles(self):
        return len(self.tri_obj)

    @property
    def num_lights(self):
        return len(self.light_kind)

    def desc(self):
        d = SceneDesc()
        d.num_objects = self.num_objects
        d.mat_kind = self.mat_kind.ctypes.data_as(_u8)
        d.mat_rgb = _fp(self.mat_rgb)
        d.num_spheres = self.num_spheres
        d.sphere_cx, d.sphere_cy, d.sphere_cz, d.sphere_r = (_fp(self.sphere_cx), _fp(self.sphere_cy),
                                                             _fp(self.sphere_cz), _fp(self.sphere_r))
        d.sphere_obj = self.sphere_obj.ctypes.data_as(_i32)
        d.num_triangles = self.num_triangles
        d.tri = _fp(self.tri)
        d.tri_obj = self.tri_obj.ctypes.data_as(_i32)
        d.num_lights = self.num_lights
        d.light_kind = self.light_kind.ctypes.data_as(_u8)
        d.light_vec = _fp(self.light_vec)
        d.light_rgb = _fp(self.light_rgb)
        return d

    @classmethod
    def default(cls, rotation=None, translation=None):
        """The reference's hard-coded scene, built and flattened by the C++ host layer; with a rotation (3x3) and a
        translation, rt::Transform is applied to every object first (SceneObject::apply_transform)."""
        H = libhost()
        if rotation is None and translation is None:
            h = C.c_void_p(H.rth_default_scene_create())
        else:
            r = np.ascontiguousarray(np.eye(3) if rotation is None else rotation, dtype=np.float32).reshape(9)
            t = np.ascontiguousarray(np.zeros(3) if translation is None else translation, dtype=np.float32).reshape(3)
            h = C.c_void_p(H.rth_default_scene_transformed_create(_fp(r), _fp(t)))
        try:
            d = SceneDesc()
            H.rth_flat_scene_desc(h, C.byref(d))
            arr = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
            ns, nt, no, nl = d.num_spheres, d.num_triangles, d.num_objects, d.num_lights
            spheres = np.stack([arr(d.sphere_cx, ns, np.float32), arr(d.sphere_cy, ns, np.float32),
                                arr(d.sphere_cz, ns, np.float32), arr(d.sphere_r, ns, np.float32)], axis=1)
            return cls(arr(d.mat_kind, no, np.uint8), arr(d.mat_rgb, no * 3, np.float32), spheres,
                       arr(d.sphere_obj, ns, np.int32), arr(d.tri, nt * 16, np.float32), arr(d.tri_obj, nt, np.int32),
                       arr(d.light_kind, nl, np.uint8), arr(d.light_vec, nl * 3, np.float32),
                       arr(d.light_rgb, nl * 3, np.float32))
        finally:
            H.rth_flat_scene_destroy(h)


# ------------------------------------------------------------------ GPU context

def device_count():
    n = C.c_int(0)
    rc = libmrt().mrt_device_count(C.byref(n))
    if rc != 0:
        raise MrtError(f"mrt_device_count: {rc} {libmrt().mrt_last_error(None).decode()}")
    return n.value


def local_rows(height, rank=0, num_ranks=1, strip_rows=8, rank_span=1):
    p = Partition(rank, num_ranks, strip_rows, 0, rank_span)
    r = C.c_int32(0)
    rc = libmrt().mrt_local_rows(height, C.byref(p), C.byref(r))
    if rc != 0:
        raise MrtError(f"mrt_local_rows: {rc}")
    return r.value


def local_row_index(height, rank, num_ranks, strip_rows=8, rank_span=1):
    """Global row numbers of a rank's local rows, in local order (host mirror of the strip rule); rank_span > 1: the
    strips of ranks rank .. rank + rank_span - 1 (mrt_partition.rank_span)."""
    rows = []
    nstrips = (height + strip_rows - 1) // strip_rows
    for s in range(nstrips):
        if rank <= s % num_ranks < rank + max(1, rank_span):
            rows.extend(range(s * strip_rows, min(height, (s + 1) * strip_rows)))
    return np.asarray(rows, dtype=np.int64)


class Context:
    """One GPU (mrt_ctx)."""

    def __init__(self, device=0):
        self._L = libmrt()
        h = C.c_void_p()
        rc = self._L.mrt_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise MrtError(f"mrt_ctx_create({device}): {rc} {self._L.mrt_last_error(None).decode()}")
        self._h = h
        self.device = device
        self.cull = int(os.environ.get("MRT_CULL", "1"))

    def close(self):
        if getattr(self, "_h", None):
            self._L.mrt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise MrtError(f"{what}: {rc} ({self._L.mrt_status_string(rc).decode()}) "
                           f"{self._L.mrt_last_error(self._h).decode()}")

    def set_cull(self, mode):
        """Render kernel: 0/False brute force, 1/True automatic (default), 2 tile culling over the primitive
        hierarchy, 5 the small-scene kernel.  Identical outputs."""
        mode = int(mode)
        self._check(self._L.mrt_ctx_set_option(self._h, 1, mode), "mrt_ctx_set_option")
        self.cull = mode

    def set_cull_team(self, mode):
        """Waves per tile in the tile-culling kernel (MRT_OPT_CULL_TEAM): 0 by the scene, 1 one wave per tile, 4 teams
        of four dealing leaf clusters, 5 teams of four dealing lights.  Identical outputs."""
        self._check(self._L.mrt_ctx_set_option(self._h, 4, int(mode)), "mrt_ctx_set_option")

    def set_lists(self, mode, cap=0):
        """MRT_OPT_LISTS: where the tile-culling kernel's tiles get their primary candidates from -- 0 by the scene,
        1 no lists (walk), 2 lists of primitives, 3 lists of leaf clusters; cap > 0 (MRT_OPT_LIST_CAP, test hook)
        shortens every list so that small frames reach the overflow paths.  Identical outputs."""
        self._check(self._L.mrt_ctx_set_option(self._h, 6, int(mode)), "mrt_ctx_set_option")
        self._check(self._L.mrt_ctx_set_option(self._h, 7, int(cap)), "mrt_ctx_set_option")

    def set_view_cache(self, on):
        """MRT_OPT_VIEW_CACHE: False = the small-scene launcher treats every render as the first of its view."""
        self._check(self._L.mrt_ctx_set_option(self._h, 5, 1 if on else 0), "mrt_ctx_set_option")

    def set_rng_seed(self, seed):
        """Seed of the per-pixel random streams of diffuse bounces (default 1984, as the reference)."""
        self._check(self._L.mrt_ctx_set_option(self._h, 2, int(seed)), "mrt_ctx_set_option")

    def fail_next_upload(self):
        """Test hook (MRT_OPT_FAIL_NEXT_UPLOAD): the next upload fails at its device allocation."""
        self._check(self._L.mrt_ctx_set_option(self._h, 3, 1), "mrt_ctx_set_option")

    def upload(self, flat):
        d = flat.desc()
        self._check(self._L.mrt_scene_upload(self._h, C.byref(d)), "mrt_scene_upload")

    @staticmethod
    def _params(W, H, spp, max_depth, rank, num_ranks, strip_rows, frame_layout=0, rank_span=1):
        p = RenderParams()
        p.width, p.height, p.spp, p.max_depth = W, H, spp, max_depth
        p.part = Partition(rank, num_ranks, strip_rows, frame_layout, rank_span)
        return p

    def render(self, cam12, W, H, spp=1, max_depth=0, rank=0, num_ranks=1, strip_rows=8, want_stats=True, rank_span=1):
        """Blocking render into host arrays of the rank's local rows."""
        rows = local_rows(H, rank, num_ranks, strip_rows, rank_span)
        rgb = np.zeros((rows, W, 3), dtype=np.float32)
        ids = np.full((rows, W), -2, dtype=np.int32)
        t = np.zeros((rows, W), dtype=np.float32)
        cam = Camera.from12(cam12)
        p = self._params(W, H, spp, max_depth, rank, num_ranks, strip_rows, 0, rank_span)
        st = Stats()
        rc = self._L.mrt_render(self._h, C.byref(cam), C.byref(p), rgb.ctypes.data, ids.ctypes.data, t.ctypes.data,
                                C.byref(st) if want_stats else None)
        self._check(rc, "mrt_render")
        return {"rgb": rgb, "id": ids, "t": t, "stats": st.as_dict() if want_stats else None}

    def render_into(self, cam12, W, H, rgb, ids=None, t=None, spp=1, max_depth=0):
        """Blocking render of the whole frame into caller-owned host arrays (e.g. views of page-locked memory
        from host_alloc): rgb float32 [H, W, 3], ids int32 [H, W], t float32 [H, W]."""
        assert rgb.dtype == np.float32 and rgb.size == H * W * 3 and rgb.flags["C_CONTIGUOUS"]
        cam = Camera.from12(cam12)
        p = self._params(W, H, spp, max_depth, 0, 1, 8)
        st = Stats()
        rc = self._L.mrt_render(self._h, C.byref(cam), C.byref(p), rgb.ctypes.data,
                                ids.ctypes.data if ids is not None else None,
                                t.ctypes.data if t is not None else None, C.byref(st))
        self._check(rc, "mrt_render")
        return st.as_dict()

    def render_device(self, cam12, W, H, rgb_ptr, id_ptr=None, t_ptr=None, stats_ptr=None, stream=None, spp=1,
                      max_depth=0, rank=0, num_ranks=1, strip_rows=8, frame_layout=0, static=False, rank_span=1):
        """Asynchronous render into device buffers (raw pointers, e.g. torch tensor.data_ptr()): the rank's rows
        back to back, or (frame_layout=1) at their final places in whole-frame buffers.  static=True goes through
        mrt_render_device_static, the entry point that allocates nothing."""
        cam = Camera.from12(cam12)
        p = self._params(W, H, spp, max_depth, rank, num_ranks, strip_rows, frame_layout, rank_span)
        fn = self._L.mrt_render_device_static if static else self._L.mrt_render_device
        rc = fn(self._h, C.byref(cam), C.byref(p), C.c_void_p(rgb_ptr), C.c_void_p(id_ptr) if id_ptr else None,
                C.c_void_p(t_ptr) if t_ptr else None, C.c_void_p(stats_ptr) if stats_ptr else None,
                C.c_void_p(stream) if stream else None)
        self._check(rc, "mrt_render_device_static" if static else "mrt_render_device")

    def scatter_strips(self, staging_ptr, rgb_ptr, id_ptr, t_ptr, W, H, num_ranks, strip_rows=8, skip_rank=0,
                       stream=None, planes=None, skip_span=1):
        """Rank 0 after the gather: the packed buffers of the ranks other than skip_rank (in rank order at
        staging_ptr) go to their strips' places in the three frame planes.  planes: PLANE_* mask of what the packed
        buffers hold (None: all three, through the older entry point)."""
        v = lambda p: C.c_void_p(p) if p else None
        if skip_span != 1:   # (the ranks skip_rank .. skip_rank + skip_span - 1 were rendered in place)
            rc = self._L.mrt_scatter_strips_span(self._h, v(staging_ptr), v(rgb_ptr), v(id_ptr), v(t_ptr), W, H, num_ranks,
                                                 strip_rows, skip_rank, skip_span, int(PLANES_ALL if planes is None else planes),
                                                 v(stream))
        elif planes is None:
            rc = self._L.mrt_scatter_strips(self._h, v(staging_ptr), v(rgb_ptr), v(id_ptr), v(t_ptr), W, H, num_ranks,
                                            strip_rows, skip_rank, v(stream))
        else:
            rc = self._L.mrt_scatter_strips_planes(self._h, v(staging_ptr), v(rgb_ptr), v(id_ptr), v(t_ptr), W, H,
                                                   num_ranks, strip_rows, skip_rank, int(planes), v(stream))
        self._check(rc, "mrt_scatter_strips")

    def gather_strips(self, comm, packed_ptr, staging_ptr, rgb_ptr, id_ptr, t_ptr, W, H, strip_rows=8, root_in_place=True,
                      stream=None, planes=None, root_span=1):
        """The exchange step of a multi-GPU frame over RCCL (mrt_gather_strips): ranks other than 0 send their packed
        buffer, rank 0 receives them all and puts the strips in place.  Asynchronous on `stream`."""
        v = lambda p: C.c_void_p(p) if p else None
        if root_span != 1:   # the weighted exchange: rank 0 rendered root_span shares of ranks - 1 + root_span in place
            rc = self._L.mrt_gather_strips_root(self._h, comm._h, v(packed_ptr), v(staging_ptr), v(rgb_ptr), v(id_ptr),
                                                v(t_ptr), W, H, strip_rows, int(root_span),
                                                int(PLANES_ALL if planes is None else planes), v(stream))
        elif planes is None:
            rc = self._L.mrt_gather_strips(self._h, comm._h, v(packed_ptr), v(staging_ptr), v(rgb_ptr), v(id_ptr),
                                           v(t_ptr), W, H, strip_rows, 1 if root_in_place else 0, v(stream))
        else:
            rc = self._L.mrt_gather_strips_planes(self._h, comm._h, v(packed_ptr), v(staging_ptr), v(rgb_ptr), v(id_ptr),
                                                  v(t_ptr), W, H, strip_rows, 1 if root_in_place else 0, int(planes),
                                                  v(stream))
        self._check(rc, "mrt_gather_strips")

    def intersect(self, prim_kind, prim, rays, tmin, tmax, lower_closed=True, upper_closed=True):
        """Batched hit test of one primitive against rays [n,6] (origin, direction); the interval's bounds are closed
        unless said otherwise (rt::Interval's four modes)."""
        prim = np.ascontiguousarray(prim, dtype=np.float32).reshape(-1)
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        hit = np.zeros(n, dtype=np.uint8)
        t = np.zeros(n, dtype=np.float32)
        point = np.zeros((n, 3), dtype=np.float32)
        normal = np.zeros((n, 3), dtype=np.float32)
        if lower_closed and upper_closed:
            rc = self._L.mrt_intersect(self._h, int(prim_kind), _fp(prim), _fp(rays), float(tmin), float(tmax), n,
                                       hit.ctypes.data_as(_u8), _fp(t), _fp(point), _fp(normal))
        else:
            rc = self._L.mrt_intersect_interval(self._h, int(prim_kind), _fp(prim), _fp(rays), float(tmin), float(tmax),
                                                1 if lower_closed else 0, 1 if upper_closed else 0, n,
                                                hit.ctypes.data_as(_u8), _fp(t), _fp(point), _fp(normal))
        self._check(rc, "mrt_intersect")
        return hit.astype(bool), t, point, normal


def comm_unique_id():
    """128 bytes that name a new communicator (ncclGetUniqueId): made on rank 0, handed to every rank by the caller."""
    L = libmrt()
    buf = (C.c_uint8 * 128)()
    rc = L.mrt_comm_unique_id(buf)
    if rc != 0:
        raise MrtError(f"mrt_comm_unique_id: {rc} {L.mrt_last_error(None).decode()}")
    return bytes(buf)


class Comm:
    """One rank's RCCL communicator (mrt_comm), tied to a Context."""

    def __init__(self, ctx, unique_id, rank, num_ranks):
        self._L, self.ctx, self.rank, self.num_ranks = ctx._L, ctx, rank, num_ranks
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        ctx._check(self._L.mrt_comm_create(ctx._h, buf, rank, num_ranks, C.byref(h)), "mrt_comm_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.mrt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render_multi(ctxs, cam12, W, H, spp=1, max_depth=0, strip_rows=8, comms=None, want_stats=True, planes=PLANES_ALL):
    """One process, len(ctxs) GPUs (mrt_render_multi): blocking render of the whole frame into host arrays.  comms:
    raw handles from mrt_comm_create_local, created here (and destroyed) when None and more than one context."""
    L = libmrt()
    n = len(ctxs)
    hs = (C.c_void_p * n)(*[c._h for c in ctxs])
    own = None
    if comms is None and n > 1:
        own = (C.c_void_p * n)()
        ctxs[0]._check(L.mrt_comm_create_local(hs, n, own), "mrt_comm_create_local")
        comms = own
    rgb = np.zeros((H, W, 3), dtype=np.float32)
    ids = np.full((H, W), -2, dtype=np.int32)
    t = np.zeros((H, W), dtype=np.float32)
    cam = Camera.from12(cam12)
    p = Context._params(W, H, spp, max_depth, 0, 1, strip_rows)
    st = Stats()
    try:
        # a plane the caller does not ask for (NULL) is neither rendered by the other ranks nor gathered
        rc = L.mrt_render_multi(hs, comms, n, C.byref(cam), C.byref(p), rgb.ctypes.data,
                                ids.ctypes.data if planes & PLANE_ID else None,
                                t.ctypes.data if planes & PLANE_T else None, C.byref(st) if want_stats else None)
        ctxs[0]._check(rc, "mrt_render_multi")
    finally:
        if own is not None:
            for h in own:
                L.mrt_comm_destroy(h)
    return {"rgb": rgb, "id": ids, "t": t, "stats": st.as_dict() if want_stats else None}
