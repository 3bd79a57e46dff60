"""Scene specifications for the BASELINE configs.

A SceneSpec is the vertex-level description both sides start from: the product turns it into
the SoA of mrt_scene_desc through the C++ host layer's constructors (capi.triangle_precompute
etc.); the test oracle builds its own AoS objects from the same numbers.  Generators are
deterministic (splitmix64 seeded 1984, echoing the reference's RNG seed,
include/cuda_utils/curand_state.cuh:22).
"""
import numpy as np

from . import capi

MAT_DIFFUSE, MAT_METALLIC = capi.MAT_DIFFUSE, capi.MAT_METALLIC
LIGHT_DIRECTIONAL, LIGHT_POINT = capi.LIGHT_DIRECTIONAL, capi.LIGHT_POINT

# colour factories of the reference (include/core/material.cuh:81-114, :163-196)
_S = np.float32(1.0) / np.float32(16.0)
DIFFUSE_COLOURS = np.array([[_S, 0, 0], [0, _S, 0], [0, 0, _S], [_S, _S, 0], [_S, 0, _S], [_S, _S, _S],
                            [0.001, 0.001, 0.001]], dtype=np.float32)
METALLIC_COLOURS = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [1, 1, 1],
                             [0.001, 0.001, 0.001]], dtype=np.float32)

DEFAULT_LIGHT_DIR = (2.25, 1.0, -3.0)   # src/main.cu:184,236
DEFAULT_LIGHT_RGB = (16.0, 16.0, 16.0)  # src/main.cu:237
LOOK_FROM, LOOK_AT, UP = (0.0, 4.0, 6.0), (0.0, 0.0, -20.0), (0.0, 1.0, 0.0)  # src/main.cu:36-38


class SceneSpec:
    """Objects in scene order + lights.  kind: 0 sphere, 2 triangle (the variant indices)."""

    def __init__(self, name, spheres, sphere_obj, tri_verts, tri_obj, mat_kind, mat_rgb, lights, camera):
        self.name = name
        self.spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1, 4)
        self.sphere_obj = np.ascontiguousarray(sphere_obj, dtype=np.int32)
        self.tri_verts = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 3, 3)
        self.tri_obj = np.ascontiguousarray(tri_obj, dtype=np.int32)
        self.mat_kind = np.ascontiguousarray(mat_kind, dtype=np.uint8)
        self.mat_rgb = np.ascontiguousarray(mat_rgb, dtype=np.float32).reshape(-1, 3)
        self.lights = [(int(k), tuple(float(x) for x in v), tuple(float(x) for x in c)) for k, v, c in lights]
        self.camera = dict(camera)  # look_from, look_at, up, vfov
        assert len(self.sphere_obj) + len(self.tri_obj) <= len(self.mat_kind)

    @property
    def num_objects(self):
        return len(self.mat_kind)

    def flops_per_ray(self):
        """Algorithmic flops of one brute-force ray: 23 per sphere, 36 per triangle."""
        return 23 * len(self.sphere_obj) + 36 * len(self.tri_obj)

    def cam12(self, aspect):
        c = self.camera
        return capi.viewport(c["look_from"], c["look_at"], c["up"], c["vfov"], aspect)

    def flatten(self):
        """SoA for the C-ABI, through the C++ host constructors."""
        tri16 = capi.triangle_precompute(self.tri_verts)
        lk = [k for k, _, _ in self.lights]
        lv = [capi.directional_light_dir(v) if k == LIGHT_DIRECTIONAL else np.asarray(v, np.float32)
              for k, v, _ in self.lights]
        lc = [c for _, _, c in self.lights]
        return capi.FlatScene(self.mat_kind, self.mat_rgb, self.spheres, self.sphere_obj, tri16, self.tri_obj,
                              np.asarray(lk, np.uint8), np.asarray(lv, np.float32).reshape(-1, 3),
                              np.asarray(lc, np.float32).reshape(-1, 3))


def _splitmix64(seed, n):
    """n uniform floats in [0,1) with 24 random bits each (vectorised splitmix64)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) / np.float32(1 << 24)).astype(np.float32)


def _cycle_materials(n):
    """7 diffuse + 7 metallic factories, cycled."""
    k = np.arange(n) % 14
    kind = np.where(k < 7, MAT_DIFFUSE, MAT_METALLIC).astype(np.uint8)
    rgb = np.where((k < 7)[:, None], DIFFUSE_COLOURS[k % 7], METALLIC_COLOURS[k % 7]).astype(np.float32)
    return kind, rgb


def _camera(vfov=20.0):
    return {"look_from": LOOK_FROM, "look_at": LOOK_AT, "up": UP, "vfov": vfov}


def _default_lights():
    return [(LIGHT_DIRECTIONAL, DEFAULT_LIGHT_DIR, DEFAULT_LIGHT_RGB)]


def random_spheres(n, seed=1984, box=((-40.0, 40.0), (-10.0, 10.0), (-120.0, -10.0)), radii=(0.1, 0.6)):
    u = _splitmix64(seed, 4 * n).reshape(n, 4)
    out = np.empty((n, 4), dtype=np.float32)
    for a in range(3):
        lo, hi = np.float32(box[a][0]), np.float32(box[a][1])
        out[:, a] = lo + (hi - lo) * u[:, a]
    out[:, 3] = np.float32(radii[0]) + np.float32(radii[1] - radii[0]) * u[:, 3]
    return out


def terrain_triangles(nx, nz, x_range=(-50.0, 50.0), z_range=(-150.0, -10.0), amp=0.5, y0=-2.0):
    """Height-field grid of nx x nz cells, 2 triangles per cell: y = y0 + amp*sin(x*0.35)*cos(z*0.21)."""
    xs = np.linspace(x_range[0], x_range[1], nx + 1, dtype=np.float32)
    zs = np.linspace(z_range[0], z_range[1], nz + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = (np.float32(y0) + np.float32(amp) * np.sin(X * np.float32(0.35)) * np.cos(Z * np.float32(0.21))).astype(np.float32)
    P = np.stack([X, Y, Z], axis=-1)  # [nx+1, nz+1, 3]
    p00, p10, p01, p11 = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    t0 = np.stack([p00, p10, p11], axis=2)  # [nx, nz, 3(vertex), 3]
    t1 = np.stack([p00, p11, p01], axis=2)
    tris = np.stack([t0, t1], axis=2).reshape(-1, 3, 3)
    return np.ascontiguousarray(tris, dtype=np.float32)


def default_scene():
    """C2: the reference's hard-coded scene (src/main.cu:183-237), via the C++ host layer.

    Returned as (FlatScene, camera dict); the vertex-level numbers live in
    host/pipeline/default_scene.hpp and, independently, in the oracle."""
    return capi.FlatScene.default(), _camera(20.0)


def small_sphere_scene():
    """C1 'small sphere scene': the default scene's four spheres only."""
    flat = capi.FlatScene.default()
    sp = np.stack([flat.sphere_cx, flat.sphere_cy, flat.sphere_cz, flat.sphere_r], axis=1)
    obj = flat.sphere_obj
    return SceneSpec("small_spheres", sp, np.arange(len(obj)), np.zeros((0, 3, 3)), [], flat.mat_kind[obj],
                     flat.mat_rgb[obj], _default_lights(), _camera(20.0))


def sphere_field(n=10000, seed=1984):
    """C3: n random spheres, vfov widened to 40 degrees."""
    sp = random_spheres(n, seed)
    kind, rgb = _cycle_materials(n)
    return SceneSpec(f"spheres_{n}", sp, np.arange(n), np.zeros((0, 3, 3)), [], kind, rgb, _default_lights(),
                     _camera(40.0))


def triangle_terrain(nx=1000, nz=500):
    """C4: height-field mesh of 2*nx*nz triangles (1 000 000 by default)."""
    tv = terrain_triangles(nx, nz)
    n = tv.shape[0]
    kind, rgb = _cycle_materials(n)
    return SceneSpec(f"terrain_{n}", np.zeros((0, 4)), [], tv, np.arange(n), kind, rgb, _default_lights(),
                     _camera(40.0))


def mirror_hall(n_spheres=24, seed=1984):
    """All-metallic scene for the mirror-chain path (max_depth > 0): a floor, a back wall and two
    side walls (two triangles each) around a cloud of spheres, one directional and one point light."""
    sp = random_spheres(n_spheres, seed, box=((-6.0, 6.0), (-2.0, 3.0), (-24.0, -12.0)), radii=(0.4, 1.4))
    def quad(p, e1, e2):
        p, e1, e2 = (np.asarray(x, np.float32) for x in (p, e1, e2))
        return [[p, p + e1, p + e1 + e2], [p, p + e1 + e2, p + e2]]
    tv = np.asarray(quad((-9, -3.5, -8), (18, 0, 0), (0, 0, -20)) + quad((-9, -3.5, -28), (18, 0, 0), (0, 9, 0)) +
                    quad((-9, -3.5, -8), (0, 0, -20), (0, 9, 0)) + quad((9, -3.5, -8), (0, 9, 0), (0, 0, -20)), np.float32)
    n = n_spheres + len(tv)
    u = _splitmix64(seed + 7, 3 * n).reshape(n, 3)
    rgb = (np.float32(0.35) + np.float32(0.6) * u).astype(np.float32)
    kind = np.full(n, MAT_METALLIC, np.uint8)
    lights = _default_lights() + [(LIGHT_POINT, (0.0, 4.5, -14.0), (2.0, 2.0, 2.0))]
    return SceneSpec(f"mirror_hall_{n_spheres}", sp, np.arange(n_spheres), tv, n_spheres + np.arange(len(tv)), kind, rgb,
                     lights, _camera(40.0))


def light_stress(n_spheres=256, n_tris=256, grid=8, seed=1984):
    """C5: spheres and triangles interleaved in object order + grid*grid point lights at y = 12."""
    sp = random_spheres(n_spheres, seed, box=((-12.0, 12.0), (-3.0, 3.0), (-45.0, -12.0)), radii=(0.3, 1.0))
    # a coarse terrain patch scaled into the same box
    nx = max(1, int(round((n_tris / 2) ** 0.5)))
    nz = max(1, (n_tris // 2 + nx - 1) // nx)
    tv = terrain_triangles(nx, nz, x_range=(-14.0, 14.0), z_range=(-48.0, -10.0), amp=0.8, y0=-3.5)[:n_tris]
    n_tris = tv.shape[0]
    n = n_spheres + n_tris
    # interleave: spheres take the even object numbers while both kinds last
    m = min(n_spheres, n_tris)
    sphere_obj = np.concatenate([np.arange(m) * 2, 2 * m + np.arange(n_spheres - m)])
    tri_obj = np.concatenate([np.arange(m) * 2 + 1, 2 * m + np.arange(n_tris - m)])
    kind, rgb = _cycle_materials(n)
    gx = np.linspace(-14.0, 14.0, grid, dtype=np.float32)
    gz = np.linspace(-46.0, -8.0, grid, dtype=np.float32)
    lights = [(LIGHT_POINT, (float(x), 12.0, float(z)), (0.5, 0.5, 0.5)) for x in gx for z in gz]
    return SceneSpec(f"lights_{grid * grid}", sp, sphere_obj, tv, tri_obj, kind, rgb, lights, _camera(40.0))
