"""Scene mutations shared by the stress tools."""
import numpy as np
from raytracing_amd import capi


def moved(spec, shift=0.0, scale=1.0):
    """The whole scene (geometry, point lights, camera) scaled, then moved by shift * (0.6, -0.4, 1)."""
    if shift == 0.0 and scale == 1.0:
        return spec
    off = np.array([0.6, -0.4, 1.0]) * shift
    spec.spheres = (spec.spheres.astype(np.float64) * scale + np.append(off, 0.0)).astype(np.float32)
    spec.tri_verts = (spec.tri_verts.astype(np.float64) * scale + off).astype(np.float32)
    spec.lights = [(k, tuple(np.asarray(v) * scale + off) if k == capi.LIGHT_POINT else v, c) for k, v, c in spec.lights]
    spec.camera["look_from"] = tuple(np.asarray(spec.camera["look_from"], np.float64) * scale + off)
    spec.camera["look_at"] = tuple(np.asarray(spec.camera["look_at"], np.float64) * scale + off)
    return spec


def make_wild(spec, seed):
    rng = np.random.default_rng(77000 + seed)
    what = int(rng.integers(0, 7))
    lf, la = np.asarray(spec.camera["look_from"], np.float64), np.asarray(spec.camera["look_at"], np.float64)
    if what == 0:
        spec.camera["vfov"] = float(rng.choice([0.02, 0.5, 120.0, 170.0, 179.0]))
    elif what == 1:   # camera far away or very close to what it looks at
        spec.camera["look_from"] = tuple(la + (lf - la) * float(10.0 ** rng.uniform(-3, 3)))
    elif what == 2 and len(spec.spheres):
        spec.spheres[:, 3] = (10.0 ** rng.uniform(-4, 4, len(spec.spheres))).astype(np.float32)
    elif what == 3 and len(spec.tri_verts):   # repeated / collinear vertices
        k = rng.integers(0, len(spec.tri_verts), max(1, len(spec.tri_verts) // 4))
        spec.tri_verts[k, 2] = spec.tri_verts[k, 1]
        k = rng.integers(0, len(spec.tri_verts), max(1, len(spec.tri_verts) // 4))
        spec.tri_verts[k, 2] = 2 * spec.tri_verts[k, 1] - spec.tri_verts[k, 0]
    elif what == 4:
        bad = float(rng.choice([np.nan, np.inf, -np.inf]))
        if len(spec.spheres) and rng.random() < 0.5:
            spec.spheres[rng.integers(0, len(spec.spheres)), rng.integers(0, 4)] = bad
        elif len(spec.tri_verts):
            spec.tri_verts[rng.integers(0, len(spec.tri_verts)), rng.integers(0, 3), rng.integers(0, 3)] = bad
    elif what == 5 and spec.lights:   # a point light on top of a primitive, or very far away
        c = spec.spheres[0, :3] if len(spec.spheres) else spec.tri_verts[0, 0]
        far = float(10.0 ** rng.uniform(3, 6))
        pos = tuple(np.asarray(c, np.float64) + (rng.normal(size=3) * (far if rng.random() < 0.5 else 1e-3)))
        spec.lights[0] = (capi.LIGHT_POINT, pos, spec.lights[0][2])
    else:             # camera inside a sphere / on a triangle
        if len(spec.spheres):
            spec.camera["look_from"] = tuple(spec.spheres[0, :3].astype(np.float64) + 0.3 * spec.spheres[0, 3])
    return spec
