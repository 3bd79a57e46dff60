"""The random streams of the diffuse bounce: cuRAND's XORWOW restated from its published
algorithm -- in the oracle (oracle/rt_oracle.c) and, independently, in the library (raytracing_amd/csrc/mrt_rng.hip).

cuRAND itself is absent (no CUDA in this image) and the reference holds no output vector of it, so parity with
cuRAND's actual streams is UNPINNED.  What is pinned here, on the CPU:
  - the recurrence and the 2^67 subsequence jump against rocRAND's precomputed tables
    (/opt/rocm/include/rocrand/rocrand_xorwow_precomputed.h: a second implementation of the same generator; its
    SEEDING differs from cuRAND's, which is why the library does not simply call rocRAND);
  - jumps against plain stepping;
  - the library's states against the oracle's;
  - the uniform mapping's range and first moments.
"""
import os
import re

import numpy as np
import pytest

import _oracle as O
from raytracing_amd import capi

ROCRAND_TABLES = "/opt/rocm/include/rocrand/rocrand_xorwow_precomputed.h"


def _apply(m, v):
    r = np.zeros(5, np.uint32)
    for w in range(5):
        for j in range(32):
            if (int(v[w]) >> j) & 1:
                r ^= m[w * 32 + j]
    return r


def _rocrand_table(name):
    txt = open(ROCRAND_TABLES).read()
    i = txt.index(name + "[XORWOW_JUMP_MATRICES][XORWOW_SIZE] = {")
    body = txt[txt.index("{", i):txt.index("};", i)]
    return np.array(re.findall(r"\d+", body), dtype=np.uint64).astype(np.uint32).reshape(-1, 160, 5)


@pytest.mark.skipif(not os.path.exists(ROCRAND_TABLES), reason="rocRAND headers not installed")
def test_jump_matrices_equal_rocrands():
    seq = _rocrand_table("h_xorwow_sequence_jump_matrices")   # A^(4^i * 2^67)
    jmp = _rocrand_table("h_xorwow_jump_matrices")            # A^(4^i)
    for i in (0, 1, 2, 7, 15):
        assert np.array_equal(seq[i], O.xorwow_pow2_matrix(67 + 2 * i)), i
        assert np.array_equal(jmp[i], O.xorwow_pow2_matrix(2 * i)), i
    # subsequence n = 1 + 4 + 4^7 through rocRAND's tables == the oracle's curand_init
    s0 = O.xorwow_init(1984, 0)
    n = 1 + 4 + 4 ** 7
    v = _apply(seq[7], _apply(seq[1], _apply(seq[0], s0[:5])))
    sn = O.xorwow_init(1984, n)
    assert np.array_equal(sn[:5], v) and sn[5] == s0[5]


def test_seeding_constants_and_first_draws():
    """curand_init(1984, 0, 0): the seed scramble of curand_kernel.h written out by hand, then five draws of
    Marsaglia's xorwow by hand."""
    M = 0xFFFFFFFF
    seed = 1984
    s0, s1 = (seed & M) ^ 0xaad26b49, (seed >> 32) ^ 0xf7dcefdd
    t0, t1 = (1099087573 * s0) & M, (2591861531 * s1) & M
    v = [(123456789 + t0) & M, 362436069 ^ t0, (521288629 + t1) & M, 88675123 ^ t1, (5783321 + t0) & M]
    d = (6615241 + t1 + t0) & M
    st = O.xorwow_init(seed, 0)
    assert list(map(int, st)) == v + [d]
    raw = []
    for _ in range(5):
        t = v[0] ^ (v[0] >> 2)
        v = v[1:] + [(v[4] ^ ((v[4] << 4) & M)) ^ (t ^ ((t << 1) & M))]
        d = (d + 362437) & M
        raw.append((v[4] + d) & M)
    st2, got_raw, uni = O.xorwow_draw(st, 5)
    assert list(map(int, got_raw)) == raw and list(map(int, st2)) == v + [d]
    want = (np.array(raw, np.uint32).astype(np.float32) * np.float32(2.3283064e-10)
            + np.float32(2.3283064e-10) / np.float32(2.0))
    assert np.array_equal(uni, want)


def test_jump_equals_stepping():
    st = O.xorwow_init(12345, 3)
    after, _, _ = O.xorwow_draw(st, 4096)
    assert np.array_equal(after[:5], _apply(O.xorwow_pow2_matrix(12), st[:5]))
    assert int(after[5]) == (int(st[5]) + 362437 * 4096) % 2 ** 32


def test_library_states_equal_oracle_states():
    rng = np.random.default_rng(5)
    subs = [0, 1, 2, 63, 64, 2073599, 8294399, 2 ** 28 - 1, 2 ** 32 - 1] + [int(x) for x in rng.integers(0, 2 ** 32, 40)]
    for seed in (1984, 0, 7, 2 ** 31 - 1):
        for sub in subs:
            assert np.array_equal(capi.rng_state(seed, sub), O.xorwow_init(seed, sub)), (seed, sub)


def test_uniform_range_and_moments():
    """curand_uniform is (0, 1]; streams of neighbouring pixels are unrelated."""
    st = O.xorwow_init(1984, 0)
    _, raw, u = O.xorwow_draw(st, 200000)
    assert u.min() > 0.0 and u.max() <= 1.0
    assert abs(u.mean() - 0.5) < 0.005 and abs(u.var() - 1.0 / 12.0) < 0.002
    _, _, u1 = O.xorwow_draw(O.xorwow_init(1984, 1), 200000)
    assert abs(np.corrcoef(u, u1)[0, 1]) < 0.01
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.01
    # x = 2^32 - 1 converts to 2^32 and maps to exactly 1.0; x = 0 to 2^-33
    f = lambda x: np.float32(np.uint32(x)) * np.float32(2.3283064e-10) + np.float32(2.3283064e-10) / np.float32(2.0)
    assert f(2 ** 32 - 1) == np.float32(1.0) and f(0) == np.float32(2.0 ** -33)


def test_oracle_diffuse_bounce_properties():
    """Oracle only: bounces change the hit pixels and nothing else; the frame is a function of
    (seed, pixel index in the full frame): a row range rendered on its own equals the same rows of the frame."""
    W, H = 96, 48
    cam = O.default_viewport(W / H)
    sc = O.Scene.default()
    d0 = sc.render(cam, W, H, max_depth=0, nthreads=4)
    d8 = sc.render(cam, W, H, max_depth=8, nthreads=4)
    assert np.array_equal(d0["id"], d8["id"])
    hit = d0["id"] >= 0
    assert (d8["rgb"][hit] != d0["rgb"][hit]).any() and (d8["rgb"] >= 0).all() and np.isfinite(d8["rgb"]).all()
    assert np.array_equal(d8["rgb"][~hit], d0["rgb"][~hit])     # sky pixels draw nothing
    part = sc.render(cam, W, H, rows=(16, 32), max_depth=8, nthreads=2)
    assert np.array_equal(part["rgb"][16:32], d8["rgb"][16:32])
    sc.set_rng_seed(5)
    assert (sc.render(cam, W, H, max_depth=8, nthreads=4)["rgb"] != d8["rgb"]).any()
