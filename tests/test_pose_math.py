"""4x4 helpers (ref: src/tsdf.cu:253-273 multiply_matrix, :276-403 invert_matrix).

These are member functions of a class whose header needs OpenCV + CUDA, so the reference's own
code cannot be built for them: PARITY UNPINNED by reference output.  They are checked by
known-answer tests, by an independent numpy-float32 emulation of the same operation order, and
the product's host implementation (csrc/pose_math.h, exported by libtsdf_hip.so) is checked
bit-for-bit against the oracle's independently written one.
"""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

f32 = np.float32


def emulate_multiply(a, b):
    a, b = a.astype(f32).reshape(4, 4), b.astype(f32).reshape(4, 4)
    out = np.empty((4, 4), f32)
    for i in range(4):
        for j in range(4):
            s = f32(a[i, 0] * b[0, j])
            for k in range(1, 4):
                s = f32(s + f32(a[i, k] * b[k, j]))
            out[i, j] = s
    return out.ravel()


def test_multiply_known_answers(oracle):
    I = np.eye(4, dtype=f32).ravel()
    rng = np.random.default_rng(0)
    A = rng.normal(size=16).astype(f32)
    assert np.array_equal(oracle.multiply(I, A), A)
    assert np.array_equal(oracle.multiply(A, I), A)
    B = rng.normal(size=16).astype(f32)
    assert np.array_equal(oracle.multiply(A, B), emulate_multiply(A, B))
    # left-to-right summation is observable: (1e8 + 1) - 1e8 in fp32
    a = np.zeros(16, f32); b = np.zeros(16, f32)
    a[0:4] = [1e8, 1.0, -1e8, 0.0]
    b[0], b[4], b[8] = 1.0, 1.0, 1.0
    assert oracle.multiply(a, b)[0] == f32(f32(f32(1e8) + f32(1.0)) - f32(1e8))


def test_invert_known_answers(oracle):
    ok, inv = oracle.invert(np.eye(4, dtype=f32).ravel())
    assert ok and np.array_equal(inv, np.eye(4, dtype=f32).ravel())
    ok, inv = oracle.invert(np.zeros(16, f32))
    assert not ok and np.all(inv == 0)          # ref: src/tsdf.cu:394-395, output untouched
    D = np.diag([2.0, 4.0, 0.5, 1.0]).astype(f32).ravel()
    ok, inv = oracle.invert(D)
    assert ok and np.array_equal(inv, np.diag([0.5, 0.25, 2.0, 1.0]).astype(f32).ravel())
    rng = np.random.default_rng(1)
    for _ in range(20):
        T = synth.random_pose(rng, 1.0, 2.0)
        ok, inv = oracle.invert(T)
        assert ok
        R, t = T.reshape(4, 4)[:3, :3].astype(np.float64), T.reshape(4, 4)[:3, 3].astype(np.float64)
        want = np.eye(4); want[:3, :3] = R.T; want[:3, 3] = -R.T @ t
        assert np.allclose(inv.reshape(4, 4), want, atol=2e-6)
        assert np.allclose(oracle.multiply(T, inv).reshape(4, 4), np.eye(4), atol=2e-6)


def test_cam2base_composition(oracle):
    rng = np.random.default_rng(2)
    base, cam = synth.random_pose(rng), synth.random_pose(rng)
    ok, inv = oracle.invert(base)
    assert np.array_equal(oracle.cam2base(base, cam), oracle.multiply(inv, cam))
    # singular base: the reference ignores the failure and multiplies by an all-zero inverse
    assert np.array_equal(oracle.cam2base(np.zeros(16, f32), cam), oracle.multiply(np.zeros(16, f32), cam))


def test_product_pose_math_equals_oracle_bitwise(oracle):
    """csrc/pose_math.h (table-driven cofactors) vs oracle/tsdf_oracle.c (spelled-out formulas)."""
    rng = np.random.default_rng(3)
    for k in range(300):
        if k % 3 == 0:
            A = synth.random_pose(rng, 3.0, 5.0)
        elif k % 3 == 1:
            A = rng.normal(size=16).astype(f32)
        else:
            A = (rng.normal(size=16) * 10.0 ** rng.integers(-6, 6, 16)).astype(f32)
        B = rng.normal(size=16).astype(f32)
        assert np.array_equal(capi.multiply_matrix(A, B).view(np.uint32), oracle.multiply(A, B).view(np.uint32))
        ok1, i1 = capi.invert_matrix(A)
        ok2, i2 = oracle.invert(A)
        assert ok1 == ok2
        assert np.array_equal(i1.view(np.uint32), i2.view(np.uint32))
    assert capi.invert_matrix(np.zeros(16, f32))[0] is False
