"""4x4 helpers (ref: src/tsdf.cu:253-273 multiply_matrix, :276-403 invert_matrix).

They are members of a class whose header needs OpenCV + CUDA, so the class cannot be built here.  Their two definitions are
plain arithmetic, though, and `make -C oracle ref_host` compiles them as they stand (behind two prototypes at namespace
scope, oracle/ref_host_driver.cpp) into oracle/_ref/libtsdf_ref_host.so: the tests at the bottom of this file hold the
oracle's and the product's implementations, and the pose composition of TSDF::TSDF / TSDF::Integrate (src/tsdf.cu:74,142), to
that build bit for bit.  Above them: known-answer tests, an independent numpy-float32 emulation of the same operation
order, and the product's host implementation (csrc/pose_math.h, exported by libtsdf_hip.so) against the oracle's
independently written one.
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


# ---- against the reference's own two functions, compiled as they stand (oracle/_ref/libtsdf_ref_host.so) ------------------
from oracle.oracle import RefHost  # noqa: E402

needs_ref_host = pytest.mark.skipif(not RefHost.available(), reason="oracle/_ref/libtsdf_ref_host.so not built (make -C oracle ref_host)")


def _matrices(rng, n):
    """Rigid poses, general matrices over many magnitudes, near-singular and exactly singular ones, special values."""
    out = []
    for k in range(n):
        kind = k % 8
        if kind in (0, 1):
            m = synth.random_pose(rng, 0.8, 3.0).reshape(4, 4)
        elif kind == 2:
            m = rng.standard_normal((4, 4)) * 10.0 ** rng.integers(-6, 7)
        elif kind == 3:
            m = rng.standard_normal((4, 4)) * 10.0 ** rng.integers(-6, 7, (4, 4))
        elif kind == 4:                       # near-singular: two almost equal rows
            m = rng.standard_normal((4, 4))
            m[2] = m[1] * (1.0 + rng.standard_normal() * 1e-6)
        elif kind == 5:                       # exactly singular: det == 0 on the reference's own operation order too
            m = rng.integers(-3, 4, (4, 4)).astype(np.float64)
            m[3] = m[0]
        elif kind == 6:                       # denormals and huge values
            m = rng.standard_normal((4, 4)) * np.where(rng.random((4, 4)) < 0.5, 1e-40, 1e18)
        else:                                 # integers: exact arithmetic
            m = rng.integers(-50, 51, (4, 4)).astype(np.float64)
        out.append(m.astype(f32).ravel())
    return out


def _same_bits(a, b):
    return np.array_equal(np.asarray(a, f32).view(np.uint32), np.asarray(b, f32).view(np.uint32))


@needs_ref_host
def test_multiply_equals_the_references_function_bit_for_bit(oracle):
    ref = RefHost()
    rng = np.random.default_rng(253)
    ms = _matrices(rng, 4000)
    for a, b in zip(ms[::2], ms[1::2]):
        want = ref.multiply(a, b)
        assert _same_bits(oracle.multiply(a, b), want) or (np.isnan(want).any() and np.array_equal(np.isnan(oracle.multiply(a, b)), np.isnan(want)))
        got = capi.multiply_matrix(a, b)
        assert _same_bits(got, want) or (np.isnan(want).any() and np.array_equal(np.isnan(got), np.isnan(want)))


@needs_ref_host
def test_invert_equals_the_references_function_bit_for_bit(oracle):
    ref = RefHost()
    rng = np.random.default_rng(276)
    n_singular = 0
    for m in _matrices(rng, 4000):
        ok_r, inv_r = ref.invert(m)
        ok_o, inv_o = oracle.invert(m)
        ok_p, inv_p = capi.invert_matrix(m)
        assert ok_o == ok_r and ok_p == ok_r
        if not ok_r:
            n_singular += 1
            continue
        for got in (inv_o, inv_p):
            nan = np.isnan(inv_r)
            assert np.array_equal(np.isnan(got), nan) and _same_bits(np.where(nan, 0, got), np.where(nan, 0, inv_r))
    assert n_singular > 100, "the singular branch (det == 0 -> false, ref: src/tsdf.cu:392-393) should be exercised"


@needs_ref_host
def test_pose_composition_equals_the_references_two_calls(oracle):
    """TSDF::TSDF inverts the base pose once (ref: src/tsdf.cu:74), TSDF::Integrate multiplies base2world_inv x cam2world
    (ref: src/tsdf.cu:142): the same two calls on the reference's own functions."""
    ref = RefHost()
    rng = np.random.default_rng(142)
    for _ in range(500):
        base, cam = synth.random_pose(rng, 0.8, 3.0), synth.random_pose(rng, 0.8, 3.0)
        ok, inv = ref.invert(base)
        assert ok
        assert _same_bits(oracle.cam2base(base, cam), ref.multiply(inv, cam))
