"""The CPU restatement against the reference's own GpuIntegrate body compiled for the host
(oracle/_ref, `make -C oracle ref`; exists only where /root/reference was present at build
time).  Random grids, poses and depth images: bit-exact."""
import numpy as np
import pytest

from oracle.oracle import Ref
from semantic_slam_amd import synth

pytestmark = pytest.mark.skipif(not Ref.available(), reason="oracle/_ref not built (no /root/reference here)")


@pytest.mark.parametrize("seed", range(8))
def test_oracle_equals_reference_body(oracle, seed):
    ref = Ref()
    rng = np.random.default_rng(seed)
    dims = tuple(int(x) for x in rng.integers(8, 60, 3))
    vs = float(rng.choice([0.004, 0.01, 0.02, 0.05]))
    trunc = float(np.float32(vs) * np.float32(5))
    origin = synth.surf_volume(max(dims), vs, z0=float(rng.uniform(-0.2, 2.0)))
    h, w = int(rng.integers(30, 120)), int(rng.integers(40, 160))
    K = np.array([rng.uniform(50, 150), 0, w / 2 + rng.uniform(-3, 3), 0, rng.uniform(50, 150),
                  h / 2 + rng.uniform(-3, 3), 0, 0, 1], np.float32)
    sc = synth.SurfScene(dims, vs, origin, K=K, h=h, w=w)
    t1, w1 = oracle.init_grid(dims)
    t2, w2 = t1.copy(), w1.copy()
    for k in range(4):
        base = synth.identity_pose() if seed % 2 == 0 else synth.random_pose(rng)
        c2b = oracle.cam2base(base, synth.random_pose(rng, 0.5, 0.5))
        depth = sc.depth(c2b, quantize=bool(k & 1))
        depth[rng.integers(0, h, 40), rng.integers(0, w, 40)] = rng.choice([0.0, -1.0, 6.5, 6.0], 40)
        oracle.integrate(K, c2b, depth, dims, origin, vs, trunc, t1, w1, threads=3)
        ref.integrate(K, c2b, depth, dims, origin, vs, trunc, t2, w2, threads=2)
    assert np.array_equal(w1, w2)
    assert np.array_equal(t1.view(np.uint32), t2.view(np.uint32))


def test_degenerate_poses_match(oracle):
    """Singular / huge / tiny pose entries: both sides take the same branches, no crash."""
    ref = Ref()
    dims, vs = (12, 12, 12), 0.05
    origin = np.array([-0.3, -0.3, -0.3], np.float32)
    depth = np.full((48, 64), 1.0, np.float32)
    K = np.array([50, 0, 32, 0, 50, 24, 0, 0, 1], np.float32)
    for c2b in (np.zeros(16, np.float32),
                synth.make_pose(np.eye(3) * 1e-30, [0, 0, 0]),
                synth.make_pose(np.eye(3) * 1e30, [0, 0, 0]),
                synth.make_pose(np.eye(3), [0, 0, 1e20])):
        t1, w1 = oracle.init_grid(dims)
        t2, w2 = t1.copy(), w1.copy()
        oracle.integrate(K, c2b, depth, dims, origin, vs, 0.25, t1, w1)
        ref.integrate(K, c2b, depth, dims, origin, vs, 0.25, t2, w2)
        assert np.array_equal(w1, w2) and np.array_equal(t1.view(np.uint32), t2.view(np.uint32))


def test_non_finite_depth_samples_match(oracle):
    """NaN passes both of the reference's depth tests (ref: src/tsdf.cu:46,49 -- every comparison with NaN is false)
    and updates the voxel with dist = fmin(1, NaN) = 1; +-inf, negative and denormal samples take the ordinary
    branches.  The restatement must do exactly the same."""
    ref = Ref()
    rng = np.random.default_rng(11)
    dims, vs = (40, 36, 30), 0.02
    origin = synth.surf_volume(max(dims), vs, 0.6)
    sc = synth.SurfScene(dims, vs, origin)
    t1, w1 = oracle.init_grid(dims)
    t2, w2 = t1.copy(), w1.copy()
    odd = np.array([np.nan, np.inf, -np.inf, -0.0, 1e-42, -1e-42, 6.0, np.nextafter(np.float32(6.0), np.float32(7.0))], np.float32)
    for k in range(4):
        c2b = sc.pose(k, 8)
        depth = sc.depth(c2b, quantize=True)
        depth[rng.integers(0, 480, 20000), rng.integers(0, 640, 20000)] = rng.choice(odd, 20000)
        oracle.integrate(synth.TUM_K, c2b, depth, dims, origin, vs, 0.1, t1, w1, threads=3)
        ref.integrate(synth.TUM_K, c2b, depth, dims, origin, vs, 0.1, t2, w2, threads=2)
    assert np.array_equal(w1, w2) and np.array_equal(t1.view(np.uint32), t2.view(np.uint32))
    assert np.isfinite(t1).all() and w1.max() >= 3
