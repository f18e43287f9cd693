"""The synthetic workloads do what SURVEY.md section 8(d) says they do, checked with the oracle."""
import numpy as np
import pytest

from semantic_slam_amd import synth


@pytest.mark.parametrize("D,vs", [(512, 0.005), (1024, 0.002)])
def test_sfull_updates_every_voxel(oracle, D, vs):
    """N_upd == N on slab samples (corners of the z range are the worst case for the frustum)."""
    origin = synth.sfull_volume(D, vs)
    depth = synth.sfull_depth()
    trunc = float(np.float32(vs) * np.float32(5))
    for zb in (0, D - 2):
        t, w = oracle.init_grid((D, D, D), zb, zb + 2)
        for k in (0, 15, 16, 47):   # includes the largest roll (sin(0.1k) ~ +-1)
            n = oracle.integrate(synth.TUM_K, synth.sfull_pose(k), depth, (D, D, D), origin, vs, trunc,
                                 t, w, z_begin=zb, z_end=zb + 2)
            assert n == 2 * D * D
        assert np.all(t == 1.0) and np.all(w == 4.0)


def test_surf_scene_mixes_branches(oracle):
    dims, vs = (64, 64, 64), 0.04
    origin = synth.surf_volume(64, vs, 1.0)
    sc = synth.SurfScene(dims, vs, origin)
    t, w = oracle.init_grid(dims)
    p = sc.pose(5)
    d = sc.depth(p, quantize=True)
    assert d.min() == 0.0 and 1.0 < d.max() <= 6.0
    n = oracle.integrate(synth.TUM_K, p, d, dims, origin, vs, 0.2, t, w)
    assert 0.05 * t.size < n < 0.9 * t.size
    assert np.count_nonzero(t < 1.0) > 100 and t.min() < 0.0   # inside the truncation band on both sides
    assert np.allclose(sc.pose(0), sc.pose(64), atol=1e-6)     # orbit is periodic


@pytest.mark.parametrize("D,vs", [(512, 0.005), (1024, 0.002)])
def test_sband_updates_every_voxel_inside_the_band(oracle, D, vs):
    """S-band: N_upd == N, every dist < 1 (inside the truncation band), and the TSDF values keep changing from
    frame to frame -- no update of it can be elided.  Slab samples at both ends of the z range."""
    origin = synth.sband_volume(D, vs)
    depth = synth.sfull_depth()
    for zb in (0, D - 2):
        t, w = oracle.init_grid((D, D, D), zb, zb + 2)
        prev = t.copy()
        for i, k in enumerate((0, 4, 15, 16, 47)):   # largest roll, both signs of the axial wobble
            n = oracle.integrate(synth.TUM_K, synth.sband_pose(k), depth, (D, D, D), origin, vs, synth.SBAND_TRUNC,
                                 t, w, z_begin=zb, z_end=zb + 2)
            assert n == 2 * D * D
            assert t.max() < 1.0 and t.min() > 0.0
            if i > 0:   # the first frame sets dist itself; every later one moves (almost) every value
                assert np.count_nonzero(t.view(np.uint32) != prev.view(np.uint32)) > 0.99 * t.size
            prev = t.copy()
        assert np.all(w == 5.0)
