"""Zero-crossing surface vertices on the device (tsdf_extract_crossings) against the CPU
restatement of the same project-defined rule (oracle_zero_crossings; no reference function
exists for it -- parity unpinned by reference output), including the one-voxel z halo between
slabs: two slab handles fed each other's boundary slice reproduce the unsharded list exactly."""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def fused_scene(cuda, oracle, dims, vs, z_cuts):
    origin = synth.surf_volume(max(dims), vs, 0.6)
    scene = synth.SurfScene(dims, vs, origin)
    vols = [capi.Volume(capi.make_config(dims, vs, origin, z_begin=a, z_end=b)) for a, b in zip(z_cuts[:-1], z_cuts[1:])]
    ref_t, ref_w = oracle.init_grid(dims)
    for k in range(4):
        c2w = scene.pose(k, n=6)
        depth = scene.depth(c2w, quantize=True)
        for v in vols:
            v.integrate(depth, c2w)
        oracle.integrate(vols[0].cfg.cam_K, c2w, depth, dims, origin, vs, vols[0].cfg.trunc_margin, ref_t, ref_w)
    return origin, vols, ref_t, ref_w


def test_crossings_match_oracle(cuda, oracle):
    dims, vs = (96, 70, 52), 0.01
    origin, (vol,), ref_t, ref_w = fused_scene(cuda, oracle, dims, vs, [0, dims[2]])
    want = oracle.zero_crossings(ref_t, ref_w, dims[:2], 0, dims[2], vs, origin)
    got = vol.extract_crossings()
    vol.close()
    assert len(want) > 2000
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # every vertex lies within one voxel edge of the sphere/wall surface band
    assert np.all(np.isfinite(got))


def test_crossings_across_slabs_need_and_use_the_halo(cuda, oracle):
    dims, vs = (64, 48, 40), 0.012
    cuts = [0, 13, 27, 40]
    origin, vols, ref_t, ref_w = fused_scene(cuda, oracle, dims, vs, cuts)
    whole = oracle.zero_crossings(ref_t, ref_w, dims[:2], 0, dims[2], vs, origin)
    parts, parts_nohalo = [], []
    for i, v in enumerate(vols):
        halo = vols[i + 1].copy_slices(0, 1) if i + 1 < len(vols) else None
        parts.append(v.extract_crossings(halo))
        parts_nohalo.append(v.extract_crossings(None))
        if halo is not None:   # the same halo handed over as device memory (an RCCL receive buffer)
            ht, hw = cuda.from_numpy(halo[0]).cuda(), cuda.from_numpy(halo[1]).cuda()
            assert np.array_equal(v.extract_crossings((ht.data_ptr(), hw.data_ptr())), parts[-1])
    got = np.concatenate(parts)
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
    assert len(np.concatenate(parts_nohalo)) < len(whole), "the scene must have crossings on slab boundaries"
    for v in vols:
        v.close()
