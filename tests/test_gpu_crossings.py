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


def test_flat_segments_are_skipped_only_where_nothing_can_cross(cuda, oracle):
    """Grids with 256-voxel row segments (512-wide rows: two per row) take the early-out of the crossing and mesh kernels: a
    segment is skipped unread when the free-space summary says that it and every segment it is compared with hold 1.0.
    (a) a fused scene: crossings and mesh equal the oracle's, in three slabs with halos as well;  (b) uploaded volumes that
    are 1.0 everywhere except one voxel placed so that the sign change belongs to a NEIGHBOUR of the segment that holds it
    -- the last voxel of a segment (its -x neighbour's edge ends there... its own +x edge leaves the segment), the first
    voxel of a row, of a slice, of the next segment -- each must still produce exactly the oracle's vertices and triangles."""
    dims, vs = (512, 24, 12), 0.004
    origin, vols, ref_t, ref_w = fused_scene(cuda, oracle, dims, vs, [0, 5, 12])
    whole_x = oracle.zero_crossings(ref_t, ref_w, dims[:2], 0, dims[2], vs, origin)
    whole_m = oracle.mesh_triangles(ref_t, ref_w, dims[:2], 0, dims[2], vs, origin)
    assert len(whole_x) > 300 and len(whole_m) > 300
    px, pm = [], []
    for i, v in enumerate(vols):
        halo = vols[i + 1].copy_slices(0, 1) if i + 1 < len(vols) else None
        px.append(v.extract_crossings(halo))
        pm.append(v.extract_mesh(halo))
        v.close()
    assert np.array_equal(np.concatenate(px).view(np.uint32), whole_x.view(np.uint32))
    assert np.array_equal(np.concatenate(pm).view(np.uint32), whole_m.view(np.uint32))

    cfg = capi.make_config(dims, vs, origin)
    n = dims[0] * dims[1] * dims[2]
    slice_ = dims[0] * dims[1]
    spots = [(255, 3, 2), (256, 3, 2), (0, 4, 2), (511, 4, 2), (17, 0, 3), (17, 23, 3), (300, 5, 0), (300, 5, 11), (255, 23, 10), (256, 0, 1)]
    with capi.Volume(cfg) as vol:
        for x, y, z in spots:
            t = np.ones(n, np.float32)
            w = np.ones(n, np.float32)
            t[z * slice_ + y * dims[0] + x] = -0.5
            vol.upload(t, w)
            want_x = oracle.zero_crossings(t, w, dims[:2], 0, dims[2], vs, origin)
            want_m = oracle.mesh_triangles(t, w, dims[:2], 0, dims[2], vs, origin)
            assert len(want_x) >= 3, (x, y, z)
            got_x, got_m = vol.extract_crossings(), vol.extract_mesh()
            assert got_x.shape == want_x.shape and np.array_equal(got_x.view(np.uint32), want_x.view(np.uint32)), (x, y, z)
            assert got_m.shape == want_m.shape and np.array_equal(got_m.view(np.uint32), want_m.view(np.uint32)), (x, y, z)
