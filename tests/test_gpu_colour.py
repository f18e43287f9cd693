"""Per-voxel colour fusion behind the TSDFfusion surface (csrc/tsdf_colour.hip.h): the HIP pass against this project's
CPU restatement of tsdf-fusion-python's published rule (oracle_integrate_colour) -- bit for bit.  The package itself is
absent, so parity with the reference's Python backend is unpinned; the known-answer cases live in
tests/test_oracle_extras.py."""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def images(h, w, k):
    vv, uu = np.mgrid[0:h, 0:w]
    return np.stack([(uu * 3 + 50 * k) % 256, (vv * 2 + 70 * k) % 256, (uu + 2 * vv + 31 * k) % 256], axis=-1).astype(np.uint8)


@pytest.mark.parametrize("dims,vs", [((256, 40, 24), 0.004), ((200, 60, 30), 0.005), ((36, 20, 12), 0.03)])
def test_colour_pass_matches_oracle(cuda, oracle, dims, vs):
    origin = synth.surf_volume(dims[0], vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)
    ref_c = np.zeros(ref_t.size, np.uint32)
    with capi.Volume(cfg) as vol:
        vol.colour_enable()
        for k in range(6):
            pose = scene.pose(k % 4, 7)
            depth = scene.depth(pose, quantize=True)
            rgb = images(480, 640, k)
            oracle.integrate(cfg.cam_K, pose, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
            n = oracle.integrate_colour(cfg.cam_K, pose, depth, rgb, dims, origin, vs, cfg.trunc_margin, ref_w, ref_c)
            if k % 2 == 0:     # host images through the pinned staging ring
                vol.integrate_rgbd(depth, rgb, pose)
            else:              # device-resident images: Integrate, then the colour pass of the same frame
                d_dev, c_dev = cuda.from_numpy(depth).cuda(), cuda.from_numpy(rgb).cuda()
                vol.integrate_device(d_dev.data_ptr(), pose)
                vol.integrate_colour_device(d_dev.data_ptr(), c_dev.data_ptr(), pose)
                vol.sync()
            assert n > 0
        t, w = vol.download()
        c = vol.download_colour()
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
    assert ref_w.max() >= 3 and np.count_nonzero(ref_c) > 100
    assert np.array_equal(c, ref_c), f"{np.count_nonzero(c != ref_c)} packed colours differ"
    assert np.all(c[ref_w == 0] == 0)          # a voxel no frame updated keeps colour 0


def test_colour_needs_enabling(cuda):
    cfg = capi.make_config((16, 16, 8), 0.01, [0, 0, 1])
    with capi.Volume(cfg) as vol:
        with pytest.raises(capi.TsdfError, match="tsdf_colour_enable"):
            vol.integrate_rgbd(np.ones((480, 640), np.float32), np.zeros((480, 640, 3), np.uint8), synth.identity_pose())


@pytest.mark.parametrize("dims", [(256, 40, 24), (200, 60, 30)])
def test_colour_touches_exactly_the_voxels_integrate_updated(cuda, oracle, dims):
    """The colour pass re-derives which voxels the frame updates with its own copy of the projection (one row per wavefront;
    Integrate's one-frame kernel decides fast or generic projection per two-row wavefront).  The two must agree voxel for
    voxel or colour and weight drift apart silently: a camera INSIDE the volume (camera-frame z crosses zero inside rows,
    so wavefronts of both kernels fall back to the generic projection in different places), every colour channel >= 1,
    one frame into a fresh volume: colour != 0 exactly where the weight became 1."""
    vs = 0.005
    origin = synth.surf_volume(dims[0], vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    centre = origin + np.array(dims, np.float32) * vs / 2
    for k, pose in enumerate([synth.make_pose(synth.rot_y(0.3) @ synth.rot_x(-0.2), centre + np.array([0.01, -0.02, -0.03], np.float32)),
                              synth.make_pose(synth.rot_y(1.2), centre + np.array([-0.2, 0.0, 0.0], np.float32)),
                              synth.make_pose(np.eye(3), [0.0, float(centre[1]), 0.0])]):     # in front of the slab, fast projection everywhere
        depth = np.full((480, 640), 0.9 + 0.1 * k, np.float32)
        depth[::7, ::5] = 0.0
        rgb = np.maximum(images(480, 640, k), 1)
        with capi.Volume(cfg) as vol:
            vol.colour_enable()
            vol.set_deferral(0)
            d_dev, c_dev = cuda.from_numpy(depth).cuda(), cuda.from_numpy(rgb).cuda()
            vol.integrate_device(d_dev.data_ptr(), pose)
            vol.integrate_colour_device(d_dev.data_ptr(), c_dev.data_ptr(), pose)
            _, w = vol.download()
            c = vol.download_colour()
        rt, rw = oracle.init_grid(dims)
        oracle.integrate(cfg.cam_K, pose, depth, dims, origin, vs, cfg.trunc_margin, rt, rw)
        assert np.array_equal(w, rw) and 100 < np.count_nonzero(w) < w.size
        assert np.array_equal(c != 0, w == 1.0), f"pose {k}: {np.count_nonzero((c != 0) != (w == 1.0))} voxels coloured without an update or updated without a colour"
