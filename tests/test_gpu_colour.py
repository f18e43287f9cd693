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
