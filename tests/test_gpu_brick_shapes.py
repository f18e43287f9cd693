"""The wavefront brick of the classified launches (tsdf_set_brick_shape: q quads x r rows x s slices) is a tuning knob:
whatever the shape -- planar, spanning slices, lanes left idle, rows / slices that do not divide into bricks -- fused
sequences, single masked frames, batched object volumes and label-fusing launches equal the oracle bit for bit, and
claims are made (so the test is about classified wavefronts, not about a path that never classifies)."""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu

# (dims, shapes): 64 / 50 / 9 quads per row; dim_y and dim_z chosen so that rows and slices leave partial bricks
CASES = [
    ((256, 42, 30), [(8, 8, 1), (16, 4, 1), (4, 4, 4), (2, 4, 8), (8, 2, 4), (1, 1, 64), (64, 1, 1), (4, 5, 3), (1, 7, 9)]),
    ((200, 45, 21), [(10, 6, 1), (5, 12, 1), (5, 6, 2), (2, 4, 8), (2, 5, 6), (25, 2, 1), (50, 1, 1), (1, 8, 8), (5, 1, 12)]),
    ((36, 20, 12), [(9, 7, 1), (3, 4, 5), (1, 4, 16), (9, 1, 7)]),
]


def scene_frames(dims, vs, origin, n_frames, seed):
    scene = synth.SurfScene(dims, vs, origin)
    rng = np.random.default_rng(seed)
    frames = []
    for k in range(n_frames):
        c2w = scene.pose(k % 5, n=7)
        depth = np.full((480, 640), 5.9, np.float32) if k % 3 == 2 else scene.depth(c2w, quantize=True)
        mask = None
        if k % 4 == 1:
            mask = np.zeros((480, 640), np.uint8)
            mask[rng.integers(50, 150):rng.integers(300, 450), rng.integers(50, 200):rng.integers(400, 600)] = 255
        frames.append((c2w, depth, mask))
    return frames


@pytest.mark.parametrize("dims,shapes", CASES)
def test_fused_sequences_do_not_depend_on_the_brick_shape(cuda, oracle, dims, shapes):
    vs = 2.0 / dims[0]
    origin = synth.surf_volume(dims[0], vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    frames = scene_frames(dims, vs, origin, 9, 3)
    ref_t, ref_w = oracle.init_grid(dims)
    for c2w, depth, mask in frames:
        d = depth if mask is None else oracle.mask_depth(depth, mask)
        oracle.integrate(cfg.cam_K, c2w, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
    keep = [(cuda.from_numpy(d).cuda(), None if m is None else cuda.from_numpy(m).cuda()) for _, d, m in frames]
    for shape in shapes:
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(8)              # classify whatever the launch size
            vol.set_brick_shape(*shape)
            assert vol.brick_shape() == shape
            vol.shortcut_stats(True)
            vol.integrate_frames_device([d.data_ptr() for d, _ in keep], np.stack([f[0] for f in frames]),
                                        [None if m is None else m.data_ptr() for _, m in keep])
            per_voxel, free, skipped = vol.shortcut_stats(False)
            assert free + skipped > 0 and per_voxel > 0, shape
            t, w = vol.download()
        assert np.array_equal(w, ref_w), f"{shape}: weights differ"
        assert np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)), f"{shape}: TSDF differs"


@pytest.mark.parametrize("dims,shapes", CASES[:2])
def test_single_masked_frames_do_not_depend_on_the_brick_shape(cuda, oracle, dims, shapes):
    """tsdf_integrate_masked_device on a flat-mapped grid: class table per brick, then integrate_single_bricks."""
    vs = 2.0 / dims[0]
    origin = synth.surf_volume(dims[0], vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    frames = [(p, d, m) for p, d, m in scene_frames(dims, vs, origin, 6, 5)]
    rect = np.zeros((480, 640), np.uint8)
    rect[140:330, 210:470] = 255
    frames = [(p, d, rect if m is None else m) for p, d, m in frames]
    ref_t, ref_w = oracle.init_grid(dims)
    for c2w, depth, mask in frames:
        oracle.integrate(cfg.cam_K, c2w, oracle.mask_depth(depth, mask), dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
    keep = [(cuda.from_numpy(d).cuda(), cuda.from_numpy(m).cuda()) for _, d, m in frames]
    for shape in shapes:
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(8)
            vol.set_deferral(0)                    # one launch per call
            vol.set_brick_shape(*shape)
            for (c2w, _, _), (d, m) in zip(frames, keep):
                vol.integrate_masked_device(d.data_ptr(), m.data_ptr(), c2w)
            t, w = vol.download()
        assert np.array_equal(w, ref_w), f"{shape}: weights differ"
        assert np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)), f"{shape}: TSDF differs"


@pytest.mark.parametrize("deferred", [False, True])   # one batched launch per frame (group map) / one fused launch per member
def test_batched_objects_with_their_own_brick_shapes(cuda, oracle, deferred):
    """Every member of a batch brings its own brick (and its own number of slice groups); changing a member's shape
    between frames rebuilds the launch's group map."""
    rng = np.random.default_rng(11)
    specs = [((200, 45, 21), 0.004, (-0.40, -0.10, 0.80), (2, 4, 8)),
             ((64, 48, 40), 0.010, (-0.30, -0.20, 0.90), (4, 4, 4)),
             ((128, 32, 17), 0.008, (-0.50, -0.10, 1.10), (8, 8, 1)),
             ((36, 20, 12), 0.020, (-0.35, -0.20, 0.90), (3, 4, 5))]
    K = synth.TUM_K
    cfgs = [capi.make_config(d, vs, np.array(o, np.float32), vol_id=i) for i, (d, vs, o, _) in enumerate(specs)]
    scene = synth.SurfScene((200, 200, 200), 0.004, np.array([-0.4, -0.4, 0.7], np.float32))
    masks = []
    for d, vs, o, _ in specs:
        c = np.array(o) + np.array(d) * vs / 2
        m = np.zeros((480, 640), np.uint8)
        u0, u1 = K[0] * (c[0] - 0.15) / c[2] + K[2], K[0] * (c[0] + 0.15) / c[2] + K[2]
        v0, v1 = K[4] * (c[1] - 0.1) / c[2] + K[5], K[4] * (c[1] + 0.1) / c[2] + K[5]
        m[max(0, int(v0)):max(0, min(480, int(v1))), max(0, int(u0)):max(0, min(640, int(u1)))] = 255
        masks.append(m)
    refs = [oracle.init_grid(d) for d, _, _, _ in specs]
    with capi.Batch(cfgs) as batch:
        for vol, (_, _, _, shape) in zip(batch.volumes, specs):
            vol.set_kernel_variant(8)
            vol.set_brick_shape(*shape)
        if not deferred:
            batch.volumes[0].set_deferral(0)
        m_dev = [cuda.from_numpy(m).cuda() for m in masks]
        keep = []
        for k in range(4):
            if k == 2:                          # new shapes mid-sequence
                batch.volumes[0].set_brick_shape(5, 6, 2)
                batch.volumes[2].set_brick_shape(0, 0, 0)
            c2w = scene.pose(k, n=6)
            depth = scene.depth(c2w, quantize=True)
            d_dev = cuda.from_numpy(depth).cuda()
            keep.append(d_dev)
            batch.integrate_device(d_dev.data_ptr(), [t.data_ptr() for t in m_dev], c2w)
            if not deferred:
                batch.sync()
            for (d, vs, o, _), m, (rt, rw), cfg in zip(specs, masks, refs, cfgs):
                oracle.integrate(cfg.cam_K, c2w, oracle.mask_depth(depth, m), d, np.array(o, np.float32), vs, cfg.trunc_margin, rt, rw)
        for vol, (rt, rw) in zip(batch.volumes, refs):
            t, w = vol.download()
            assert np.array_equal(w, rw), f"volume {vol.cfg.id}: weights differ"
            assert np.array_equal(t.view(np.uint32), rt.view(np.uint32)), f"volume {vol.cfg.id}: TSDF differs"
        assert sum(int(rw.sum()) for _, rw in refs) > 10000


def test_shape_validation():
    with capi.Volume(capi.make_config((200, 8, 8), 0.01, [0, 0, 1])) as vol:
        q, r, s = vol.brick_shape()
        assert q > 0 and 50 % q == 0 and q * r * s <= 64
        for bad in [(3, 4, 4), (8, 8, 2), (0, 4, 4), (5, 0, 1), (5, 13, 1), (-5, 6, 2)]:
            with pytest.raises(capi.TsdfError, match="tsdf_set_brick_shape"):
                vol.set_brick_shape(*bad)
        vol.set_brick_shape(25, 2, 1)
        assert vol.brick_shape() == (25, 2, 1)
        vol.set_brick_shape(0, 0, 0)
        assert vol.brick_shape() == (q, r, s)
    with capi.Volume(capi.make_config((7, 8, 8), 0.01, [0, 0, 1])) as vol:      # scalar kernel: no brick view
        assert vol.brick_shape() == (0, 0, 0)


def test_wide_flat_slices_fall_back_to_memory_order_on_the_per_voxel_path(cuda, oracle):
    """The unclassified fused launch (variant 7) of a flat-mapped volume whose slice holds more than 65 535 workgroups: slices
    fastest would put the workgroups of a slice into a slow grid dimension, so the launch must go out in memory order instead of
    failing (or faulting) -- 8 200 x 8 200 voxels per slice = 65 664 workgroups.  Same bits as the oracle."""
    dims, vs = (8200, 8200, 2), 0.0002
    origin = np.array([-dims[0] * vs / 2, -dims[1] * vs / 2, 1.5], np.float32)
    cfg = capi.make_config(dims, vs, origin)
    assert dims[0] % 256 != 0 and (dims[0] * dims[1] // 256 + 3) // 4 > 65535
    far = float(origin[2]) + dims[2] * vs
    poses = [synth.identity_pose(), synth.make_pose(synth.rot_z(0.02), [0.003, -0.002, 0.0]), synth.make_pose(synth.rot_y(0.01), [0.0, 0.004, 0.0])]
    depths = [np.full((480, 640), far + 0.2, np.float32) for _ in poses]
    depths[1][:, 320:] = float(origin[2]) + 0.5 * dims[2] * vs          # a surface through the right half of the slab
    depths[2][:240, :] = float(origin[2]) - 0.3                         # in front of the slab: nothing updated up there
    ref_t, ref_w = oracle.init_grid(dims)
    for p, d in zip(poses, depths):
        oracle.integrate(cfg.cam_K, p, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, threads=16)
    keep = [cuda.from_numpy(d).cuda() for d in depths]
    for variant in (7, 8):
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(variant)
            vol.integrate_frames_device([d.data_ptr() for d in keep], np.stack(poses))
            t, w = vol.download()
        assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)), variant
    assert ref_w.max() == 3 and ref_w.min() < 3
