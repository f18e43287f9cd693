"""Batched per-object fusion: n object volumes, each with its own grid, origin,
base pose and instance mask, integrated by ONE launch per frame -- against the oracle run per
object on depth * (mask/255) (ref: src/Engine.cpp:192-193, src/Object.cpp:143-166)."""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("deferred", [False, True])   # one batched launch per frame / frames collected, one fused launch per member
def test_batch_matches_per_object_oracle(cuda, oracle, deferred):
    rng = np.random.default_rng(42)
    specs = [  # dims, voxel size, grid origin (base frame), mask rectangle or None
        ((200, 200, 200), 0.004, (-0.40, -0.40, 0.70), (60, 420, 80, 560)),   # the reference's default grid
        ((64, 48, 40), 0.010, (-0.30, -0.20, 0.90), (100, 300, 200, 500)),
        ((128, 32, 16), 0.008, (-0.50, -0.10, 1.10), None),                   # unmasked member
        ((4, 4, 4), 0.050, (-0.10, -0.10, 1.00), (0, 480, 0, 640)),
        ((256, 8, 24), 0.004, (-0.51, 0.00, 0.80), (200, 280, 0, 640)),
    ]
    bases = [synth.random_pose(rng, 0.1, 0.2) for _ in specs]       # each object's first-keyframe pose
    cfgs = [capi.make_config(d, vs, np.array(o, np.float32), base2world=b, vol_id=i)
            for i, ((d, vs, o, _), b) in enumerate(zip(specs, bases))]
    scene = synth.SurfScene((200, 200, 200), 0.004, np.array([-0.4, -0.4, 0.7], np.float32))
    masks = []
    for _, _, _, rect in specs:
        if rect is None:
            masks.append(None)
        else:
            m = np.zeros((480, 640), np.uint8)
            m[rect[0]:rect[1], rect[2]:rect[3]] = 255
            masks.append(m)
    refs = [oracle.init_grid(d) for d, _, _, _ in specs]
    with capi.Batch(cfgs) as batch:
        if not deferred:
            batch.volumes[0].set_deferral(0)
        m_dev = [None if m is None else cuda.from_numpy(m).cuda() for m in masks]
        keep = []
        for k in range(3):
            c2w = scene.pose(k, n=6)
            depth = scene.depth(c2w, quantize=True)
            d_dev = cuda.from_numpy(depth).cuda()
            keep.append(d_dev)
            batch.integrate_device(d_dev.data_ptr(), [None if t is None else t.data_ptr() for t in m_dev], c2w)
            if not deferred:
                batch.sync()
            for (d, vs, o, _), b, m, (rt, rw), cfg in zip(specs, bases, masks, refs, cfgs):
                dm = depth if m is None else oracle.mask_depth(depth, m)
                oracle.integrate(cfg.cam_K, oracle.cam2base(b, c2w), dm, d, np.array(o, np.float32), vs,
                                 cfg.trunc_margin, rt, rw)
        total = 0
        for vol, (rt, rw) in zip(batch.volumes, refs):
            t, w = vol.download()
            assert np.array_equal(w, rw), f"volume {vol.cfg.id}: weights differ"
            assert np.array_equal(t.view(np.uint32), rt.view(np.uint32)), f"volume {vol.cfg.id}: TSDF differs"
            total += int(rw.sum())
        assert total > 100000
        assert batch.volumes[0].count_surface() == len(oracle.surface_points(refs[0][0], refs[0][1], specs[0][0], 0.004,
                                                                             np.array(specs[0][2], np.float32)))


def test_batch_rejects_bad_members():
    a = capi.make_config((64, 64, 64), 0.01, [0, 0, 1])
    b = capi.make_config((62, 64, 64), 0.01, [0, 0, 1])        # dim_x % 4 != 0
    with pytest.raises(capi.TsdfError, match="multiple of 4"):
        capi.Batch([a, b])
    c = capi.make_config((64, 64, 64), 0.01, [0, 0, 1], im_height=240, im_width=320)
    with pytest.raises(capi.TsdfError, match="image size"):
        capi.Batch([a, c])


def test_object_origin_on_device_equals_the_host_loop(cuda, oracle):
    """ref: src/Object.cpp:37-49 -- per-axis minimum of the back-projected masked depth, bit for bit."""
    rng = np.random.default_rng(5)
    scene = synth.SurfScene((200, 200, 200), 0.004, np.array([-0.4, -0.4, 0.7], np.float32))
    depth = scene.depth(scene.pose(3, 16), quantize=True)
    depth[rng.integers(0, 480, 500), rng.integers(0, 640, 500)] = -1.0
    mask = np.zeros((480, 640), np.uint8)
    mask[120:400, 200:520] = 255
    d_dev, m_dev = cuda.from_numpy(depth).cuda(), cuda.from_numpy(mask).cuda()
    got = capi.object_origin(d_dev.data_ptr(), m_dev.data_ptr(), 480, 640, synth.TUM_K)
    want = oracle.object_origin(oracle.mask_depth(depth, mask), synth.TUM_K)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and want[0] < 0 < want[2] < 1000
    got = capi.object_origin(d_dev.data_ptr(), None, 480, 640, synth.TUM_K)
    assert np.array_equal(got.view(np.uint32), oracle.object_origin(depth, synth.TUM_K).view(np.uint32))
    zero = cuda.zeros((480, 640), dtype=cuda.float32, device="cuda")
    assert np.array_equal(capi.object_origin(zero.data_ptr(), None, 480, 640, synth.TUM_K), [1000, 1000, 1000])


def test_object_origin_negative_zero_and_stream_order(cuda, oracle):
    """A denormal depth left of the principal point back-projects to x = -0.0f, which the reference's running
    std::min keeps against every positive x (src/Object.cpp:45: `origin < x ? origin : x`); on the device the
    atomic must take its branch by the sign BIT (pattern 0x80000000 is INT_MIN as a signed integer).  Also: the
    frame is produced on a handle's own (non-blocking) stream right before the call -- no tsdf_sync in between."""
    depth = np.zeros((480, 640), np.float32)
    depth[100:300, 400:600] = 1.5                      # right of the principal point: every other x is > 0
    depth[200, 320] = 1e-45                            # denormal: (320 - 320.1) * 1.4e-45 underflows to -0.0
    want = oracle.object_origin(depth, synth.TUM_K)
    assert want.view(np.uint32)[0] == 0x80000000, "the case must produce x = -0.0 in the reference's loop"
    raw = np.zeros((480, 640), np.uint16)
    raw[100:300, 400:600] = 7500
    with capi.Volume(capi.make_config((8, 8, 8), 0.01, [0, 0, 0])) as vol:
        d_dev = cuda.from_numpy(depth).cuda()
        got = capi.object_origin(d_dev.data_ptr(), None, 480, 640, synth.TUM_K)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        # conversion queued on the handle's stream, origin read immediately afterwards
        r_dev = cuda.from_numpy(raw).cuda()
        out = cuda.zeros((480, 640), dtype=cuda.float32, device="cuda")
        cuda.cuda.synchronize()
        vol.convert_depth_u16(r_dev.data_ptr(), out.data_ptr(), 5000.0, 1, 1)
        got2 = capi.object_origin(out.data_ptr(), None, 480, 640, synth.TUM_K)
        want2 = oracle.object_origin((raw.astype(np.float32) * (np.float32(1.0) / np.float32(5000.0))).astype(np.float32), synth.TUM_K)
        assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32)) and want2[2] == np.float32(1.5)


@pytest.mark.parametrize("deferred", [False, True])
@pytest.mark.parametrize("classified", [True, False])
def test_instance_masked_object_volumes_skip_exactly(cuda, oracle, classified, deferred):
    """The reference's call shape at its own grid size: 200^3 @ 4 mm object volumes, each fed depth x its instance mask,
    one frame per call -- batched launch and per-volume tsdf_integrate_masked_device.  With the per-workgroup
    classification (default) the workgroups outside an instance leave at once; results must not depend on it (variant 7
    switches it off) and must equal the oracle bit for bit.  One frame carries +inf outside the masks: inf x 0 = NaN,
    which DOES update a voxel (ref: src/tsdf.cu:46,49), so those tiles may claim nothing.
    deferred (the default for members of >= 4 M voxels): the batch collects the frames -- the depth once, the masks by one
    gather launch -- and applies them with one fused launch per member when it is observed; not deferred: one batched
    launch per frame."""
    rng = np.random.default_rng(21)
    dims, vs = (200, 200, 200), 0.004
    K = synth.TUM_K
    objs = []
    for i in range(4):
        o = np.array([-0.4 + rng.uniform(-0.25, 0.25), -0.4 + rng.uniform(-0.2, 0.2), 0.7 + rng.uniform(0, 0.6)], np.float32)
        c = o + 0.4
        m = np.zeros((480, 640), np.uint8)
        u0, u1 = K[0] * (c[0] - 0.2) / c[2] + K[2], K[0] * (c[0] + 0.2) / c[2] + K[2]
        v0, v1 = K[4] * (c[1] - 0.2) / c[2] + K[5], K[4] * (c[1] + 0.2) / c[2] + K[5]
        m[max(0, int(v0)):max(0, min(480, int(v1))), max(0, int(u0)):max(0, min(640, int(u1)))] = 255
        objs.append((o, m))
    cfgs = [capi.make_config(dims, vs, o, vol_id=i) for i, (o, _) in enumerate(objs)]
    scene = synth.SurfScene(dims, vs, np.array([-0.4, -0.4, 0.7], np.float32))
    frames = []
    for k in range(3):
        pose = scene.pose(k, 6)
        depth = scene.depth(pose, quantize=True)
        if k == 1:
            depth[5:9, 600:620] = np.inf
        frames.append((pose, depth))
    refs = [oracle.init_grid(dims) for _ in objs]
    with np.errstate(invalid="ignore"):
        for pose, depth in frames:
            for (o, m), (rt, rw), cfg in zip(objs, refs, cfgs):
                oracle.integrate(cfg.cam_K, pose, oracle.mask_depth(depth, m), dims, o, vs, cfg.trunc_margin, rt, rw, threads=8)
    m_dev = [cuda.from_numpy(m).cuda() for _, m in objs]
    d_dev = [cuda.from_numpy(d).cuda() for _, d in frames]
    with capi.Batch(cfgs) as batch:
        for vol in batch.volumes:
            vol.set_kernel_variant(8 if classified else 7)   # 8: classify whatever the launch size (batched launch: the first member's)
        if not deferred:
            batch.volumes[0].set_deferral(0)
        for (pose, _), d in zip(frames, d_dev):
            batch.integrate_device(d.data_ptr(), [m.data_ptr() for m in m_dev], pose)
        batch.sync()
        for vol, (rt, rw) in zip(batch.volumes, refs):
            t, w = vol.download()
            assert np.array_equal(w, rw) and np.array_equal(t.view(np.uint32), rt.view(np.uint32)), f"batch volume {vol.cfg.id}"
    for cfg, (o, m), md, (rt, rw) in zip(cfgs, objs, m_dev, refs):
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(8 if classified else 7)
            for (pose, _), d in zip(frames, d_dev):
                vol.integrate_masked_device(d.data_ptr(), md.data_ptr(), pose)
            t, w = vol.download()
        assert np.array_equal(w, rw) and np.array_equal(t.view(np.uint32), rt.view(np.uint32)), f"volume {cfg.id}"
    assert sum(float(rw.sum()) for _, rw in refs) > 10000 and sum(1 for _, rw in refs if rw.sum() > 1000) >= 2


def test_deferred_batch_keeps_call_order_with_borrowed_handles(cuda, oracle):
    """Frames given to the batch and frames given to a member through its borrowed handle are applied in call order:
    batch frame A (collected), member frame B (the batch's collected frames are applied first), batch frame C, an
    observation of ONE member (flushes the whole batch), batch frame D, destruction with D still collected (dropped
    without a crash; nothing can tell whether it was applied)."""
    dims, vs = (200, 200, 200), 0.004       # 8 M voxels each: the deferred route
    origins = [np.array([-0.4, -0.4, 0.7], np.float32), np.array([-0.3, -0.45, 0.9], np.float32)]
    cfgs = [capi.make_config(dims, vs, o, vol_id=i) for i, o in enumerate(origins)]
    scene = synth.SurfScene(dims, vs, origins[0])
    masks = []
    for r in ((100, 400, 150, 500), (60, 300, 250, 620)):
        m = np.zeros((480, 640), np.uint8)
        m[r[0]:r[1], r[2]:r[3]] = 255
        masks.append(m)
    poses = [scene.pose(k, 6) for k in range(4)]
    depths = [scene.depth(p, quantize=True) for p in poses]
    refs = [oracle.init_grid(dims) for _ in cfgs]

    def ref_integrate(i, k, masked=True):
        d = oracle.mask_depth(depths[k], masks[i]) if masked else depths[k]
        oracle.integrate(cfgs[i].cam_K, poses[k], d, dims, origins[i], vs, cfgs[i].trunc_margin, refs[i][0], refs[i][1], threads=8)

    d_dev = [cuda.from_numpy(d).cuda() for d in depths]
    m_dev = [cuda.from_numpy(m).cuda() for m in masks]
    batch = capi.Batch(cfgs)
    try:
        mp = [m.data_ptr() for m in m_dev]
        batch.integrate_device(d_dev[0].data_ptr(), mp, poses[0])               # A
        for i in range(2):
            ref_integrate(i, 0)
        batch.volumes[1].integrate_device(d_dev[1].data_ptr(), poses[1])        # B: member 1 only, unmasked
        ref_integrate(1, 1, masked=False)
        batch.integrate_device(d_dev[2].data_ptr(), mp, poses[2])               # C
        for i in range(2):
            ref_integrate(i, 2)
        t1, w1 = batch.volumes[1].download()                                    # observes member 1: everything so far applies
        assert np.array_equal(w1, refs[1][1]) and np.array_equal(t1.view(np.uint32), refs[1][0].view(np.uint32))
        t0, w0 = batch.volumes[0].download()
        assert np.array_equal(w0, refs[0][1]) and np.array_equal(t0.view(np.uint32), refs[0][0].view(np.uint32))
        assert refs[1][1].max() >= 3 and refs[0][1].max() >= 2
        batch.integrate_device(d_dev[3].data_ptr(), mp, poses[3])               # D: still collected when the batch goes
    finally:
        batch.close()


def test_deferred_batch_of_many_members_on_side_streams(cuda, oracle):
    """Eight and more members: the members' fused launches of a flush go out on four side streams forked from and joined
    back into the batch's stream.  Ten members, instance masks, two flushes (35 frames: a full pass + 3) and an observation
    in between -- every member equal to the oracle."""
    rng = np.random.default_rng(77)
    dims, vs = (96, 80, 48), 0.006
    n_members, n_frames = 10, 35
    scene = synth.SurfScene((200, 200, 200), 0.004, np.array([-0.4, -0.4, 0.7], np.float32))
    members = []
    for i in range(n_members):
        o = np.array([-0.45 + 0.07 * i, -0.3 + rng.uniform(0, 0.2), 0.7 + rng.uniform(0, 0.5)], np.float32)
        m = np.zeros((480, 640), np.uint8)
        r0, c0 = int(rng.integers(0, 200)), int(rng.integers(0, 300))
        m[r0:r0 + 280, c0:c0 + 340] = 255
        members.append((o, m))
    cfgs = [capi.make_config(dims, vs, o, vol_id=i) for i, (o, _) in enumerate(members)]
    poses = [scene.pose(k % 16, 16) for k in range(n_frames)]
    depths = [scene.depth(scene.pose(k, 16), quantize=True) for k in range(16)]
    refs = [oracle.init_grid(dims) for _ in members]
    d_dev = [cuda.from_numpy(d).cuda() for d in depths]
    m_dev = [cuda.from_numpy(m).cuda() for _, m in members]
    with capi.Batch(cfgs) as batch:
        for k in range(n_frames):
            batch.integrate_device(d_dev[k % 16].data_ptr(), [m.data_ptr() for m in m_dev], poses[k])
            for (o, m), (rt, rw), cfg in zip(members, refs, cfgs):
                oracle.integrate(cfg.cam_K, poses[k], oracle.mask_depth(depths[k % 16], m), dims, o, vs, cfg.trunc_margin, rt, rw, threads=8)
            if k == 20:     # an observation of one member in the middle of a pass flushes all of them
                t, w = batch.volumes[3].download()
                assert np.array_equal(w, refs[3][1]) and np.array_equal(t.view(np.uint32), refs[3][0].view(np.uint32))
        batch.sync()
        for vol, (rt, rw) in zip(batch.volumes, refs):
            t, w = vol.download()
            assert np.array_equal(w, rw), f"member {vol.cfg.id}: weights differ"
            assert np.array_equal(t.view(np.uint32), rt.view(np.uint32)), f"member {vol.cfg.id}: TSDF differs"
    assert sum(float(rw.sum()) for _, rw in refs) > 100000
