"""Deferred integration of host frames: tsdf_integrate (the reference's TSDF::Integrate shape, ref: src/tsdf.cu:135-168)
collects frames and applies them 32 at a time as one fused sequence; results are only observable through calls that
flush first.  Whatever the batch size and however the calls interleave, the volume must equal the oracle's frame-by-frame
replay bit for bit."""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def scene_frames(dims, vs, origin, n):
    sc = synth.SurfScene(dims, vs, origin)
    return [(sc.pose(k % 9, 11), sc.depth(sc.pose(k % 9, 11), quantize=True)) for k in range(n)]


@pytest.mark.parametrize("dims,vs", [((256, 48, 20), 0.008), ((200, 40, 24), 0.01), ((37, 20, 12), 0.05)])
@pytest.mark.parametrize("defer", [32, 5, 0])
def test_deferred_host_frames_equal_frame_by_frame(cuda, oracle, dims, vs, defer):
    origin = synth.surf_volume(dims[0], vs, 0.8)
    cfg = capi.make_config(dims, vs, origin)
    frames = scene_frames(dims, vs, origin, 70)          # two full batches of 32 and a partial one
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Volume(cfg) as vol:
        vol.set_deferral(defer)
        for k, (pose, depth) in enumerate(frames):
            buf = depth.copy()
            vol.integrate(buf, pose)
            buf[:] = -7.0                                  # the caller's buffer is free again when the call returns
            assert np.array_equal(vol.last_cam2base(), oracle.cam2base(synth.identity_pose(), pose))
            oracle.integrate(cfg.cam_K, pose, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
            if k in (3, 40):                               # observing in the middle of a batch applies what was collected
                t, w = vol.download()
                assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)), k
        t, w = vol.download()
    assert ref_w.max() >= 5
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))


def test_deferred_frames_keep_their_place_among_other_calls(cuda, oracle):
    """Host frames, device-resident frames, a masked frame, a fused sequence, a reset and an upload interleaved: every
    call sees the volume as if all earlier calls had already run."""
    dims, vs = (256, 40, 16), 0.008
    origin = synth.surf_volume(256, vs, 0.8)
    cfg = capi.make_config(dims, vs, origin)
    fr = scene_frames(dims, vs, origin, 12)
    mask = np.zeros((480, 640), np.uint8)
    mask[100:400, 150:500] = 255
    ref_t, ref_w = oracle.init_grid(dims)
    def ref(pose, depth, m=None):
        oracle.integrate(cfg.cam_K, pose, depth if m is None else oracle.mask_depth(depth, m), dims, origin, vs,
                         cfg.trunc_margin, ref_t, ref_w)
    with capi.Volume(cfg) as vol:
        dev = [cuda.from_numpy(d).cuda() for _, d in fr]
        m_dev = cuda.from_numpy(mask).cuda()
        vol.integrate(fr[0][1], fr[0][0]); ref(*fr[0])
        vol.integrate(fr[1][1], fr[1][0]); ref(*fr[1])
        vol.integrate_device(dev[2].data_ptr(), fr[2][0]); ref(*fr[2])                      # after the two collected ones
        vol.integrate(fr[3][1], fr[3][0]); ref(*fr[3])
        vol.integrate_masked_device(dev[4].data_ptr(), m_dev.data_ptr(), fr[4][0]); ref(fr[4][0], fr[4][1], mask)
        vol.integrate(fr[5][1], fr[5][0]); ref(*fr[5])
        vol.integrate_frames_device([d.data_ptr() for d in dev[6:9]], np.stack([p for p, _ in fr[6:9]]))
        for p, d in fr[6:9]:
            ref(p, d)
        vol.integrate(fr[9][1], fr[9][0]); ref(*fr[9])
        assert vol.count_surface() == len(oracle.surface_points(ref_t, ref_w, dims, vs, origin))
        t, w = vol.download()
        assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
        # a reset discards collected frames' effect like everything else; an upload replaces the state they were applied to
        vol.integrate(fr[10][1], fr[10][0])
        vol.reset()
        t, w = vol.download()
        assert np.all(t == 1.0) and np.all(w == 0.0)
        vol.integrate(fr[10][1], fr[10][0])
        vol.upload(ref_t, ref_w)
        vol.integrate(fr[11][1], fr[11][0]); ref(*fr[11])
        t, w = vol.download()
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))


@pytest.mark.parametrize("defer", [32, 0])
def test_group_deferral(cuda, oracle, defer):
    dims, vs = (256, 32, 21), 0.008
    origin = synth.surf_volume(256, vs, 0.8)
    cfg = capi.make_config(dims, vs, origin)
    frames = scene_frames(dims, vs, origin, 37)
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Group(cfg, [0, 0, 0]) as grp:
        grp.set_deferral(defer)
        for k, (pose, depth) in enumerate(frames):
            grp.integrate(depth, pose)
            oracle.integrate(cfg.cam_K, pose, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
            if k == 10:
                assert len(grp.extract_surface()) == len(oracle.surface_points(ref_t, ref_w, dims, vs, origin))
        t, w = grp.download()
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))


def test_many_object_handles_share_one_frame_store(cuda, oracle):
    """The reference keeps one TSDF per object instance (ref: src/Object.cpp:67, src/Engine.cpp:172-233 creates them by the
    dozen).  Staging, deferral and tile-table memory is one store per device and image size (csrc/frame_store.h), not
    per handle: 64 handles of 64^3 voxels (2 MB each), each fed host frames through the reference's own call, must cost far
    less than the 64 x ~105 MB of per-handle pools they used to -- asserted on hipMemGetInfo -- and every one must still equal
    the oracle.  More frames are collected (64 x 5 = 320) than the store's soft cap holds (160 slots), so handles also
    flush themselves to make room."""
    dims, vs = (64, 64, 64), 0.0125
    n_obj, n_frames = 64, 5
    scene = synth.SurfScene((200, 200, 200), 0.004, np.array([-0.4, -0.4, 0.7], np.float32))
    poses = [scene.pose(k, 7) for k in range(n_frames)]
    depths = [scene.depth(p, quantize=True) for p in poses]
    rng = np.random.default_rng(64)
    origins = [np.array([-0.5 + rng.uniform(0, 0.3), -0.5 + rng.uniform(0, 0.3), 0.7 + rng.uniform(0, 0.4)], np.float32) for _ in range(n_obj)]
    cfgs = [capi.make_config(dims, vs, o, vol_id=i) for i, o in enumerate(origins)]
    cuda.cuda.synchronize()
    free0, _ = cuda.cuda.mem_get_info()
    vols = [capi.Volume(c) for c in cfgs]
    try:
        for k in range(n_frames):                 # the caller's loop: every object gets every keyframe (its own masked copy)
            for i, vol in enumerate(vols):
                d = depths[k].copy()
                d[:, : 10 * (i % 8)] = 0.0        # a different frame per object, as depth x its instance mask would be
                vol.integrate(d, poses[k])
        for vol in vols:
            vol.sync()
        free1, _ = cuda.cuda.mem_get_info()
        used = free0 - free1
        volumes = n_obj * 2 * 4 * dims[0] * dims[1] * dims[2]
        # volumes + summaries + streams + one shared store at its caps (160 frame slots of 1.2 MB, 16 table slots of 9.2 MB); the per-handle pools of round 2
        # would be 64 x (7.4 + 78.6 + 9.4) MB = 6.1 GB
        assert used < volumes + 640 * 2 ** 20, f"{used / 2 ** 20:.0f} MiB in use for {volumes / 2 ** 20:.0f} MiB of volumes"
        for i, vol in enumerate(vols):
            rt, rw = oracle.init_grid(dims)
            for k in range(n_frames):
                d = depths[k].copy()
                d[:, : 10 * (i % 8)] = 0.0
                oracle.integrate(cfgs[i].cam_K, poses[k], d, dims, origins[i], vs, cfgs[i].trunc_margin, rt, rw)
            t, w = vol.download()
            assert np.array_equal(w, rw) and np.array_equal(t.view(np.uint32), rt.view(np.uint32)), f"object {i}"
    finally:
        for vol in vols:
            vol.close()
