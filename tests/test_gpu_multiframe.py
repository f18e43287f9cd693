"""Several frames per pass over the volume (tsdf_integrate_frames_device -> integrate_multi) must equal
the same frames integrated one launch at a time -- which the oracle pins -- bit for bit."""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims", [(256, 42, 30), (200, 42, 30), (36, 20, 12)])   # row mapping, flat, flat + partial chunk
@pytest.mark.parametrize("variant", capi.variants(0, 3, 7, 8, 4, 5, 6, 9, 10, 11, 12, 13))   # 9: memory-order dispatch; 10: slices fastest without the rotation; 12: bricks without the super-brick pre-pass; 11: rows classified per workgroup (default: bricks per wavefront)
@pytest.mark.parametrize("n_frames", [1, 3, 4, 9, 32, 33, 70])   # kMaxFramesPerLaunch = 32: one full pass, +1, 2 full + 6
def test_fused_frames_equal_sequential_and_oracle(cuda, oracle, n_frames, variant, dims):
    if n_frames > 9 and (variant != 0 or dims[0] == 36):
        pytest.skip("long sequences: default kernel, row and flat mapping")
    vs = 2.0 / dims[0]                       # ragged row groups for R = 1 and R = 2
    origin = synth.surf_volume(dims[0], vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    rng = np.random.default_rng(n_frames)
    frames, keep = [], []
    for k in range(n_frames):
        c2w = scene.pose(k % 5, n=7)          # repeats: weights > 1 on non-trivial TSDF values
        if k % 3 == 2:
            depth = np.full((480, 640), 5.9, np.float32)   # a pure free-space frame in the middle
        else:
            depth = scene.depth(c2w, quantize=True)
        mask = None
        if k % 4 == 1:
            mask = np.zeros((480, 640), np.uint8)
            mask[rng.integers(50, 150):rng.integers(300, 450), rng.integers(50, 200):rng.integers(400, 600)] = 255
        frames.append((c2w, depth, mask))
    ref_t, ref_w = oracle.init_grid(dims)
    for c2w, depth, mask in frames:
        d = depth if mask is None else oracle.mask_depth(depth, mask)
        oracle.integrate(cfg.cam_K, c2w, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(variant)
        for c2w, depth, mask in frames:
            keep.append((cuda.from_numpy(depth).cuda(), None if mask is None else cuda.from_numpy(mask).cuda()))
        vol.integrate_frames_device([d.data_ptr() for d, _ in keep], np.stack([f[0] for f in frames]),
                                    [None if m is None else m.data_ptr() for _, m in keep])
        t, w = vol.download()
        assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
        # and a second batch on top of the first (state carried through memory + summary)
        vol.integrate_frames_device([d.data_ptr() for d, _ in keep][:2], np.stack([f[0] for f in frames[:2]]),
                                    [None if m is None else m.data_ptr() for _, m in keep][:2])
        for c2w, depth, mask in frames[:2]:
            d = depth if mask is None else oracle.mask_depth(depth, mask)
            oracle.integrate(cfg.cam_K, c2w, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        t, w = vol.download()
    assert ref_w.max() >= 2 or n_frames == 1
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))


def test_patch_classification_experiment_is_exact_and_fires(cuda, oracle):
    """Variant 8: wavefronts whose patch the depth tile summaries decide (all free space / nothing to update) never
    project a voxel.  Planes at depths around cz + trunc (where the free-space claim must stop), a frame the volume
    misses entirely and a frame with invalid pixels: bit-exact against the oracle, and every class is taken."""
    dims, vs = (256, 64, 40), 0.002      # a 256-voxel row fits the image at this range
    origin = synth.surf_volume(256, vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    far = origin[2] + dims[2] * vs
    depths = [np.full((480, 640), far + 0.5, np.float32),                      # everything free space
              np.full((480, 640), far + cfg.trunc_margin * 1.0001, np.float32),  # the far slices just outside the claim
              np.full((480, 640), far - 0.03, np.float32),                     # band inside the volume
              np.full((480, 640), origin[2] - 0.2, np.float32),                # surface in front of the volume: nothing updated
              np.full((480, 640), far + 0.5, np.float32)]
    depths[4][100:300, 200:400] = 0.0                                          # invalid pixels in the middle
    poses = [synth.identity_pose(), synth.make_pose(synth.rot_z(0.05), [0.01, 0.0, 0.0]), synth.identity_pose(),
             synth.identity_pose(), synth.make_pose(synth.rot_y(0.03), [0.0, 0.01, 0.0]),
             synth.make_pose(np.eye(3), [5.0, 0.0, 0.0])]                        # the volume is out of view
    depths.append(depths[0])
    ref_t, ref_w = oracle.init_grid(dims)
    for d, p in zip(depths, poses):
        oracle.integrate(cfg.cam_K, p, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
    keep = [cuda.from_numpy(d).cuda() for d in depths]
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(8)
        vol.shortcut_stats(True)
        vol.integrate_frames_device([d.data_ptr() for d in keep], np.stack(poses))
        per_voxel, free, skipped = vol.shortcut_stats(False)
        t, w = vol.download()
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
    assert per_voxel > 0 and free > 0 and skipped > 0 and per_voxel + free + skipped == 6 * dims[1] * dims[2]


def test_patch_classification_with_instance_masks(cuda, oracle):
    """Variant 8 on masked frames: the tile summary is taken of depth x mask, so workgroups that project outside the
    instance mask are skipped -- the per-object volumes of the reference's real usage.  Bit-exact, and skips happen."""
    dims, vs = (200, 120, 60), 0.004
    origin = synth.surf_volume(200, vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    rng = np.random.default_rng(8)
    frames = []
    for k in range(10):
        c2w = scene.pose(k % 6, n=9)
        depth = scene.depth(c2w, quantize=True)
        if k == 4:
            depth[50:60, 300:340] = np.inf          # inf x 0 = NaN outside the mask: that tile must claim nothing
        mask = np.zeros((480, 640), np.uint8)
        y0, x0 = int(rng.integers(100, 200)), int(rng.integers(150, 300))
        mask[y0:y0 + 120, x0:x0 + 160] = 255
        frames.append((c2w, depth, mask if k % 5 else None))
    ref_t, ref_w = oracle.init_grid(dims)
    with np.errstate(invalid="ignore"):
        for c2w, depth, mask in frames:
            d = depth if mask is None else oracle.mask_depth(depth, mask)
            oracle.integrate(cfg.cam_K, c2w, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
    keep = [(cuda.from_numpy(d).cuda(), None if m is None else cuda.from_numpy(m).cuda()) for _, d, m in frames]
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(8)
        vol.shortcut_stats(True)
        vol.integrate_frames_device([d.data_ptr() for d, _ in keep], np.stack([f[0] for f in frames]),
                                    [None if m is None else m.data_ptr() for _, m in keep])
        per_voxel, free, skipped = vol.shortcut_stats(False)
        t, w = vol.download()
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
    assert skipped > 5000 and per_voxel > 0


def test_classification_is_dropped_when_it_claims_nothing_and_probed_again(cuda, oracle):
    """Default variant: the first launch classifies and counts; with one invalid pixel in every depth tile and the
    volume in view nothing can be claimed, so the following launches go without, except a probe every eighth.  Then a
    clean frame stream: the probe finds everything claimable and classification stays on.  Bit-exact throughout."""
    dims, vs = (256, 32, 16), 0.002
    origin = synth.surf_volume(256, vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    far = origin[2] + dims[2] * vs
    holes = np.full((480, 640), far + 0.5, np.float32)
    holes[::16, ::16] = 0.0
    clean = np.full((480, 640), far + 0.5, np.float32)
    pose = synth.identity_pose()
    ref_t, ref_w = oracle.init_grid(dims)
    d_holes, d_clean = cuda.from_numpy(holes).cuda(), cuda.from_numpy(clean).cuda()
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(0)      # the default policy is what is tested
        fpl = vol.frames_per_launch
        seen = []
        for launch in range(12):
            vol.integrate_frames_device([d_holes.data_ptr()] * fpl, np.stack([pose] * fpl))
            for _ in range(fpl):
                oracle.integrate(cfg.cam_K, pose, holes, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
            seen.append(vol.classification_info())
        assert seen[0][0] == 0.0 and seen[0][1] == 0            # the first launch classified, claimed nothing
        assert max(s[1] for s in seen) == 7 and seen[-1][1] < 7  # ... then 7 launches without, a probe, and again
        for launch in range(3):
            vol.integrate_frames_device([d_clean.data_ptr()] * fpl, np.stack([pose] * fpl))
            for _ in range(fpl):
                oracle.integrate(cfg.cam_K, pose, clean, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        for launch in range(9):   # reach the next probe
            vol.integrate_frames_device([d_clean.data_ptr()] * fpl, np.stack([pose] * fpl))
            for _ in range(fpl):
                oracle.integrate(cfg.cam_K, pose, clean, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        frac, idle = vol.classification_info()
        assert frac > 0.9 and idle == 0
        t, w = vol.download()
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))


@pytest.mark.parametrize("variant", capi.variants(8, 11, 13))
def test_runs_of_free_space_frames_with_awkward_weights(cuda, oracle, variant):
    """A run of claimed free-space frames moves the weights by its length at once -- only when that is the same bits
    as that many "+ 1": weights that are not integers (an upload), weights at and around 2^24 (where w + 1 == w) and
    ordinary counts, all with TSDF = 1 so that the free-space path is taken; the run is interrupted by a frame that
    sees nothing and one with a surface inside the volume.  Bit-exact against the oracle, classification forced on."""
    dims, vs = (256, 16, 12), 0.002
    origin = synth.surf_volume(256, vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    far = origin[2] + dims[2] * vs
    n = dims[0] * dims[1] * dims[2]
    rng = np.random.default_rng(4)
    w0 = rng.choice(np.array([0.0, 3.0, 0.5, 2.75, 16777215.0, 16777216.0, 16777184.0, 16777183.0, 16777150.0, 8388607.5,
                              3.9999998, 1e7], np.float32), n).astype(np.float32)
    w0[: 256 * 16] = 5.0                      # one slice of plain counts (the aggregated path must fire somewhere)
    t0 = np.ones(n, np.float32)
    free = np.full((480, 640), far + 0.5, np.float32)
    none = np.full((480, 640), origin[2] - 0.2, np.float32)
    band = np.full((480, 640), far - 0.01, np.float32)
    depths = [free] * 9 + [none] + [free] * 7 + [band] + [free] * 16 + [free] * 5      # 39 frames: two passes
    pose = synth.identity_pose()
    ref_t, ref_w = t0.copy(), w0.copy()
    for d in depths:
        oracle.integrate(cfg.cam_K, pose, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
    keep = {id(d): cuda.from_numpy(d).cuda() for d in (free, none, band)}
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(variant)
        vol.upload(t0, w0)
        vol.shortcut_stats(True)
        vol.integrate_frames_device([keep[id(d)].data_ptr() for d in depths], np.stack([pose] * len(depths)))
        per_voxel, fr, sk = vol.shortcut_stats(False)
        t, w = vol.download()
    assert fr > 0 and sk > 0
    assert np.array_equal(w, ref_w), f"{np.count_nonzero(w != ref_w)} weights differ, e.g. {w[w != ref_w][:4]} vs {ref_w[w != ref_w][:4]}"
    assert np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
    assert ref_w[0] == 5.0 + 38.0 or ref_w[0] == 5.0 + 37.0     # counts moved by the frames that saw the voxel


def test_wide_slices_fall_back_to_memory_order(cuda, oracle):
    """A slice of 4096 x 2048 voxels holds 65 536 groups of four bricks -- one more than a slow grid dimension holds.  The
    classified launch runs over the brick work list (a one-dimensional grid, so the limit no longer shapes it; the
    measurement build's brick workgroups fall back to memory order here).  Same bits; claims are made."""
    dims, vs = (4096, 2048, 8), 0.00025
    origin = np.array([-dims[0] * vs / 2, -dims[1] * vs / 2, 1.2], np.float32)
    cfg = capi.make_config(dims, vs, origin)
    far = float(origin[2]) + dims[2] * vs
    poses = [synth.identity_pose(), synth.make_pose(synth.rot_z(0.02), [0.003, -0.002, 0.0]), synth.make_pose(synth.rot_y(0.01), [0.0, 0.004, 0.0])]
    depths = [np.full((480, 640), far + 0.2, np.float32) for _ in poses]
    depths[1][:, 320:] = float(origin[2]) + 0.5 * dims[2] * vs          # a surface through the right half of the slab
    depths[2][:240, :] = float(origin[2]) - 0.3                         # in front of the slab: nothing updated up there
    ref_t, ref_w = oracle.init_grid(dims)
    for p, d in zip(poses, depths):
        oracle.integrate(cfg.cam_K, p, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, threads=8)
    keep = [cuda.from_numpy(d).cuda() for d in depths]
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(8)
        assert vol.brick_shape() == (2, 4, 8)
        vol.shortcut_stats(True)
        vol.integrate_frames_device([d.data_ptr() for d in keep], np.stack(poses))
        per_voxel, free, skipped = vol.shortcut_stats(False)
        t, w = vol.download()
    assert free > 0 and skipped > 0 and per_voxel > 0 and per_voxel + free + skipped == 3 * (4096 // 8) * (2048 // 4)
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
    assert ref_w.max() == 3 and ref_w.min() < 3


@pytest.mark.parametrize("caller_stream", [False, True])
@pytest.mark.parametrize("variant", [8, 0])
def test_pipelined_sequence_calls_keep_call_order(cuda, oracle, variant, caller_stream):
    """A sequence call on a slab below 64 M voxels runs the pre-pass of launch k + 1 (depth tile tables, the brick work list) on a
    side stream beside launch k's Integrate kernel, with two work lists, two counter blocks and two table slots in turn
    (csrc/tsdf_capi.hip, launch_multi).  Many launches per call (7, then 3, then 5: the parity of the buffers changes between
    calls), a different depth frame for every pose, frames that update nothing in between, and calls of other kinds wedged between
    the sequence calls -- one frame through its own launch, a download, a host-pointer frame that is collected and flushed by the
    next call: the result must be the oracle's for the same frames in call order, bit for bit.  caller_stream: the handle runs on a
    stream the caller owns (tsdf_set_stream), the frames are uploaded on that stream right before each call (no synchronisation
    in between: the side stream must wait for what precedes the call on the caller's stream)."""
    dims, vs = (200, 96, 64), 0.01
    origin = synth.surf_volume(200, vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)

    def frame(k):
        c2w = scene.pose(k % 37, n=37)
        depth = scene.depth(c2w, quantize=True) if k % 11 else np.zeros((480, 640), np.float32)     # every eleventh: nothing to see
        return c2w, depth

    def oracle_apply(fr):
        for c2w, depth in fr:
            oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)

    keep = []

    stream = cuda.cuda.Stream() if caller_stream else None

    def sequence(vol, fr):
        if stream is None:
            dev = [cuda.from_numpy(d).cuda() for _, d in fr]
        else:
            with cuda.cuda.stream(stream):      # uploads queued on the caller's stream, not waited for
                pinned = [cuda.from_numpy(d).pin_memory() for _, d in fr]
                dev = [t.to("cuda", non_blocking=True) for t in pinned]
            keep.append(pinned)
        keep.append(dev)
        vol.integrate_frames_device([d.data_ptr() for d in dev], np.stack([p for p, _ in fr]))

    with capi.Volume(cfg) as vol:
        if stream is not None:
            vol.set_stream(stream.cuda_stream)
        vol.set_kernel_variant(variant)
        a = [frame(k) for k in range(0, 7 * 32 - 5)]
        sequence(vol, a)
        oracle_apply(a)
        one = frame(500)
        d_one = cuda.from_numpy(one[1]).cuda()
        cuda.cuda.synchronize()
        vol.set_deferral(0)
        vol.integrate_device(d_one.data_ptr(), one[0])                  # its own launch on the handle's stream
        oracle_apply([one])
        b = [frame(k) for k in range(300, 300 + 3 * 32)]
        sequence(vol, b)
        oracle_apply(b)
        t, w = vol.download()
        assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)), "after the second sequence"
        vol.set_deferral(32)
        host = frame(700)
        vol.integrate(host[1], host[0])                                 # collected; the next call applies it first
        oracle_apply([host])
        c = [frame(k) for k in range(800, 800 + 5 * 32 + 1)]
        sequence(vol, c)
        oracle_apply(c)
        frac, idle = vol.classification_info()
        t, w = vol.download()
    assert ref_w.max() > 100 and (ref_t != 1.0).sum() > 10000
    assert np.array_equal(w, ref_w), f"weights differ at {np.flatnonzero(w != ref_w)[:5]}"
    assert np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
    if variant == 8:
        assert idle == 0 and frac > 0.3, (frac, idle)
