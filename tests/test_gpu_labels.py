"""Per-voxel semantic-label fusion (BASELINE config 5) on the device against the CPU
restatement of the same project-defined rule (no reference function exists: parity unpinned by
reference output; the evidence rule itself mirrors ref: src/ObjectPoint.cpp:190-219)."""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def instance_masks(rng, k, h=480, w=640):
    """k overlapping rectangles in MaskRCNN's output format: uint8 {0,255}, label 1..80, score in (0.8, 1]."""
    masks = np.zeros((k, h, w), np.uint8)
    for m in range(k):
        y0, x0 = rng.integers(0, h - 120), rng.integers(0, w - 160)
        masks[m, y0:y0 + rng.integers(80, 300), x0:x0 + rng.integers(100, 400)] = 255
    labels = rng.integers(1, 81, k).astype(np.uint16)
    scores = rng.uniform(0.8, 1.0, k).astype(np.float32)
    return masks, labels, scores


def test_compose_and_fuse_labels_match_oracle(cuda, oracle):
    dims, vs = (128, 96, 64), 0.01
    origin = synth.surf_volume(128, vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    rng = np.random.default_rng(3)
    n = dims[0] * dims[1] * dims[2]
    ref_l, ref_f, ref_b = np.zeros(n, np.uint16), np.zeros(n, np.float32), np.zeros(n, np.float32)
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Volume(cfg) as vol:
        vol.labels_enable(0.5)
        lab_dev = cuda.empty((480, 640), dtype=cuda.uint16, device="cuda")
        sc_dev = cuda.empty((480, 640), dtype=cuda.float32, device="cuda")
        changed = 0
        for k in range(6):
            c2w = scene.pose(k % 4, n=8)
            depth = scene.depth(c2w, quantize=True)
            masks, labels, scores = instance_masks(rng, 5)
            if k >= 3:
                labels[:] = labels[::-1]          # same regions, other classes: background evidence and re-adoption
            m_dev, d_dev = cuda.from_numpy(masks).cuda(), cuda.from_numpy(depth).cuda()
            vol.compose_labels(m_dev.data_ptr(), labels, scores, lab_dev.data_ptr(), sc_dev.data_ptr())
            want_lab, want_sc = oracle.compose_labels(masks, labels, scores)
            vol.sync()
            assert np.array_equal(lab_dev.cpu().numpy(), want_lab) and np.array_equal(sc_dev.cpu().numpy(), want_sc)
            vol.integrate_labels_device(d_dev.data_ptr(), lab_dev.data_ptr(), sc_dev.data_ptr(), c2w)
            vol.integrate_device(d_dev.data_ptr(), c2w)         # the TSDF pass of the same frame is independent
            vol.sync()
            changed += oracle.integrate_labels(cfg.cam_K, c2w, depth, want_lab, want_sc, dims, origin, vs,
                                               cfg.trunc_margin, ref_l, ref_f, ref_b)
            oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        lab, fp, bp = vol.download_labels()
        t, w = vol.download()
    assert changed > 5000 and np.count_nonzero(ref_b) > 100 and len(np.unique(ref_l)) > 3
    assert np.array_equal(lab, ref_l)
    assert np.array_equal(fp.view(np.uint32), ref_f.view(np.uint32))
    assert np.array_equal(bp.view(np.uint32), ref_b.view(np.uint32))
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
    # labels only ever attach to voxels the TSDF saw inside the truncation band
    assert np.all(ref_w[ref_l != 0] > 0)


def _config4_frames(oracle, scene, n_frames, seed=9):
    """n_frames labelled frames of the S-surf orbit: (cam2world, depth, label image, score image); the instance masks of
    every third frame carry their classes in reverse order, so background evidence accumulates and labels get re-adopted."""
    rng = np.random.default_rng(seed)
    frames = []
    for k in range(n_frames):
        c2w = scene.pose(k, n=48)
        depth = scene.depth(c2w, quantize=True)
        masks, labels, scores = instance_masks(rng, 6)
        if k % 3 == 2:
            labels[:] = labels[::-1]
        lab, sc = oracle.compose_labels(masks, labels, scores)
        frames.append((c2w, depth, lab, sc))
    return frames


@pytest.mark.parametrize("rank_", [1, 2])
def test_config4_whole_slab_every_voxel(cuda, oracle, rank_):
    """BASELINE configs[4] on one GPU, EVERY voxel: one rank's slab of a 2048^3 @ 2 mm grid cut eight ways -- 2048 x 2048 x 256
    voxels, 8.6 GB TSDF + weight and 10.7 GB label state, dim_y = 2048 (twice the reference kernel's block limit,
    so the CPU oracle is the only checker), byte offsets beyond 2^32 -- rank 1 (z_begin = 256: the slab that holds the visible cap
    of the sphere, i.e. the truncation band and nearly all label evidence) and rank 2 (z_begin = 512: free space in front of the
    wall, the sphere's shadow and the band where the sphere leaves the view) -- takes 35 labelled frames (one more than a
    32-frame pass) through the path a config-5 run takes, tsdf_integrate_frames_labels_device, with the classification
    decided per launch (variant 0) and forced (variant 8: classify_brick_list + integrate_brick_list<NT, LABELS>), and
    through the per-frame pair tsdf_integrate_device + tsdf_integrate_labels_device.  TSDF bits, weights, labels, Fp
    and Bp of all 1 073 741 824 voxels are compared with the oracle (all host threads: rows are independent).
    The evidence rule being matched: ref src/ObjectPoint.cpp:190-219, :149-154; index width: src/tsdf.cu:52."""
    import time
    D, vs = 2048, 0.002
    dims = (D, D, D)
    zb, ze = rank_ * D // 8, (rank_ + 1) * D // 8
    n_frames = 35
    origin = np.array([-2.048, -2.048, 0.4], np.float32)
    cfg = capi.make_config(dims, vs, origin, z_begin=zb, z_end=ze)
    scene = synth.SurfScene(dims, vs, origin)
    frames = _config4_frames(oracle, scene, n_frames)
    poses = np.stack([f[0] for f in frames])
    n = D * D * (ze - zb)
    t0 = time.time()
    ref_t, ref_w = oracle.init_grid(dims, zb, ze)
    ref_l, ref_f, ref_b = np.zeros(n, np.uint16), np.zeros(n, np.float32), np.zeros(n, np.float32)
    upd = seen = 0
    for c2w, depth, lab, sc in frames:
        upd += oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, z_begin=zb, z_end=ze)
        seen += oracle.integrate_labels(cfg.cam_K, c2w, depth, lab, sc, dims, origin, vs, cfg.trunc_margin,
                                        ref_l, ref_f, ref_b, z_begin=zb, z_end=ze)
    t_oracle = time.time() - t0
    if rank_ == 1:
        assert upd > 10_000_000 * n_frames and seen > 20_000_000, (upd, seen)
        assert np.count_nonzero(ref_b) > 100_000 and len(np.unique(ref_l)) > 5    # background evidence and several classes
    else:
        assert upd > 10_000_000 and seen > 100_000, (upd, seen)
    keep = [(cuda.from_numpy(d).cuda(), cuda.from_numpy(l).cuda(), cuda.from_numpy(s_).cuda()) for _, d, l, s_ in frames]

    def compare(vol, what):
        t, w = vol.download()
        assert np.array_equal(w, ref_w), f"{what}: weights differ at {np.flatnonzero(w != ref_w)[:4]}"
        bad = np.flatnonzero(t.view(np.uint32) != ref_t.view(np.uint32))
        assert bad.size == 0, f"{what}: {bad.size} TSDF values differ, first at {bad[:4]}"
        del t, w
        lab, fp, bp = vol.download_labels()
        assert np.array_equal(lab, ref_l), f"{what}: labels differ at {np.flatnonzero(lab != ref_l)[:4]}"
        assert np.array_equal(fp.view(np.uint32), ref_f.view(np.uint32)), f"{what}: Fp differs"
        assert np.array_equal(bp.view(np.uint32), ref_b.view(np.uint32)), f"{what}: Bp differs"

    t0 = time.time()
    for variant in (0, 8):
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(variant)
            vol.labels_enable(0.5)
            vol.integrate_frames_labels_device([d.data_ptr() for d, _, _ in keep], [l.data_ptr() for _, l, _ in keep],
                                               [s_.data_ptr() for _, _, s_ in keep], poses)
            frac, idle = vol.classification_info()
            if variant == 8:
                assert idle == 0 and frac > 0.5, (frac, idle)   # both passes went over the brick work list and its claims
            compare(vol, f"fused, variant {variant}")
    with capi.Volume(cfg) as vol:
        vol.labels_enable(0.5)
        vol.set_deferral(0)
        for (c2w, _, _, _), (d, l, s_) in zip(frames, keep):
            vol.integrate_labels_device(d.data_ptr(), l.data_ptr(), s_.data_ptr(), c2w)
            vol.integrate_device(d.data_ptr(), c2w)
        compare(vol, "one launch per call")
    print(f"configs[4] slab of rank {rank_}: {n} voxels x {n_frames} frames, {upd} updates, {seen} label events; oracle {t_oracle:.0f} s "
          f"({oracle.max_threads()} threads), three device paths + compares {time.time() - t0:.0f} s")


@pytest.mark.parametrize("shape", [None, (2, 8, 4), (2, 5, 6)])   # the wavefront brick: the library's choice, spanning slices
@pytest.mark.parametrize("variant", [0, 8, 7])   # classified through wavefront bricks: per launch / always / never
@pytest.mark.parametrize("dims,n_frames", [((256, 40, 24), 7), ((200, 40, 24), 35)])   # row mapping; flat mapping across a pass boundary
def test_fused_integrate_and_labels_equal_separate_passes(cuda, oracle, dims, n_frames, variant, shape):
    """tsdf_integrate_frames_labels_device == tsdf_integrate_device + tsdf_integrate_labels_device per frame == the
    oracle's two functions, bit for bit, for the TSDF, the weights and the three label arrays."""
    if shape is not None and variant != 8:
        pytest.skip("brick shapes: with the classification forced on")
    vs = 2.0 / dims[0]
    origin = synth.surf_volume(dims[0], vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    rng = np.random.default_rng(n_frames)
    n = dims[0] * dims[1] * dims[2]
    ref_l, ref_f, ref_b = np.zeros(n, np.uint16), np.zeros(n, np.float32), np.zeros(n, np.float32)
    ref_t, ref_w = oracle.init_grid(dims)
    frames = []
    for k in range(n_frames):
        c2w = scene.pose(k % 5, n=8)
        depth = scene.depth(c2w, quantize=True)
        masks, labels, scores = instance_masks(rng, 4)
        if k % 3 == 2:
            labels[:] = labels[::-1]
        lab, sc = oracle.compose_labels(masks, labels, scores)
        frames.append((c2w, depth, lab, sc))
        oracle.integrate_labels(cfg.cam_K, c2w, depth, lab, sc, dims, origin, vs, cfg.trunc_margin, ref_l, ref_f, ref_b)
        oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
    assert np.count_nonzero(ref_l) > 1000 and np.count_nonzero(ref_b) > 50
    keep = [(cuda.from_numpy(d).cuda(), cuda.from_numpy(l).cuda(), cuda.from_numpy(s_).cuda()) for _, d, l, s_ in frames]
    poses = np.stack([f[0] for f in frames])
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(variant)
        if shape is not None:
            vol.set_brick_shape(*shape)
        vol.labels_enable(0.5)
        vol.integrate_frames_labels_device([d.data_ptr() for d, _, _ in keep], [l.data_ptr() for _, l, _ in keep],
                                           [s_.data_ptr() for _, _, s_ in keep], poses)
        lab, fp, bp = vol.download_labels()
        t, w = vol.download()
    assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32))
    assert np.array_equal(lab, ref_l)
    assert np.array_equal(fp.view(np.uint32), ref_f.view(np.uint32)) and np.array_equal(bp.view(np.uint32), ref_b.view(np.uint32))
    with capi.Volume(cfg) as vol:   # the separate passes give the same arrays
        vol.labels_enable(0.5)
        for (c2w, _, _, _), (d, l, s_) in zip(frames, keep):
            vol.integrate_device(d.data_ptr(), c2w)
            vol.integrate_labels_device(d.data_ptr(), l.data_ptr(), s_.data_ptr(), c2w)
        lab2, fp2, bp2 = vol.download_labels()
    assert np.array_equal(lab2, lab) and np.array_equal(fp2.view(np.uint32), fp.view(np.uint32)) and np.array_equal(bp2.view(np.uint32), bp.view(np.uint32))
