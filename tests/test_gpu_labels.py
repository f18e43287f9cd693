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


def test_config4_slab_of_2048_cube_with_labels(cuda, oracle):
    """BASELINE configs[4] rehearsed on one GPU: rank 2 of 8 of a 2048^3 @ 2 mm grid (a 2048x2048x256
    slab, 8.6 GB TSDF+weight + 10.7 GB label state) takes one labelled frame; three 2-slice samples are
    checked against the oracle bit for bit (64-bit indexing: slab offsets exceed 2^31 elements)."""
    D, vs = 2048, 0.002
    dims = (D, D, D)
    zb, ze = 2 * D // 8, 3 * D // 8      # the slab that holds the visible cap of the sphere
    origin = np.array([-2.048, -2.048, 0.4], np.float32)
    cfg = capi.make_config(dims, vs, origin, z_begin=zb, z_end=ze)
    scene = synth.SurfScene(dims, vs, origin)
    rng = np.random.default_rng(9)
    c2w = scene.pose(2, n=16)
    depth = scene.depth(c2w, quantize=True)
    masks, labels, scores = instance_masks(rng, 6)
    want_lab, want_sc = oracle.compose_labels(masks, labels, scores)
    with capi.Volume(cfg) as vol:
        vol.labels_enable(0.5)
        m_dev, d_dev = cuda.from_numpy(masks).cuda(), cuda.from_numpy(depth).cuda()
        lab_dev = cuda.empty((480, 640), dtype=cuda.uint16, device="cuda")
        sc_dev = cuda.empty((480, 640), dtype=cuda.float32, device="cuda")
        vol.compose_labels(m_dev.data_ptr(), labels, scores, lab_dev.data_ptr(), sc_dev.data_ptr())
        vol.integrate_labels_device(d_dev.data_ptr(), lab_dev.data_ptr(), sc_dev.data_ptr(), c2w)
        vol.integrate_device(d_dev.data_ptr(), c2w)
        lab, fp, bp = vol.download_labels()
        # check the three slice pairs that received the most label evidence, plus the slab's first pair
        per_slice = np.count_nonzero(lab.reshape(ze - zb, -1), axis=1)
        picks = sorted(set([0] + [int(z) for z in np.argsort(per_slice)[-3:]]))
        seen = 0
        for zl in picks:
            zl = min(zl, ze - zb - 2)
            z0 = zb + zl
            st, sw = oracle.init_grid(dims, z0, z0 + 2)
            sl, sf, sb = np.zeros(st.size, np.uint16), np.zeros(st.size, np.float32), np.zeros(st.size, np.float32)
            oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, st, sw, z_begin=z0, z_end=z0 + 2)
            seen += oracle.integrate_labels(cfg.cam_K, c2w, depth, want_lab, want_sc, dims, origin, vs, cfg.trunc_margin,
                                            sl, sf, sb, z_begin=z0, z_end=z0 + 2)
            gt, gw = vol.copy_slices(zl, 2)
            lo, hi = zl * D * D, (zl + 2) * D * D
            assert np.array_equal(gw, sw) and np.array_equal(gt.view(np.uint32), st.view(np.uint32))
            assert np.array_equal(lab[lo:hi], sl) and np.array_equal(fp[lo:hi], sf) and np.array_equal(bp[lo:hi], sb)
    assert seen > 1000 and np.count_nonzero(lab) > 10000, (seen, int(np.count_nonzero(lab)))


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
