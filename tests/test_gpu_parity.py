"""GPU parity: the HIP Integrate path, called through the C ABI, against the CPU oracle.

Bar: BIT-EXACT fp32 (stronger than the 1e-5 relative tolerance BASELINE.json states).  The
kernel keeps the reference's operation order, IEEE division and no FMA contraction, so every
TSDF and weight value must equal the oracle's bit for bit; the tolerance form
abs(d) <= 1e-5*max(abs(ref),1) is asserted as well so a regression reports how far off it is.
"""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu

TOL = 1e-5  # BASELINE.json north_star: "SDF/weight within 1e-5 rel of CPU reference"


def assert_parity(got_t, got_w, ref_t, ref_w):
    assert np.array_equal(got_w, ref_w), f"weights differ at {np.flatnonzero(got_w != ref_w)[:5]}"
    err = np.abs(got_t.astype(np.float64) - ref_t)
    assert np.all(err <= TOL * np.maximum(np.abs(ref_t), 1.0)), f"max abs err {err.max()}"
    nbad = int(np.count_nonzero(got_t.view(np.uint32) != ref_t.view(np.uint32)))
    assert nbad == 0, f"{nbad} of {got_t.size} TSDF values not bit-identical"


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


CASES = [
    # dims, voxel, z0, base pose random?, frames
    ((64, 64, 64), 0.01, 0.8, False, 4),
    ((48, 40, 36), 0.02, 0.5, True, 5),
    ((200, 200, 40), 0.004, 0.6, True, 3),     # reference default row length / voxel size
    ((37, 29, 31), 0.02, 0.7, True, 4),        # dim_x % 4 != 0 -> scalar kernel
    ((256, 8, 8), 0.004, 1.0, False, 3),       # one full wavefront per row
    ((4, 4, 4), 0.05, 1.0, False, 2),          # tiny
    ((300, 5, 3), 0.003, 0.9, True, 3),        # ragged: partial wavefronts and partial blocks
]


@pytest.mark.parametrize("sync_every_frame", [True, False])   # False: the frames are collected and applied as one sequence
@pytest.mark.parametrize("dims,vs,z0,rand_base,frames", CASES)
def test_integrate_matches_oracle(cuda, oracle, dims, vs, z0, rand_base, frames, sync_every_frame):
    rng = np.random.default_rng(hash((dims, frames)) & 0xffff)
    origin = synth.surf_volume(max(dims), vs, z0)
    base = synth.random_pose(rng) if rand_base else synth.identity_pose()
    cfg = capi.make_config(dims, vs, origin, base2world=base)
    scene = synth.SurfScene(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Volume(cfg) as vol:
        t0, w0 = vol.download()
        assert np.all(t0 == 1.0) and np.all(w0 == 0.0)  # ref: src/tsdf.cu:79-81
        n_upd = 0
        keep = []
        for k in range(frames):
            c2w = synth.random_pose(rng, 0.35, 0.4)
            c2b = oracle.cam2base(base, c2w)
            depth = scene.depth(c2b, quantize=bool(k & 1))
            # a few invalid samples: zero, negative, beyond the 6 m cut-off (ref: src/tsdf.cu:46)
            ys, xs = rng.integers(0, 480, 300), rng.integers(0, 640, 300)
            depth[ys[:100], xs[:100]] = 0.0
            depth[ys[100:200], xs[100:200]] = -1.0
            depth[ys[200:], xs[200:]] = 6.5
            d_dev = dev(cuda, depth)
            vol.integrate_device(d_dev.data_ptr(), c2w)
            assert np.array_equal(vol.last_cam2base(), c2b), "host pose composition differs from the oracle"
            n_upd += oracle.integrate(cfg.cam_K, c2b, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
            if sync_every_frame:
                vol.sync()
            else:
                keep.append(d_dev)     # torch's stream is not the handle's: the buffer must outlive the queued copy
        got_t, got_w = vol.download()
    assert n_upd > 0, "test scene updated nothing: not a test"
    assert_parity(got_t, got_w, ref_t, ref_w)


# free-space summary (32+), summary + early loads (48+), early loads only (64+)
# ... and the exact shared-reciprocal projection with (80+) / without (96+) the summary
SUM_VARIANTS = [b + c for b in (32, 48, 64, 80, 96) for c in (2, 3, 6, 7, 10, 11)] + [115, 119]   # 11x: depth tile in LDS


@pytest.mark.parametrize("variant", capi.variants(*([0, 1, 3, 2] + list(range(16, 28)) + SUM_VARIANTS)))
def test_every_kernel_variant_is_bit_exact(cuda, oracle, variant):
    """Every one-frame kernel the loaded library has gives identical bits: the shipped ones (default, one launch per frame,
    scalar) and -- when the measurement build is loaded (TSDF_HIP_LIB=.../libtsdf_hip_exp.so) -- every stage of the ladder
    (rows/tile, R = 1/2/4, elision on/off, nt on/off, summary, early loads, fast projection, LDS depth tiles).
    The scene has free space (elided divisions), a truncation band and repeated frames, so both
    sides of every wave-uniform shortcut are taken; dim_y = 50 leaves ragged row groups.  dim_x = 256
    keeps the row-mapped kernels in play (other widths are served by the flat mapping, tested above)."""
    dims, vs = (256, 50, 24), 0.005
    origin = synth.surf_volume(256, vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(variant)
        for k in range(5):
            c2w = scene.pose(k % 3, n=5)  # frame 3, 4 repeat poses 0, 1: weights > 1 with tsdf != 1
            depth = scene.depth(c2w, quantize=True)
            d_dev = dev(cuda, depth)
            vol.integrate_device(d_dev.data_ptr(), c2w)
            vol.sync()
            oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        got_t, got_w = vol.download()
    assert np.count_nonzero(ref_t != 1.0) > 1000 and ref_w.max() >= 3
    assert_parity(got_t, got_w, ref_t, ref_w)


def test_free_space_summary_stays_consistent(cuda, oracle):
    """The free-space summary (flags: "this 256-voxel segment is all ones") across every event that
    can invalidate it: truncation-band updates, switching to kernels that do not maintain it and
    back, upload of a foreign state, reset."""
    dims, vs = (512, 24, 20), 0.005          # two segments per row
    origin = synth.surf_volume(512, vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)
    far = np.full((480, 640), 5.9, np.float32)   # free space everywhere: the fast path
    with capi.Volume(cfg) as vol:
        def step(depth, pose, variant):
            vol.set_kernel_variant(variant)
            d_dev = dev(cuda, depth)
            vol.integrate_device(d_dev.data_ptr(), pose)
            vol.sync()
            oracle.integrate(cfg.cam_K, pose, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)

        p0, p1 = scene.pose(0, 8), scene.pose(2, 8)
        step(far, p0, 0)                      # all segments stay "ones"
        step(scene.depth(p0), p0, 0)          # surface: some segments leave the ones state
        step(far, p1, 0)                      # free space again over changed segments: must use real TSDF
        step(scene.depth(p1), p1, 1)          # a kernel without summary support (the scalar one)
        step(far, p0, 0)                      # back on the summary kernel
        t, w = vol.download()
        assert_parity(t, w, ref_t, ref_w)
        # foreign state: zeros in "free" segments
        t2 = ref_t.copy(); t2[::3] = 0.25
        vol.upload(t2, ref_w)
        ref_t[:] = t2
        step(far, p1, 0)
        t, w = vol.download()
        assert_parity(t, w, ref_t, ref_w)
        vol.reset()
        ref_t[:], ref_w[:] = 1.0, 0.0
        step(far, p0, 0)
        step(scene.depth(p0), p0, 39 if capi.experiments_build() else 3)   # 39: the summary kernel without the fast projection (measurement build)
        t, w = vol.download()
        assert_parity(t, w, ref_t, ref_w)
        assert np.count_nonzero(ref_t != 1.0) > 1000
        # weights no counter can reach (-1 makes w+1 == 0, inf, huge): the free-space shortcut must
        # not be taken for their segments; values (NaN included) must still equal the oracle's
        vol.reset()
        ref_t[:], ref_w[:] = 1.0, 0.0
        w_odd = ref_w.copy()
        w_odd[5::4099] = -1.0
        w_odd[7::5003] = np.inf
        w_odd[11::6007] = 3.3e38
        w_odd[13::7001] = -0.5
        vol.upload(ref_t, w_odd)
        ref_w[:] = w_odd
        with np.errstate(all="ignore"):
            step(far, p0, 0)
            step(far, p1, 0)
        t, w = vol.download()
        assert np.array_equal(w, ref_w, equal_nan=True)
        assert np.array_equal(t, ref_t, equal_nan=True)
        assert np.isnan(ref_t).sum() > 0 and np.isfinite(ref_t).sum() > 0


def test_host_depth_path_equals_device_path(cuda, oracle):
    """tsdf_integrate (TSDF::Integrate semantics, host pointer) == device-resident path."""
    dims, vs = (64, 48, 40), 0.01
    origin = synth.surf_volume(64, vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Volume(cfg) as vol:
        for k in range(7):  # more frames than staging slots: exercises slot reuse
            c2w = scene.pose(k, n=7)
            depth = scene.depth(c2w)
            tmp = depth.copy()
            vol.integrate(tmp, c2w)
            tmp[:] = -5.0  # caller may reuse its buffer as soon as the call returns
            oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        got_t, got_w = vol.download()
    assert_parity(got_t, got_w, ref_t, ref_w)


def test_zslabs_equal_whole_grid(cuda, oracle):
    """Two z-slab handles == one whole-grid handle, bit for bit (global z in the kernel)."""
    dims, vs = (64, 32, 50), 0.013
    origin = synth.surf_volume(64, vs, 0.6)
    scene = synth.SurfScene(dims, vs, origin)
    cuts = [0, 17, 17, 50]  # includes an empty slab
    slabs = [capi.Volume(capi.make_config(dims, vs, origin, z_begin=a, z_end=b))
             for a, b in zip(cuts[:-1], cuts[1:])]
    whole = capi.Volume(capi.make_config(dims, vs, origin))
    ref_t, ref_w = oracle.init_grid(dims)
    for k in range(3):
        c2w = scene.pose(k, n=3)
        d_dev = dev(cuda, scene.depth(c2w))
        for v in slabs + [whole]:
            v.integrate_device(d_dev.data_ptr(), c2w)
            v.sync()
        oracle.integrate(whole.cfg.cam_K, c2w, scene.depth(c2w), dims, origin, vs,
                         whole.cfg.trunc_margin, ref_t, ref_w)
    parts = [v.download() for v in slabs]
    t = np.concatenate([p[0] for p in parts])
    w = np.concatenate([p[1] for p in parts])
    wt, ww = whole.download()
    assert_parity(t, w, wt, ww)
    assert_parity(t, w, ref_t, ref_w)
    # and the oracle's own slab form agrees with its whole-grid form
    st, sw = oracle.init_grid(dims, 17, 50)
    for k in range(3):
        c2w = scene.pose(k, n=3)
        oracle.integrate(whole.cfg.cam_K, c2w, scene.depth(c2w), dims, origin, vs,
                         whole.cfg.trunc_margin, st, sw, z_begin=17, z_end=50)
    assert np.array_equal(st, ref_t[17 * 32 * 64:]) and np.array_equal(sw, ref_w[17 * 32 * 64:])
    for v in slabs + [whole]:
        v.close()


def test_masked_integrate(cuda, oracle):
    """depth * (mask/255) fused into the kernel == oracle on the pre-multiplied image."""
    dims, vs = (64, 64, 32), 0.01
    origin = synth.surf_volume(64, vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    rng = np.random.default_rng(7)
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Volume(cfg) as vol:
        for k in range(3):
            c2w = scene.pose(k, n=3)
            depth = scene.depth(c2w)
            mask = np.zeros((480, 640), np.uint8)
            y0, x0 = rng.integers(100, 200, 2)
            mask[y0:y0 + 250, x0:x0 + 300] = 255  # MaskRCNN output format {0,255}
            d_dev, m_dev = dev(cuda, depth), dev(cuda, mask)
            vol.integrate_masked_device(d_dev.data_ptr(), m_dev.data_ptr(), c2w)
            vol.sync()
            masked = oracle.mask_depth(depth, mask)
            oracle.integrate(cfg.cam_K, c2w, masked, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        got_t, got_w = vol.download()
    assert ref_w.max() > 0
    assert_parity(got_t, got_w, ref_t, ref_w)


def test_edge_inputs(cuda, oracle):
    """All-invalid depth, camera behind the volume, volume outside the image: nothing changes."""
    dims, vs = (32, 32, 32), 0.01
    origin = synth.surf_volume(32, vs, 1.0)
    cfg = capi.make_config(dims, vs, origin)
    with capi.Volume(cfg) as vol:
        zero = dev(cuda, np.zeros((480, 640), np.float32))
        far = dev(cuda, np.full((480, 640), 6.0001, np.float32))
        ok = dev(cuda, np.full((480, 640), 2.0, np.float32))
        vol.integrate_device(zero.data_ptr(), synth.identity_pose())
        vol.integrate_device(far.data_ptr(), synth.identity_pose())
        behind = synth.make_pose(synth.rot_y(np.pi), [0, 0, 0])  # looking away
        vol.integrate_device(ok.data_ptr(), behind)
        aside = synth.make_pose(np.eye(3), [50.0, 0, 0])  # volume far outside the frustum
        vol.integrate_device(ok.data_ptr(), aside)
        t, w = vol.download()
        assert np.all(t == 1.0) and np.all(w == 0.0)
        # exactly 6.0 m is still valid (ref: `depth_val > 6` rejects, src/tsdf.cu:46)
        six = dev(cuda, np.full((480, 640), 6.0, np.float32))
        vol.integrate_device(six.data_ptr(), synth.identity_pose())
        t, w = vol.download()
        rt, rw = oracle.init_grid(dims)
        oracle.integrate(cfg.cam_K, synth.identity_pose(), np.full((480, 640), 6.0, np.float32), dims,
                         origin, vs, cfg.trunc_margin, rt, rw)
        assert rw.sum() > 0
        assert_parity(t, w, rt, rw)
        vol.reset()
        t, w = vol.download()
        assert np.all(t == 1.0) and np.all(w == 0.0)


def test_singular_base_pose_is_ignored_like_the_reference(cuda, oracle):
    """ref: src/tsdf.cu:74 ignores invert_matrix's false -> base2world_inv stays zero."""
    dims, vs = (16, 16, 16), 0.02
    origin = synth.surf_volume(16, vs, 1.0)
    cfg = capi.make_config(dims, vs, origin, base2world=np.zeros(16, np.float32))
    with capi.Volume(cfg) as vol:
        d = dev(cuda, np.full((480, 640), 2.0, np.float32))
        vol.integrate_device(d.data_ptr(), synth.identity_pose())
        c2b = vol.last_cam2base()
        assert np.array_equal(c2b, oracle.cam2base(np.zeros(16, np.float32), synth.identity_pose()))
        t, w = vol.download()
        rt, rw = oracle.init_grid(dims)
        oracle.integrate(cfg.cam_K, c2b, np.full((480, 640), 2.0, np.float32), dims, origin, vs,
                         cfg.trunc_margin, rt, rw)
        assert_parity(t, w, rt, rw)


def test_surface_and_files_match_oracle(cuda, oracle, tmp_path):
    """Device compaction == the reference's host scan; .ply/.bin byte-identical to the oracle's."""
    dims, vs = (72, 56, 40), 0.01
    origin = synth.surf_volume(72, vs, 0.6)
    cfg = capi.make_config(dims, vs, origin, vol_id=3)
    scene = synth.SurfScene(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Volume(cfg) as vol:
        for k in range(4):
            c2w = scene.pose(k, n=4)
            depth = scene.depth(c2w)
            vol.integrate(depth, c2w)
            oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        # make some TSDF values exactly zero so the |tsdf| != 0 half of the test matters
        t, w = vol.download()
        idx = np.flatnonzero(w > 0)[::97]
        t[idx] = 0.0
        ref_t[idx] = 0.0
        vol.upload(t, w)
        want = oracle.surface_points(ref_t, ref_w, dims, vs, origin)
        assert 0 < len(want) < t.size
        assert vol.count_surface() == len(want)
        got = vol.extract_surface()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        vol.save_ply(str(tmp_path / "a.ply"))
        vol.save_bin(str(tmp_path / "a.bin"))
    oracle.save_ply(str(tmp_path / "b.ply"), ref_t, ref_w, dims, vs, origin)
    oracle.save_bin(str(tmp_path / "b.bin"), ref_t, dims, origin, vs, cfg.trunc_margin)
    assert (tmp_path / "a.ply").read_bytes() == (tmp_path / "b.ply").read_bytes()
    assert (tmp_path / "a.bin").read_bytes() == (tmp_path / "b.bin").read_bytes()
    from oracle.oracle import RefHost      # the reference's own .ply writer, compiled as it stands (oracle/ref_host_driver.cpp)
    if RefHost.available():
        RefHost().save_ply(str(tmp_path / "r.ply"), ref_t, ref_w, dims, vs, origin)
        assert (tmp_path / "a.ply").read_bytes() == (tmp_path / "r.ply").read_bytes()


@pytest.mark.parametrize("path", ["frame", "fused", "fused_classified", "fused_per_voxel"])
def test_full_size_512_sfull(cuda, oracle, path):
    """BASELINE config[1]: 512^3 @ 5 mm, full-coverage input, through the one-launch-per-frame kernel and through the
    fused sequence path (default, classification forced on, forced off).  Checked against the oracle on three
    z-slabs and through size-independent properties on the whole grid."""
    D, vs = 512, 0.005
    origin = synth.sfull_volume(D, vs)
    cfg = capi.make_config((D, D, D), vs, origin)
    depth = synth.sfull_depth()
    frames = 3 if path == "frame" else 35          # 35: one full 32-frame pass + a 3-frame one
    poses = np.stack([synth.sfull_pose(k) for k in range(frames)])
    with capi.Volume(cfg) as vol:
        d_dev = dev(cuda, depth)
        if path == "frame":
            vol.set_deferral(0)                    # one kernel per call: the bench headline's kernel
            for k in range(frames):
                vol.integrate_device(d_dev.data_ptr(), poses[k])
        else:
            vol.set_kernel_variant({"fused": 0, "fused_classified": 8, "fused_per_voxel": 7}[path])
            vol.integrate_frames_device([d_dev.data_ptr()] * frames, poses)
        t, w = vol.download()
    # property: every voxel updated every frame, and the mean of dist = 1 stays exactly 1
    assert np.all(w == float(frames)), "S-full must update every voxel every frame (N_upd == N)"
    assert np.all(t == 1.0)
    # oracle on three slabs (first, middle, last 4 slices)
    for zb in (0, 254, 508):
        st, sw = oracle.init_grid((D, D, D), zb, zb + 4)
        n = 0
        for k in range(frames):
            n += oracle.integrate(cfg.cam_K, poses[k], depth, (D, D, D), origin, vs,
                                  cfg.trunc_margin, st, sw, z_begin=zb, z_end=zb + 4)
        assert n == frames * 4 * D * D
        lo, hi = zb * D * D, (zb + 4) * D * D
        assert_parity(t[lo:hi], w[lo:hi], st, sw)


@pytest.mark.parametrize("path", ["frame", "fused", "fused_classified"])
def test_full_size_512_sband(cuda, oracle, path):
    """The bench headline's workload at full size: 512^3 @ 5 mm S-band -- every voxel updated by every frame INSIDE the
    truncation band (4 m margin), so dist < 1, both divisions run and every TSDF value changes every frame; nothing can
    be elided or claimed.  Per-frame launches and the fused sequence path against the oracle on four z-slabs, bit for
    bit, plus whole-grid properties."""
    D, vs = 512, 0.005
    origin = synth.sband_volume(D, vs)
    cfg = capi.make_config((D, D, D), vs, origin, trunc=synth.SBAND_TRUNC)
    depth = synth.sfull_depth()
    frames = 5 if path == "frame" else 37
    poses = np.stack([synth.sband_pose(k) for k in range(frames)])
    with capi.Volume(cfg) as vol:
        d_dev = dev(cuda, depth)
        if path == "frame":
            vol.set_deferral(0)                    # one kernel per call: the bench headline's kernel
            for k in range(frames):
                vol.integrate_device(d_dev.data_ptr(), poses[k])
        else:
            vol.set_kernel_variant(0 if path == "fused" else 8)
            vol.integrate_frames_device([d_dev.data_ptr()] * frames, poses)
        t, w = vol.download()
    assert np.all(w == float(frames)), "S-band must update every voxel every frame (N_upd == N)"
    assert 0.0 < t.min() and t.max() < 1.0, "every voxel inside the truncation band"
    for zb in (0, 171, 340, 509):
        st, sw = oracle.init_grid((D, D, D), zb, zb + 3)
        n = 0
        for k in range(frames):
            n += oracle.integrate(cfg.cam_K, poses[k], depth, (D, D, D), origin, vs,
                                  cfg.trunc_margin, st, sw, z_begin=zb, z_end=zb + 3, threads=8)
        assert n == frames * 3 * D * D
        lo, hi = zb * D * D, (zb + 3) * D * D
        assert_parity(t[lo:hi], w[lo:hi], st, sw)


def test_full_size_512_surface_scene(cuda, oracle):
    """512^3 with a real surface: whole-grid bit parity against the (multi-threaded) oracle."""
    D, vs = 512, 0.005
    origin = synth.surf_volume(D, vs, 1.0)
    cfg = capi.make_config((D, D, D), vs, origin)
    scene = synth.SurfScene((D, D, D), vs, origin)
    ref_t, ref_w = oracle.init_grid((D, D, D))
    with capi.Volume(cfg) as vol:
        for k in range(2):
            c2w = scene.pose(k * 9, n=64)
            depth = scene.depth(c2w, quantize=True)
            vol.integrate(depth, c2w)
            oracle.integrate(cfg.cam_K, c2w, depth, (D, D, D), origin, vs, cfg.trunc_margin, ref_t, ref_w)
        got_t, got_w = vol.download()
    frac = float(np.count_nonzero(ref_w)) / ref_w.size
    assert 0.05 < frac < 0.95, f"updated fraction {frac}: scene should mix updated and skipped voxels"
    assert_parity(got_t, got_w, ref_t, ref_w)


def test_caller_owned_stream(cuda, oracle):
    """tsdf_set_stream: the handle's work runs on a stream the caller owns (here a PyTorch stream), so
    the caller's own events and synchronisation cover it."""
    dims, vs = (64, 64, 32), 0.01
    origin = synth.surf_volume(64, vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)
    stream = cuda.cuda.Stream()
    with capi.Volume(cfg) as vol:
        vol.set_stream(stream.cuda_stream)
        with cuda.cuda.stream(stream):
            for k in range(3):
                c2w = scene.pose(k, n=5)
                depth = scene.depth(c2w)
                d_dev = cuda.from_numpy(depth).cuda()            # allocated and filled on `stream`
                vol.integrate_device(d_dev.data_ptr(), c2w)
                oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
            done = cuda.cuda.Event()
            done.record(stream)
        done.synchronize()                                        # the caller's event waits for our kernels
        t, w = vol.download()
        vol.set_stream(None)                                      # back on the handle's own stream
        vol.integrate(scene.depth(scene.pose(0, 5)), scene.pose(0, 5))
        oracle.integrate(cfg.cam_K, scene.pose(0, 5), scene.depth(scene.pose(0, 5)), dims, origin, vs,
                         cfg.trunc_margin, ref_t, ref_w)
        t, w = vol.download()
    assert_parity(t, w, ref_t, ref_w)


def test_checkpoint_resume(cuda, oracle, tmp_path):
    """Interrupted fusion continues bit-exactly from tsdf_save_state / tsdf_load_state; a reference-format
    .bin (TSDF only) written by the oracle's writer loads through tsdf_load_bin."""
    dims, vs = (96, 64, 40), 0.01
    origin = synth.surf_volume(96, vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    frames = [(scene.pose(k, 8), scene.depth(scene.pose(k, 8), quantize=True)) for k in range(6)]
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Volume(cfg) as a:
        for c2w, d in frames[:3]:
            a.integrate(d, c2w)
            oracle.integrate(cfg.cam_K, c2w, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        a.save_state(str(tmp_path / "half.state"))
        oracle.save_bin(str(tmp_path / "half.bin"), ref_t, dims, origin, vs, cfg.trunc_margin)
    with capi.Volume(cfg) as b:
        b.load_state(str(tmp_path / "half.state"))
        for c2w, d in frames[3:]:
            b.integrate(d, c2w)
            oracle.integrate(cfg.cam_K, c2w, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
        t, w = b.download()
        assert_parity(t, w, ref_t, ref_w)
        b.load_bin(str(tmp_path / "half.bin"))          # reference format: TSDF only, weights stay
        t2, w2 = b.download()
        assert np.array_equal(w2, ref_w) and not np.array_equal(t2, t)
        with pytest.raises(capi.TsdfError):
            b.load_state(str(tmp_path / "half.bin"))     # wrong format is refused
    with capi.Volume(capi.make_config((32, 32, 32), vs, origin)) as c:
        with pytest.raises(capi.TsdfError):
            c.load_state(str(tmp_path / "half.state"))   # wrong slab is refused


def test_files_larger_than_a_staging_piece_and_a_full_disk(cuda, oracle, tmp_path):
    """The writers stream device memory to the file in 32 MiB pieces through two pinned buffers: a 256 x 256 x 200 volume
    (52 MB per array, 63 MB of surface points: more than one piece, the last one partial) gives files byte-identical to the
    oracle's writers and a checkpoint that loads back bit for bit; a disk that takes no byte (/dev/full) is reported as
    TSDF_ERR_IO by every writer and leaves the handle usable."""
    dims, vs = (256, 256, 200), 0.008
    origin = synth.surf_volume(256, vs, 0.8)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)
    with capi.Volume(cfg) as vol:
        for k in range(3):
            c2w = scene.pose(4 * k, 16)
            d = scene.depth(c2w, quantize=True)
            vol.integrate(d, c2w)
            oracle.integrate(cfg.cam_K, c2w, d, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, threads=8)
        vol.save_ply(str(tmp_path / "a.ply"))
        vol.save_bin(str(tmp_path / "a.bin"))
        vol.save_state(str(tmp_path / "a.state"))
        n_pts = vol.count_surface()
        assert n_pts * 12 > (32 << 20) and (n_pts * 12) % (32 << 20) != 0, "the point list should span more than one piece, the last one partial"
        for name, fn in (("save_ply", vol.save_ply), ("save_bin", vol.save_bin), ("save_state", vol.save_state)):
            with pytest.raises(capi.TsdfError) as e:
                fn("/dev/full")
            assert "short write" in str(e.value) or "cannot" in str(e.value), name
        t, w = vol.download()                               # still usable, nothing changed
        assert_parity(t, w, ref_t, ref_w)
    oracle.save_ply(str(tmp_path / "b.ply"), ref_t, ref_w, dims, vs, origin)
    oracle.save_bin(str(tmp_path / "b.bin"), ref_t, dims, origin, vs, cfg.trunc_margin)
    assert (tmp_path / "a.ply").read_bytes() == (tmp_path / "b.ply").read_bytes()
    assert (tmp_path / "a.bin").read_bytes() == (tmp_path / "b.bin").read_bytes()
    with capi.Volume(cfg) as other:
        other.load_state(str(tmp_path / "a.state"))
        t, w = other.download()
        assert_parity(t, w, ref_t, ref_w)


@pytest.mark.parametrize("dims", [(256, 24, 20), (200, 24, 20), (37, 20, 16)])   # row mapping, flat mapping, scalar kernel
def test_non_finite_depth_samples(cuda, oracle, dims):
    """NaN depth passes both depth tests of the reference (every comparison with NaN is false, ref: src/tsdf.cu:46,49)
    and updates with dist = fmin(1, NaN) = 1; +-inf, -0, denormals and the 6 m boundary take the ordinary branches.
    One launch per frame, fused launches and the first-version kernel must all reproduce that (the oracle is checked
    against the reference's own body on the same kind of input in tests/test_oracle_vs_ref.py)."""
    rng = np.random.default_rng(23)
    vs = 2.0 / dims[0]
    origin = synth.surf_volume(dims[0], vs, 0.6)
    cfg = capi.make_config(dims, vs, origin)
    sc = synth.SurfScene(dims, vs, origin)
    odd = np.array([np.nan, np.inf, -np.inf, -0.0, 1e-42, -1e-42, 6.0, np.nextafter(np.float32(6.0), np.float32(7.0))], np.float32)
    frames = []
    for k in range(5):
        c2w = sc.pose(k, 8)
        depth = sc.depth(c2w, quantize=True)
        depth[rng.integers(0, 480, 60000), rng.integers(0, 640, 60000)] = rng.choice(odd, 60000)
        frames.append((c2w, depth))
    ref_t, ref_w = oracle.init_grid(dims)
    for c2w, depth in frames:
        oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w)
    assert np.isfinite(ref_t).all() and ref_w.max() >= 3
    keep = [dev(cuda, d) for _, d in frames]
    poses = np.stack([c for c, _ in frames])
    for variant, fused in [(0, False), (0, True), (3, False), (1, False)] + ([(2, False), (17, False)] if capi.experiments_build() else []):
        if dims[0] % 4 and (fused or variant):
            continue
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(variant)
            if fused:
                vol.integrate_frames_device([d.data_ptr() for d in keep], poses)
            else:
                for d, c2w in zip(keep, poses):
                    vol.integrate_device(d.data_ptr(), c2w)
            t, w = vol.download()
        assert_parity(t, w, ref_t, ref_w)


@pytest.mark.parametrize("dims", [(256, 12, 12), (12, 12, 12)])   # row mapping (fast-projection candidates), flat mapping
def test_degenerate_poses(cuda, oracle, dims):
    """Zero, vanishing, huge, NaN and infinite pose entries: the host bounds send them down the generic IEEE-division
    path, which takes the same branches as the reference body (tests/test_oracle_vs_ref.py::test_degenerate_poses_match
    pins the oracle on these), one frame per launch and fused."""
    vs = 0.6 / dims[0]
    origin = np.array([-0.3, -0.3, -0.3], np.float32)
    cfg = capi.make_config(dims, vs, origin)
    depth = np.full((480, 640), 1.0, np.float32)
    nan_pose = synth.identity_pose(); nan_pose[5] = np.nan
    inf_pose = synth.identity_pose(); inf_pose[3] = np.inf
    poses = [np.zeros(16, np.float32), synth.make_pose(np.eye(3) * 1e-30, [0, 0, 0]), synth.make_pose(np.eye(3) * 1e30, [0, 0, 0]),
             synth.make_pose(np.eye(3), [0, 0, 1e20]), synth.make_pose(np.eye(3) * 1e-20, [0, 0, -1e-20]), nan_pose, inf_pose,
             synth.make_pose(synth.rot_x(0.2), [0.0, 0.0, -0.9]), synth.identity_pose()]
    rt, rw = oracle.init_grid(dims)
    with np.errstate(all="ignore"):
        c2bs = [oracle.cam2base(synth.identity_pose(), p) for p in poses]
    for c2b in c2bs:
        oracle.integrate(cfg.cam_K, c2b, depth, dims, origin, vs, cfg.trunc_margin, rt, rw)
    assert rw.max() >= 1
    d = dev(cuda, depth)
    with capi.Volume(cfg) as vol:
        for p in poses:
            vol.integrate_device(d.data_ptr(), p)
        t, w = vol.download()
    assert np.array_equal(w, rw, equal_nan=True) and np.array_equal(t.view(np.uint32), rt.view(np.uint32))
    with capi.Volume(cfg) as vol:
        vol.integrate_frames_device([d.data_ptr()] * len(poses), np.stack(poses))
        t, w = vol.download()
    assert np.array_equal(w, rw, equal_nan=True) and np.array_equal(t.view(np.uint32), rt.view(np.uint32))


def test_image_larger_than_2_pow_24_pixels(cuda, oracle):
    """4200 x 4000 pixels: the pixel index is no longer exact in fp32, so the host withdraws the fast projection
    (fast_ok) and every launch runs the generic path; the 32-bit byte offset of the gather still holds (< 2^30 pixels)."""
    h, w = 4000, 4200
    assert h * w > 1 << 24
    dims, vs = (64, 48, 24), 0.02
    origin = synth.surf_volume(64, vs, 0.8)
    K = np.array([3500.0, 0, 2100.3, 0, 3500.0, 1999.6, 0, 0, 1], np.float32)
    cfg = capi.make_config(dims, vs, origin, K=K, im_height=h, im_width=w)
    rng = np.random.default_rng(3)
    depth = rng.uniform(0.5, 2.5, (h, w)).astype(np.float32)
    poses = [synth.identity_pose(), synth.make_pose(synth.rot_y(0.1), [0.1, 0.0, 0.0]), synth.make_pose(synth.rot_x(-0.05), [0.0, 0.05, 0.1])]
    rt, rw = oracle.init_grid(dims)
    for p in poses:
        oracle.integrate(K, p, depth, dims, origin, vs, cfg.trunc_margin, rt, rw)
    assert rw.sum() > 0.3 * rw.size
    d = dev(cuda, depth)
    with capi.Volume(cfg) as vol:
        vol.integrate_device(d.data_ptr(), poses[0])
        vol.integrate_frames_device([d.data_ptr()] * 2, np.stack(poses[1:]))
        t, wgt = vol.download()
    assert_parity(t, wgt, rt, rw)
