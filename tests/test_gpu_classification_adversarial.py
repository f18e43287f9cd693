"""Adversarial inputs for the patch classification (tsdf_multiframe.hip.h, classify_patch) at a scale where hundreds of
workgroups classify: depth discontinuities exactly on the 16-pixel tile borders at the depths where a claim must
stop, patches whose projected box touches the image borders +- the pixel margin, cameras whose distance puts the
patch corners at the cz_short threshold, non-finite and invalid pixels inside otherwise claimable tiles.

Every case runs the fused sequence path with the classification forced on (variant 8: 64 x 4 bricks classified per
wavefront; variant 11: 256 x 1 rows classified per workgroup), decided per launch (variant 0) and forbidden (variant 7), and must equal the oracle bit for bit; with it forced on, claims must
actually be made (counters), so the test is about claims that are right, not claims that are absent."""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu

H, W = 480, 640
GRIDS = [((256, 64, 32), 0.002), ((200, 120, 32), 0.0025)]   # row mapping (512 workgroups), flat mapping (750)


def run_case(cuda, oracle, dims, vs, origin, K, frames, expect_claims=True, trunc=None):
    """frames: [(cam2world, depth)].  Oracle once, then the three variants."""
    cfg = capi.make_config(dims, vs, origin, K=K, trunc=trunc)
    ref_t, ref_w = oracle.init_grid(dims)
    with np.errstate(invalid="ignore"):
        for c2w, depth in frames:
            oracle.integrate(cfg.cam_K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, threads=8)
    keep = [cuda.from_numpy(np.ascontiguousarray(d, np.float32)).cuda() for _, d in frames]
    poses = np.stack([p for p, _ in frames])
    for variant in capi.variants(8, 0, 7, 12, 13, 11):     # bricks per wavefront with / without the super-brick pre-pass, rows per workgroup (all forced), per launch, never
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(variant)
            if variant in (8, 11, 12, 13):
                vol.shortcut_stats(True)
            vol.integrate_frames_device([d.data_ptr() for d in keep], poses)
            if variant in (8, 11, 12, 13):
                per_voxel, free, skipped = vol.shortcut_stats(False)
                assert per_voxel + free + skipped > 0
                if expect_claims:
                    assert free + skipped > 0, "no workgroup-frame was claimed: the case does not exercise the classification"
            t, w = vol.download()
        assert np.array_equal(w, ref_w), f"variant {variant}: weights differ at {np.flatnonzero(w != ref_w)[:5]}"
        bad = np.flatnonzero(t.view(np.uint32) != ref_t.view(np.uint32))
        assert bad.size == 0, f"variant {variant}: {bad.size} TSDF values differ, first at {bad[:5]}"
    return ref_t, ref_w


@pytest.mark.parametrize("dims,vs", GRIDS)
def test_depth_steps_on_tile_borders(cuda, oracle, dims, vs):
    """Piecewise-constant depth whose discontinuities lie exactly on multiples of 16 pixels, at the depths where the
    free-space claim (d >= cz_max + trunc) and the skip claim (d <= cz_min - trunc) flip, seen from poses that slide
    the projected voxels across the tile borders in quarter-pixel steps."""
    origin = synth.surf_volume(dims[0], vs, 0.7)
    near, far = float(origin[2]), float(origin[2] + dims[2] * vs)
    trunc = float(np.float32(vs) * np.float32(5))
    rng = np.random.default_rng(3)
    levels = np.array([far + 0.5, far + trunc, np.nextafter(np.float32(far + trunc), np.float32(0)), far + trunc * 1.01,
                       far - 0.5 * (far - near), near + trunc, near - trunc, np.nextafter(np.float32(near - trunc), np.float32(9)),
                       near - trunc * 1.01, near - 0.2, (near + far) / 2 + trunc, 0.0], np.float32)
    frames = []
    px = near / float(synth.TUM_K[0])           # metres per pixel at the near face
    for k in range(12):
        tiles = rng.integers(0, len(levels), (H // 16, W // 16))
        if k % 3 == 0:
            tiles[:] = 0                        # everything free space, with
            tiles[rng.integers(0, H // 16, 40), rng.integers(0, W // 16, 40)] = rng.integers(1, len(levels), 40)   # islands
        depth = np.kron(levels[tiles], np.ones((16, 16), np.float32)).astype(np.float32)
        pose = synth.make_pose(synth.rot_z(0.01 * (k % 4)), [0.25 * k * px, -0.25 * k * px, 0.0])
        frames.append((pose, depth))
    run_case(cuda, oracle, dims, vs, origin, None, frames)


@pytest.mark.parametrize("edge", ["left", "right", "top", "bottom"])
@pytest.mark.parametrize("dims,vs", GRIDS)
def test_patches_touching_the_image_border(cuda, oracle, dims, vs, edge):
    """Principal point chosen so that one face of the volume projects onto an image border; the camera then slides in
    quarter-pixel steps so projected patch boxes cross 0 / W-1 / H-1 and the +-px_margin band around them.  Constant
    depth behind the volume (free-space claims need the box INSIDE the image) and in front of it (skip claims)."""
    origin = synth.surf_volume(dims[0], vs, 0.7)
    near, far = float(origin[2]), float(origin[2] + dims[2] * vs)
    fx, fy = 535.4, 539.2
    x0, x1 = float(origin[0]), float(origin[0]) + dims[0] * vs      # faces of the volume
    y0, y1 = float(origin[1]), float(origin[1]) + dims[1] * vs
    K = np.array(synth.TUM_K, np.float32)
    if edge == "left":
        K[2] = -fx * x0 / near                  # x = x0 at the near face lands on u = 0
    elif edge == "right":
        K[2] = (W - 1) - fx * x1 / near
    elif edge == "top":
        K[5] = -fy * y0 / near
    else:
        K[5] = (H - 1) - fy * y1 / near
    px = near / fx
    frames = []
    for k in range(-10, 11):
        shift = [0.25 * k * px, 0.0, 0.0] if edge in ("left", "right") else [0.0, 0.25 * k * px, 0.0]
        depth = np.full((H, W), far + 0.5 if k % 2 == 0 else near - 0.2, np.float32)
        frames.append((synth.make_pose(np.eye(3), shift), depth))
    run_case(cuda, oracle, dims, vs, origin, K, frames)


@pytest.mark.parametrize("dims,vs", GRIDS)
def test_camera_at_the_near_plane_threshold(cuda, oracle, dims, vs):
    """The classification trusts a projected box only when every corner has cz > cz_short (host: the bound on the slab's
    camera-frame coordinates / 64).  The camera approaches the near face so that the first slices' corners pass through
    that threshold in steps of a few ulp and then in coarser steps; further slices stay claimable."""
    origin = synth.surf_volume(dims[0], vs, 0.7)
    near, far = float(origin[2]), float(origin[2] + dims[2] * vs)
    ext = np.array(dims, np.float64) * vs
    frames = []
    # the host's bound for an axis-aligned camera at distance c from the near face (tsdf_capi.hip, make_params)
    def cz_short(c):
        dmax = np.array([ext[0] / 2, ext[1] / 2, c + ext[2]]) * 1.001
        return float(max(dmax) / 64.0 * 1.0001)
    c0 = 0.0040
    for _ in range(20):
        c0 = cz_short(c0)                       # fixed point: a camera whose near-face cz equals its own threshold
    cs = [np.float32(c0)]
    for _ in range(4):
        cs.append(np.nextafter(cs[-1], np.float32(1)))
    lo = np.float32(c0)
    for _ in range(4):
        lo = np.nextafter(lo, np.float32(0))
        cs.append(lo)
    cs += [np.float32(c0 * f) for f in (0.5, 0.9, 0.99, 1.01, 1.1, 2.0, 8.0)]
    for i, c in enumerate(cs):
        depth = np.full((H, W), far - near + float(c) + (0.5 if i % 2 == 0 else -0.01), np.float32)   # behind / inside
        frames.append((synth.make_pose(np.eye(3), [0.0, 0.0, near - float(c)]), depth))
    run_case(cuda, oracle, dims, vs, origin, None, frames)


@pytest.mark.parametrize("dims,vs", GRIDS)
def test_non_finite_pixels_inside_claimable_tiles(cuda, oracle, dims, vs):
    """A frame that is all free space (or all in front of the volume) except single NaN / +-inf / zero / negative /
    just-beyond-max-depth pixels at tile centres, corners and borders.  NaN depth DOES update a voxel in the reference
    (every comparison with NaN is false, ref: src/tsdf.cu:46,49), so a tile holding one may claim nothing."""
    origin = synth.surf_volume(dims[0], vs, 0.7)
    near, far = float(origin[2]), float(origin[2] + dims[2] * vs)
    odd = np.array([np.nan, np.inf, -np.inf, 0.0, -1.0, np.nextafter(np.float32(6.0), np.float32(7.0)), 1e-42], np.float32)
    rng = np.random.default_rng(5)
    frames = []
    for k in range(10):
        depth = np.full((H, W), far + 0.5 if k % 2 == 0 else near - 0.2, np.float32)
        n = 30
        ty, tx = rng.integers(0, H // 16, n), rng.integers(0, W // 16, n)
        oy, ox = rng.choice([0, 7, 15], n), rng.choice([0, 8, 15], n)
        depth[ty * 16 + oy, tx * 16 + ox] = rng.choice(odd, n)
        pose = synth.make_pose(synth.rot_y(0.01 * (k % 3)), [0.001 * k, 0.0, 0.0])
        frames.append((pose, depth))
    ref_t, ref_w = run_case(cuda, oracle, dims, vs, origin, None, frames)
    assert np.isfinite(ref_t).all()


def test_large_image_tile_table_beyond_lds(cuda, oracle):
    """A 1024 x 768 image has 64 x 48 = 3072 depth tiles: more than the sparse-table kernel keeps in LDS, so its upper
    levels are built from memory.  Claims (free space, skipped) and bit-exactness as for the small images."""
    h, w = 768, 1024
    dims, vs = (256, 48, 24), 0.003
    origin = synth.surf_volume(256, vs, 0.9)
    K = np.array([820.0, 0, 511.3, 0, 825.0, 383.6, 0, 0, 1], np.float32)
    cfg = capi.make_config(dims, vs, origin, K=K, im_height=h, im_width=w)
    near, far = float(origin[2]), float(origin[2] + dims[2] * vs)
    rng = np.random.default_rng(9)
    frames = []
    for k in range(6):
        depth = np.full((h, w), far + 0.4 if k % 2 == 0 else near - 0.1, np.float32)
        depth[rng.integers(0, h, 50), rng.integers(0, w, 50)] = rng.choice(np.array([0.0, np.nan, far - 0.02], np.float32), 50)
        if k == 4:
            depth[:, : w // 2] = (near + far) / 2
        frames.append((synth.make_pose(synth.rot_z(0.02 * k), [0.002 * k, 0.0, 0.0]), depth))
    ref_t, ref_w = oracle.init_grid(dims)
    with np.errstate(invalid="ignore"):
        for c2w, depth in frames:
            oracle.integrate(K, c2w, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, threads=8)
    keep = [cuda.from_numpy(d).cuda() for _, d in frames]
    for variant in (8, 0, 7):
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(variant)
            if variant == 8:
                vol.shortcut_stats(True)
            vol.integrate_frames_device([d.data_ptr() for d in keep], np.stack([p for p, _ in frames]))
            if variant == 8:
                per_voxel, free, skipped = vol.shortcut_stats(False)
                assert free > 0 and skipped > 0
            t, wgt = vol.download()
        assert np.array_equal(wgt, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)), variant


def test_unaligned_frame_buffers(cuda, oracle):
    """Depth frames and masks at addresses that are not multiples of 16 / 4 bytes (views into larger buffers): the tile
    summary's whole-row reads need aligned rows and must fall back to element reads -- same tables, same claims, same bits."""
    dims, vs = (256, 64, 32), 0.002
    origin = synth.surf_volume(256, vs, 0.7)
    cfg = capi.make_config(dims, vs, origin)
    scene = synth.SurfScene(dims, vs, origin)
    near, far = float(origin[2]), float(origin[2] + dims[2] * vs)
    frames = []
    for k in range(5):
        pose = scene.pose(k, 9)
        depth = scene.depth(pose, quantize=True) if k % 2 else np.full((H, W), far + 0.3, np.float32)
        mask = np.zeros((H, W), np.uint8)
        if k % 2:
            mask[225:275, 180 + 10 * k:460] = 255       # narrower than the volume's footprint: bricks outside it see nothing
        else:
            mask[40 + 10 * k:400, 60:600 - 20 * k] = 255
        frames.append((pose, depth, mask))
    ref_t, ref_w = oracle.init_grid(dims)
    for pose, depth, mask in frames:
        oracle.integrate(cfg.cam_K, pose, oracle.mask_depth(depth, mask), dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, threads=8)
    stats = {}
    for d_off, m_off in ((0, 0), (1, 1), (3, 2)):      # elements: depth pointer % 16 = 0, 4, 12; mask pointer % 4 = 0, 1, 2
        d_buf = [cuda.zeros(H * W + 8, dtype=cuda.float32, device="cuda") for _ in frames]
        m_buf = [cuda.zeros(H * W + 8, dtype=cuda.uint8, device="cuda") for _ in frames]
        for (_, depth, mask), db, mb in zip(frames, d_buf, m_buf):
            db[d_off:d_off + H * W].copy_(cuda.from_numpy(depth.ravel()))
            mb[m_off:m_off + H * W].copy_(cuda.from_numpy(mask.ravel()))
        cuda.cuda.synchronize()
        d_ptr = [db.data_ptr() + 4 * d_off for db in d_buf]
        m_ptr = [mb.data_ptr() + m_off for mb in m_buf]
        assert d_ptr[0] % 16 == (4 * d_off) % 16 and m_ptr[0] % 4 == m_off % 4
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(8)
            vol.shortcut_stats(True)
            vol.integrate_frames_device(d_ptr, np.stack([p for p, _, _ in frames]), m_ptr)
            stats[(d_off, m_off)] = vol.shortcut_stats(False)
            t, w = vol.download()
        assert np.array_equal(w, ref_w) and np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)), (d_off, m_off)
    assert stats[(1, 1)] == stats[(0, 0)] == stats[(3, 2)] and stats[(0, 0)][1] > 0 and stats[(0, 0)][2] > 0, stats


@pytest.mark.parametrize("nz,tiles", [(40, "8-pixel tiles (10.5 M voxels)"), (36, "16-pixel tiles (9.4 M voxels)"),
                                      (256, "8-pixel tables + 4-pixel fine tables (67 M voxels)")])
def test_single_pixels_at_the_edges_of_projected_boxes(cuda, oracle, nz, tiles):
    """The pixel box of a brick's projected corners is widened by px_margin = 0.5625 + the projection error (csrc/tsdf_capi.hip):
    exactly what it takes to hold the rounded pixel of every voxel of the brick.  A margin that is too small shows where ONE
    pixel differs from its neighbours: the image is a far plane (every brick free space) with single pixels ("needles") whose
    depth lies inside the volume, on a lattice of 37 x 29 pixels -- coprime with both tile sizes, so needles sit at every
    position relative to the tiles -- and the camera moves in steps of 0.13 pixel, with a slight roll, so that needles
    cross the edges of the projected boxes of all bricks along their rays.  A brick claimed "free" although one of its
    voxels rounds onto a needle gets dist = 1 where the reference writes a band value.  Every voxel of the 512 x 512 x nz
    slab against the reference's own kernel (whole_volume.py); every table the library uses (8-pixel tiles from 10 M voxels, the
    fine 4-pixel tables for brick-sized boxes from 64 M; 37 and 29 are coprime with 4 as well)."""
    import whole_volume as wv
    if not wv.available():
        pytest.skip("oracle/_ref/libtsdf_ref_hip.so not built")
    dims, vs = (512, 512, nz), 0.002
    origin = np.array([-0.512, -0.512, 1.0], np.float32)
    cfg = capi.make_config(dims, vs, origin)
    near = 1.0
    px = near / float(synth.TUM_K[0])
    depth0 = np.full((H, W), 3.0, np.float32)
    needle = np.zeros((H, W), bool)
    needle[3::29, 5::37] = True
    frames = []
    for k in range(24):
        d = depth0.copy()
        d[needle] = 1.0 + vs * (3 + (k * 7) % (nz - 6))          # inside the slab, another depth every frame
        pose = synth.make_pose(synth.rot_z(0.002 * (k % 5)), [0.13 * k * px, -0.13 * (k % 9) * px, 0.0])
        frames.append((pose, d))
    dev = [cuda.from_numpy(d).cuda() for _, d in frames]
    poses = np.stack([p for p, _ in frames])
    ref_t, ref_w = wv.replay(cuda, f"needles{nz}", cfg.cam_K, dims, cfg.origin, cfg.voxel_size, cfg.trunc_margin, poses, dev)
    band = int(((ref_t != 1.0) & (ref_w > 0)).sum())
    assert band > 20000, f"the needles should leave band values behind ({band})"
    for variant in (8, 0):
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(variant)
            vol.integrate_frames_device([d.data_ptr() for d in dev], poses)
            info = vol.classification_info()
            wv.assert_volume_equals_reference(cuda, f"needles, {tiles}, variant {variant}", vol, ref_t, ref_w, dims)
        assert info[0] > 0.25, f"a good share of the wavefront-frames should have been claimed ({info[0]})"
    wv.drop(f"needles{nz}")


@pytest.mark.parametrize("nz", [40, 256])
@pytest.mark.parametrize("edge", ["left", "right", "top", "bottom"])
def test_image_border_with_fine_tiles(cuda, oracle, edge, nz):
    """test_patches_touching_the_image_border on a slab large enough for 8-pixel depth tiles (512 x 512 x 40 voxels) and on one
    large enough for the 4-pixel fine tables beside them (512 x 512 x 256): one face
    of the volume projects onto an image border, the camera slides in quarter-pixel steps so that the projected boxes of the
    bricks along that face cross 0 / W - 1 / H - 1 and the +- px_margin band around them; alternately a far plane (free-space
    claims need the box INSIDE the image) and a near plane (skip claims), with a column / row of invalid pixels hugging the
    border every third frame.  Every voxel against the reference's own kernel."""
    import whole_volume as wv
    if not wv.available():
        pytest.skip("oracle/_ref/libtsdf_ref_hip.so not built")
    dims, vs = (512, 512, nz), 0.002
    origin = np.array([-0.512, -0.512, 1.0], np.float32)
    near, far = 1.0, 1.0 + dims[2] * vs
    fx, fy = 535.4, 539.2
    x0, x1 = float(origin[0]), float(origin[0]) + dims[0] * vs
    y0, y1 = float(origin[1]), float(origin[1]) + dims[1] * vs
    K = np.array(synth.TUM_K, np.float32)
    if edge == "left":
        K[2] = -fx * x0 / near
    elif edge == "right":
        K[2] = (W - 1) - fx * x1 / near
    elif edge == "top":
        K[5] = -fy * y0 / near
    else:
        K[5] = (H - 1) - fy * y1 / near
    cfg = capi.make_config(dims, vs, origin, K=K)
    px = near / fx
    poses, depths = [], []
    for k in range(-10, 11):
        shift = [0.25 * k * px, 0.0, 0.0] if edge in ("left", "right") else [0.0, 0.25 * k * px, 0.0]
        d = np.full((H, W), far + 0.5 if k % 2 == 0 else near - 0.2, np.float32)
        if k % 3 == 0:                     # invalid pixels along the border the volume touches
            if edge == "left": d[:, :2] = 0.0
            elif edge == "right": d[:, -2:] = 0.0
            elif edge == "top": d[:2, :] = 0.0
            else: d[-2:, :] = 0.0
        poses.append(synth.make_pose(np.eye(3), shift))
        depths.append(d)
    poses = np.stack(poses)
    dev = [cuda.from_numpy(d).cuda() for d in depths]
    ref_t, ref_w = wv.replay(cuda, f"border_{edge}{nz}", cfg.cam_K, dims, cfg.origin, cfg.voxel_size, cfg.trunc_margin, poses, dev)
    assert 0 < float((ref_w > 0).sum()) < ref_w.numel(), "the border should cut the volume"
    for variant in (8, 0):
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(variant)
            vol.integrate_frames_device([d.data_ptr() for d in dev], poses)
            wv.assert_volume_equals_reference(cuda, f"image border {edge}, nz {nz}, variant {variant}", vol, ref_t, ref_w, dims)
    wv.drop(f"border_{edge}{nz}")
