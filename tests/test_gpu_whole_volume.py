"""Whole-volume bit parity at the BASELINE sizes against the reference's OWN kernel, on the device.

The reference's GpuIntegrate (ref: src/tsdf.cu:15-60), compiled by hipcc as it stands and launched with its own shape
<<<dim_z, dim_y>>> (oracle/_ref/libtsdf_ref_hip.so, tests/whole_volume.py), replays each workload over the ENTIRE grid;
the product -- every path that ships: one launch per call, fused sequences, fused + brick classification forced on /
forbidden / decided per launch, deferred host frames, z-slab handles, batched per-object volumes -- must reproduce every
TSDF value and every weight bit for bit (compared on the device; the first differing voxel is reported).

  (a) 512^3 @ 5 mm S-surf, 64 frames (BASELINE configs[1] geometry with a real surface)
  (b) 512^3 S-band, 37 frames (the bench headline's workload) and S-full, 35 frames
  (c) 1024^3 @ 2 mm, all 194 fr3_office keyframes (configs[2]); and one rank's z-slab of it, z in [384, 512) -- rank 3
      of 8 of configs[3] -- against the same slices of the full-grid replay (global-z rounding, ref: src/tsdf.cu:29)
  (d) 16 x 200^3 per-object volumes fed depth x instance mask (ref: src/Engine.cpp:192-193; the reference kernel gets the
      product depth * mask/255 pre-multiplied on the host, as its caller prepares it)
  (e) S-surf through sensor noise (sigma 2 mm) and 5 % dropouts
Nothing here reads /root/reference: the .so was built in the build container and travels with the snapshot.
"""
import numpy as np
import pytest

import whole_volume as wv
from semantic_slam_amd import capi, synth

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not wv.available(), reason="oracle/_ref/libtsdf_ref_hip.so not built")]

_W = {}


def _workload(torch, name):
    """(cfg, dims, poses [n,16] = cam2base (base pose = identity), device depth frames, host depth frames)."""
    if name in _W:
        return _W[name]
    D, vs = 512, 0.005
    dims = (D, D, D)
    trunc = None
    if name in ("ssurf", "ssurf_noisy"):
        origin = synth.surf_volume(D, vs, 1.0)
        scene = synth.SurfScene(dims, vs, origin)
        poses = np.stack([scene.pose(k, 64) for k in range(64)])
        if "ssurf" in _W and name == "ssurf_noisy":
            depths = _W["ssurf"][4]
        else:
            depths = [scene.depth(p, quantize=True) for p in poses]
        if name == "ssurf_noisy":
            depths = synth.sensor_imperfections(depths, 2.0, 0.05)
    elif name == "sband":
        origin = synth.sband_volume(D, vs)
        trunc = synth.SBAND_TRUNC
        poses = np.stack([synth.sband_pose(k) for k in range(37)])
        depths = [synth.sfull_depth()]
    elif name == "sfull":
        origin = synth.sfull_volume(D, vs)
        poses = np.stack([synth.sfull_pose(k) for k in range(35)])
        depths = [synth.sfull_depth()]
    else:
        raise KeyError(name)
    cfg = capi.make_config(dims, vs, origin, trunc=trunc)
    dev = [torch.from_numpy(d).cuda() for d in depths]
    if len(dev) == 1:
        dev = dev * len(poses)
        depths = depths * len(poses)
    _W[name] = (cfg, dims, poses, dev, depths)
    return _W[name]


def _reference(torch, name):
    cfg, dims, poses, dev, _ = _workload(torch, name)
    return wv.replay(torch, name + "512", cfg.cam_K, dims, cfg.origin, cfg.voxel_size, cfg.trunc_margin, poses, dev)


PATHS = ["frame", "fused", "fused_bricks", "fused_per_voxel", "host_deferred"] + (["fused_rows", "fused_brick_workgroups"] if capi.experiments_build() else [])


def _run_path(vol, path, poses, dev, host):
    if path == "frame":                       # one kernel launch per call (the bench headline's kernel)
        vol.set_deferral(0)
        for p, d in zip(poses, dev):
            vol.integrate_device(d.data_ptr(), p)
    elif path == "host_deferred":             # TSDF::Integrate's own call: host frames, collected 32 at a time
        for p, d in zip(poses, host):
            vol.integrate(d, p)
    else:                                     # a known sequence: fused, classification per launch / always / never / round-1 rows
        vol.set_kernel_variant({"fused": 0, "fused_bricks": 8, "fused_per_voxel": 7, "fused_rows": 11, "fused_brick_workgroups": 13}[path])
        vol.integrate_frames_device([d.data_ptr() for d in dev], poses)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name", ["ssurf", "ssurf_noisy", "sband", "sfull"])
def test_512_cube_every_voxel_equals_the_reference_kernel(cuda, name, path):
    cfg, dims, poses, dev, host = _workload(cuda, name)
    ref_t, ref_w = _reference(cuda, name)
    if name in ("sband", "sfull"):   # what the workload is for: every voxel updated by every frame
        assert bool((ref_w == float(len(poses))).all())
    else:
        frac = float((ref_w > 0).sum()) / ref_w.numel()
        assert 0.05 < frac < 0.95, f"updated fraction {frac}: the scene should mix updated and skipped voxels"
    with capi.Volume(cfg) as vol:
        _run_path(vol, path, poses, dev, host)
        wv.assert_volume_equals_reference(cuda, f"{name} 512^3 / {path}", vol, ref_t, ref_w, dims)


# ---- (c) configs[2] and one rank's slab of configs[3] ---------------------------------------------------------------
# (the whole 1024^3 grid under the fused paths: tests/test_gpu_ingest.py::test_config2_full_fr3_trajectory_fused)
@pytest.mark.parametrize("path", ["fused", "fused_bricks", "frame", "host_deferred"])
def test_config3_one_ranks_slab_of_the_1024_cube(cuda, oracle, path):
    """BASELINE configs[3] in its own shape on one GPU: the handle of rank 3 of 8 of the 1024^3 grid -- z in [384, 512),
    1024 x 1024 x 128 voxels, z_begin != 0 -- fed the fr3 trajectory; every voxel of the slab against the same slices of
    the reference kernel's FULL-grid replay.  (The kernels take the global z index, so a slab rounds like the whole grid:
    ref src/tsdf.cu:29.)"""
    T = wv.fr3_trajectory(oracle, cuda)
    n = T["n"] if path != "frame" else 40          # one launch per call: the first 40 keyframes
    ref_t, ref_w = wv.fr3_reference(cuda, oracle, n)
    c, dims = T["cfg"], T["dims"]
    cfg = capi.make_config(dims, c.voxel_size, list(c.origin), base2world=list(c.base2world), z_begin=384, z_end=512)
    per = dims[0] * dims[1]
    assert float(ref_w[384 * per:512 * per].max()) > 20, "the slab should be seen by many keyframes"
    with capi.Volume(cfg) as vol:
        _run_path(vol, path, T["poses"][:n], T["dev"][:n], T["depths"][:n])
        assert np.array_equal(vol.last_cam2base(), T["c2b"][n - 1])
        wv.assert_volume_equals_reference(cuda, f"slab [384, 512) of 1024^3 / {path}", vol, ref_t, ref_w, dims)
    if path == "frame":
        wv.drop(f"fr3_1024_{n}")


@pytest.mark.parametrize("rank_,path", [(0, "fused"), (7, "fused_bricks"), (5, "fused")])
def test_config3_other_ranks_slabs_of_the_1024_cube(cuda, oracle, rank_, path):
    """The same for the slabs at the two ends of the grid (rank 0: z_begin = 0, the slices nearest the first keyframe's camera;
    rank 7: the last slices, z_end = dim_z) and one more in the middle: every voxel against the same slices of the reference
    kernel's full-grid replay of all 194 keyframes."""
    T = wv.fr3_trajectory(oracle, cuda)
    n = T["n"]
    ref_t, ref_w = wv.fr3_reference(cuda, oracle, n)
    c, dims = T["cfg"], T["dims"]
    zb, ze = rank_ * dims[2] // 8, (rank_ + 1) * dims[2] // 8
    cfg = capi.make_config(dims, c.voxel_size, list(c.origin), base2world=list(c.base2world), z_begin=zb, z_end=ze)
    with capi.Volume(cfg) as vol:
        _run_path(vol, path, T["poses"][:n], T["dev"][:n], T["depths"][:n])
        wv.assert_volume_equals_reference(cuda, f"slab [{zb}, {ze}) of 1024^3 / {path}", vol, ref_t, ref_w, dims)


# ---- (d) the reference's own usage: one 200^3 volume per object instance ----------------------------------------------
@pytest.mark.parametrize("path", ["batch_deferred", "batch_per_frame", "handles"])
def test_sixteen_masked_object_volumes_every_voxel_equals_the_reference_kernel(cuda, oracle, path):
    rng = np.random.default_rng(1603)
    dims, vs, K = (200, 200, 200), 0.004, synth.TUM_K
    n_obj, n_frames = 16, 35
    scene = synth.SurfScene(dims, vs, np.array([-0.4, -0.4, 0.7], np.float32))
    objs = []
    for i in range(n_obj):
        o = np.array([-0.4 + rng.uniform(-0.3, 0.3), -0.4 + rng.uniform(-0.25, 0.25), 0.7 + rng.uniform(0, 0.6)], np.float32)
        c = o + 0.4
        m = np.zeros((480, 640), np.uint8)
        u0, u1 = K[0] * (c[0] - 0.22) / c[2] + K[2], K[0] * (c[0] + 0.22) / c[2] + K[2]
        v0, v1 = K[4] * (c[1] - 0.22) / c[2] + K[5], K[4] * (c[1] + 0.22) / c[2] + K[5]
        m[max(0, int(v0)):max(0, min(480, int(v1))), max(0, int(u0)):max(0, min(640, int(u1)))] = 255
        base = synth.random_pose(rng, 0.05, 0.05) if i % 2 else synth.identity_pose()   # every object has its own base frame
        objs.append((o, m, base))
    cfgs = [capi.make_config(dims, vs, o, base2world=b, vol_id=i) for i, (o, _, b) in enumerate(objs)]
    poses = [scene.pose(k % 16, 16) for k in range(n_frames)]
    depths = [scene.depth(scene.pose(k, 16), quantize=True) for k in range(16)]
    d_dev = [cuda.from_numpy(d).cuda() for d in depths]
    m_dev = [cuda.from_numpy(m).cuda() for _, m, _ in objs]
    refs = []
    for i, (o, m, b) in enumerate(objs):
        masked = [cuda.from_numpy(oracle.mask_depth(d, m)).cuda() for d in depths]     # ref: src/Engine.cpp:192-193
        c2b = [oracle.cam2base(b, p) for p in poses]
        refs.append(wv.replay(cuda, f"obj{i}", K, dims, o, vs, cfgs[i].trunc_margin, c2b, [masked[k % 16] for k in range(n_frames)]))
        wv.drop(f"obj{i}")
    assert sum(float(rw.sum()) for _, rw in refs) > 1e6 and sum(1 for _, rw in refs if float(rw.sum()) > 1000) >= 8
    if path == "handles":
        for i, cfg in enumerate(cfgs):
            with capi.Volume(cfg) as vol:
                for k in range(n_frames):
                    vol.integrate_masked_device(d_dev[k % 16].data_ptr(), m_dev[i].data_ptr(), poses[k])
                wv.assert_volume_equals_reference(cuda, f"object {i} / own handle", vol, refs[i][0], refs[i][1], dims)
        return
    with capi.Batch(cfgs) as batch:
        if path == "batch_per_frame":
            batch.volumes[0].set_deferral(0)
        for k in range(n_frames):
            batch.integrate_device(d_dev[k % 16].data_ptr(), [m.data_ptr() for m in m_dev], poses[k])
        batch.sync()
        for i, vol in enumerate(batch.volumes):
            wv.assert_volume_equals_reference(cuda, f"object {i} / {path}", vol, refs[i][0], refs[i][1], dims)


# ---- one grid over several slab handles in one process (tsdf_group_*): the C++ host's own multi-device shape ----------
@pytest.mark.parametrize("mode", ["calls", "sequence"])
def test_512_cube_as_a_group_of_slabs_every_voxel_equals_the_reference_kernel(cuda, mode):
    """tsdf_group_create over five slab handles (uneven: 512 = 102 + 102 + 103 + 102 + 103 slices; the box has one card, so
    all on device 0 -- the fan-out, per-slab streams, deferral and global-z indexing are what runs), fed 40 host frames one
    call at a time (the reference's call, collected per slab) or as one sequence: every voxel of every slab against the
    same slices of the reference kernel's whole-grid replay."""
    cfg, dims, poses, dev, host = _workload(cuda, "ssurf")
    n = 40
    ref_t, ref_w = wv.replay(cuda, "ssurf512_40", cfg.cam_K, dims, cfg.origin, cfg.voxel_size, cfg.trunc_margin, poses[:n], dev[:n])
    with capi.Group(cfg, [0, 0, 0, 0, 0]) as grp:
        if mode == "calls":
            for p, d in zip(poses[:n], host[:n]):
                grp.integrate(d, p)
        else:
            grp.integrate_frames(host[:n], poses[:n])
        zs = [(s.cfg.z_begin, s.cfg.z_end) for s in grp.slabs]
        assert zs[0][0] == 0 and zs[-1][1] == dims[2] and all(a[1] == b[0] for a, b in zip(zs, zs[1:])) and len({b - a for a, b in zs}) == 2
        for s in grp.slabs:
            wv.assert_volume_equals_reference(cuda, f"group slab [{s.cfg.z_begin}, {s.cfg.z_end}) / {mode}", s, ref_t, ref_w, dims)
    wv.drop("ssurf512_40")


# ---- 8-pixel depth tiles with an image that is no multiple of anything ---------------------------------------------
@pytest.mark.parametrize("nz", [96, 512])      # 12.6 M voxels: 8-pixel tiles; 67 M: the 4-pixel fine tables beside them
@pytest.mark.parametrize("hw", [(333, 517), (250, 402)])
def test_fine_tiles_with_odd_image_sizes_every_voxel_equals_the_reference_kernel(cuda, oracle, hw, nz):
    """Slabs of 10 M voxels and more classify against 8 x 8-pixel depth tiles.  The other whole-volume tests use 640 x 480
    images (80 x 60 whole tiles); here the image is 517 x 333 / 402 x 250 pixels -- partial tiles on both borders, rows
    that are no multiple of four pixels (the table kernel's scalar path) -- with intrinsics to match, through sensor noise and
    dropouts, 24 frames fused over the brick work list: every voxel of the 512 x 256 x 96 grid against the reference kernel --
    and of a 512 x 256 x 512 grid, large enough for the fine tables (129 x 84 / 101 x 63 tiles of 4 pixels, partial on both
    borders, their level (0, 0) written by the strip kernel's scalar path)."""
    h, w = hw
    dims, vs = (512, 256, nz), 0.004
    K = np.array([0.83 * w, 0, 0.49 * w, 0, 0.84 * w, 0.52 * h, 0, 0, 1], np.float32)
    origin = np.array([-dims[0] * vs / 2, -dims[1] * vs / 2, 1.1], np.float32)
    cfg = capi.make_config(dims, vs, origin, K=K, im_height=h, im_width=w)
    scene = synth.SurfScene(dims, vs, origin, K=K, h=h, w=w)
    poses = np.stack([scene.pose(k, 24) for k in range(24)])
    depths = synth.sensor_imperfections([scene.depth(p, quantize=True) for p in poses], 2.0, 0.05)
    dev = [cuda.from_numpy(d).cuda() for d in depths]
    ref_t, ref_w = wv.replay(cuda, f"odd{h}x{w}x{nz}", cfg.cam_K, dims, cfg.origin, cfg.voxel_size, cfg.trunc_margin, poses, dev)
    frac = float((ref_w > 0).sum()) / ref_w.numel()
    assert 0.05 < frac < 0.98, frac
    for variant in (8, 0):
        with capi.Volume(cfg) as vol:
            vol.set_kernel_variant(variant)
            vol.integrate_frames_device([d.data_ptr() for d in dev], poses)
            info = vol.classification_info()
            wv.assert_volume_equals_reference(cuda, f"{w} x {h} image, nz {nz}, variant {variant}", vol, ref_t, ref_w, dims)
        assert info[0] > 0.3, info
    wv.drop(f"odd{h}x{w}x{nz}")
