"""Output formats (ref: src/tsdf.cu:114-132 .bin, :170-218 .ply) and the caller-side adapters
(ref: src/Object.cpp:37-49 origin, examples/label_instance_rgbd.cpp:89-100 depth prep,
src/Engine.cpp:192-193 masking) of the CPU oracle."""
import struct

import numpy as np
import pytest

from oracle.oracle import RefHost
from semantic_slam_amd import synth


def small_volume(oracle):
    dims, vs = (20, 16, 12), 0.02
    origin = synth.surf_volume(20, vs, 0.5)
    K = np.array([66.9, 0, 40.0, 0, 67.4, 30.9, 0, 0, 1], np.float32)
    sc = synth.SurfScene(dims, vs, origin, K=K, h=60, w=80)
    t, w = oracle.init_grid(dims)
    for k in range(2):
        p = sc.pose(k, n=4)
        oracle.integrate(K, p, sc.depth(p), dims, origin, vs, 0.1, t, w)
    return dims, vs, origin, t, w


def test_bin_format(oracle, tmp_path):
    dims, vs, origin, t, w = small_volume(oracle)
    p = tmp_path / "v.bin"
    oracle.save_bin(str(p), t, dims, origin, vs, 0.1)
    raw = p.read_bytes()
    assert len(raw) == 4 * (8 + t.size)
    hdr = struct.unpack("<8f", raw[:32])
    assert hdr[:3] == tuple(float(d) for d in dims)             # dims stored as floats
    assert np.array_equal(np.array(hdr[3:6], np.float32), origin)
    assert np.float32(hdr[6]) == np.float32(vs) and np.float32(hdr[7]) == np.float32(0.1)
    assert np.array_equal(np.frombuffer(raw[32:], np.float32), t)  # TSDF only, weights are not saved


def test_ply_format_and_surface_rule(oracle, tmp_path):
    dims, vs, origin, t, w = small_volume(oracle)
    t = t.copy()
    hit = np.flatnonzero(w > 0)
    t[hit[::5]] = 0.0                                           # |tsdf| == 0 voxels are dropped
    pts = oracle.surface_points(t, w, dims, vs, origin)
    keep = (np.abs(t) != 0) & (w > 0.9)                         # ref: src/tsdf.cu:179
    assert len(pts) == int(keep.sum()) and 0 < len(pts) < t.size
    idx = np.flatnonzero(keep)
    z, rem = np.divmod(idx, dims[0] * dims[1])
    y, x = np.divmod(rem, dims[0])
    want = np.stack([origin[0] + x.astype(np.float32) * np.float32(vs),
                     origin[1] + y.astype(np.float32) * np.float32(vs),
                     origin[2] + z.astype(np.float32) * np.float32(vs)], 1).astype(np.float32)
    assert np.array_equal(pts, want)                            # grid order, origin + index*size
    p = tmp_path / "v.ply"
    oracle.save_ply(str(p), t, w, dims, vs, origin)
    raw = p.read_bytes()
    head = (f"ply\nformat binary_little_endian 1.0\nelement vertex {len(pts)}\n"
            "property float x\nproperty float y\nproperty float z\nend_header\n").encode()
    assert raw.startswith(head) and raw[len(head):] == pts.tobytes()


@pytest.mark.skipif(not RefHost.available(), reason="oracle/_ref/libtsdf_ref_host.so not built (make -C oracle ref_host)")
def test_ply_equals_the_references_own_writer_byte_for_byte(oracle, tmp_path):
    """The oracle's writer against TSDF::SaveVoxelGrid2SurfacePointCloud itself (ref: src/tsdf.cu:170-218, compiled as it
    stands: oracle/ref_host_driver.cpp), called as ~TSDF calls it (ref: src/tsdf.cu:110-112): the surface rule, the point
    order, the coordinates' arithmetic and the header, byte for byte -- on an integrated volume and on arrays that hold the
    rule's edge cases (TSDF +0 / -0 / NaN / denormal, weights around the 0.9 threshold, NaN weights)."""
    ref = RefHost()
    rng = np.random.default_rng(170)
    cases = [small_volume(oracle)]
    for dims in ((1, 1, 1), (7, 5, 3), (32, 9, 4), (3, 40, 2)):
        n = dims[0] * dims[1] * dims[2]
        t = rng.choice(np.array([0.0, -0.0, 1.0, -1.0, 0.25, np.nan, 1e-42, -1e-42, 0.999], np.float32), n).astype(np.float32)
        w = rng.choice(np.array([0.0, 0.9, np.nextafter(np.float32(0.9), np.float32(1)), np.nextafter(np.float32(0.9), np.float32(0)), 1.0, 37.0,
                                 np.nan, -1.0], np.float32), n).astype(np.float32)
        cases.append((dims, 0.013, np.array([-0.31, 0.2, 1.7], np.float32), t, w))
    cases.append(((4, 4, 4), 0.02, np.zeros(3, np.float32), np.ones(64, np.float32), np.zeros(64, np.float32)))   # no point at all
    for k, (dims, vs, origin, t, w) in enumerate(cases):
        a, b = tmp_path / f"oracle{k}.ply", tmp_path / f"ref{k}.ply"
        oracle.save_ply(str(a), t, w, dims, vs, origin)
        ref.save_ply(str(b), t, w, dims, vs, origin)
        assert a.read_bytes() == b.read_bytes(), f"case {k}: dims {dims}"
        assert len(oracle.surface_points(t, w, dims, vs, origin)) * 12 + len(b.read_bytes().split(b"end_header\n")[0]) + 11 == len(b.read_bytes())


def test_object_origin_rule(oracle):
    depth = np.zeros((480, 640), np.float32)
    depth[100, 50], depth[400, 600], depth[240, 320] = 2.0, 1.5, 0.75
    depth[5, 5] = -3.0                                          # z <= 0 is skipped
    o = oracle.object_origin(depth, synth.TUM_K)
    f = np.float32
    xs = [(f(c) - f(320.1)) * f(z) * (f(1) / f(535.4)) for c, z in ((50, 2.0), (600, 1.5), (320, 0.75))]
    ys = [(f(r) - f(247.6)) * f(z) * (f(1) / f(539.2)) for r, z in ((100, 2.0), (400, 1.5), (240, 0.75))]
    assert o[0] == min(xs) and o[1] == min(ys) and o[2] == f(0.75)
    assert np.array_equal(oracle.object_origin(np.zeros((4, 4), np.float32), synth.TUM_K), [1000, 1000, 1000])


def test_depth_prep_and_mask(oracle):
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 30000, (480, 640)).astype(np.uint16)
    d = oracle.depth_prep(raw)
    keep = np.zeros((480, 640), bool)
    keep[::4, ::3] = True                                       # rows 0,4,8.. and cols 0,3,6..
    assert np.all(d[~keep] == 0)
    assert np.array_equal(d[keep], raw[keep].astype(np.float32) * (np.float32(1.0) / np.float32(5000.0)))
    mask = np.zeros((480, 640), np.uint8)
    mask[100:200, 100:300] = 255
    m = oracle.mask_depth(d, mask)
    assert np.array_equal(m[100:200, 100:300], d[100:200, 100:300]) and m.sum() == d[100:200, 100:300].sum()
