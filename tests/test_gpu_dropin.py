"""Drop-in proof: a C++ program written against include/tsdf.hpp exactly as the reference's
Object.cpp uses its TSDF member (tests/dropin_callsite.cpp) produces, on the GPU, the files the
reference's destructor would write -- compared byte for byte with the oracle's writers."""
import os
import struct
import subprocess

import numpy as np
import pytest

from semantic_slam_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "semantic_slam_amd")


def build_harness(tmp_path):
    exe = str(tmp_path / "dropin_callsite")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "dropin_callsite.cpp"), "-o", exe,
                           "-L", PKG, "-ltsdf_dropin", "-ltsdf_hip", f"-Wl,-rpath,{PKG}"])
    return exe


def test_object_callsite_sequence(cuda, oracle, tmp_path):
    assert os.path.isfile(os.path.join(PKG, "libtsdf_dropin.so")), "libtsdf_dropin.so not built"
    exe = build_harness(tmp_path)
    rng = np.random.default_rng(11)
    vol_id = 7
    dims, vs = (200, 200, 200), 0.004              # the reference's compile-time grid
    trunc = float(np.float32(vs) * np.float32(5))
    # a scene sized for that grid, seen by the first keyframe (= base frame, ref: src/Object.cpp:23-29)
    grid_origin = np.array([-0.4, -0.4, 0.7], np.float32)
    scene = synth.SurfScene(dims, vs, grid_origin)
    base2world = synth.random_pose(rng, 0.2, 0.5)   # Twc of the first keyframe
    frames = []
    for k in range(3):
        rel = scene.pose(2 * k, n=16)               # camera pose in the base frame
        cam2world = oracle.multiply(base2world, rel)
        # depth as the offline labeller prepares it (ref: examples/label_instance_rgbd.cpp:89-100),
        # then masked per instance (ref: src/Engine.cpp:192-193)
        raw = np.round(np.clip(scene.depth(rel), 0, 13.0) * 5000.0).astype(np.uint16)
        depth = oracle.depth_prep(raw)
        mask = np.zeros((480, 640), np.uint8)
        mask[40:440, 60:600] = 255
        frames.append((cam2world, oracle.mask_depth(depth, mask)))
    # origin exactly as Object::Object derives it from the first masked depth (ref: src/Object.cpp:37-49)
    origin = oracle.object_origin(frames[0][1], synth.TUM_K)

    inp = tmp_path / "frames.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<ii", vol_id, len(frames)))
        f.write(base2world.astype(np.float32).tobytes())
        f.write(origin.astype(np.float32).tobytes())
        for c2w, d in frames:
            f.write(c2w.astype(np.float32).tobytes())
            f.write(d.astype(np.float32).tobytes())
    subprocess.check_call([exe, str(inp)], cwd=str(tmp_path))

    ref_t, ref_w = oracle.init_grid(dims)
    for c2w, d in frames:
        c2b = oracle.cam2base(base2world, c2w)
        oracle.integrate(synth.TUM_K, c2b, d, dims, origin, vs, trunc, ref_t, ref_w)
    assert ref_w.sum() > 1000, "scene must update voxels"
    oracle.save_ply(str(tmp_path / "want.ply"), ref_t, ref_w, dims, vs, origin)
    oracle.save_bin(str(tmp_path / "want.bin"), ref_t, dims, origin, vs, trunc)
    assert (tmp_path / f"tsdf{vol_id}.bin").read_bytes() == (tmp_path / "want.bin").read_bytes()
    assert (tmp_path / f"tsdf{vol_id}.ply").read_bytes() == (tmp_path / "want.ply").read_bytes()


def test_tsdffusion_native_backend(cuda, oracle, tmp_path):
    """class TSDFfusion over the native library: the volume of the reference's Python glue
    ([0,10]^3 m at 0.02 m = 500^3, world frame, TUM intrinsics; ref: src/TSDFfusion.py.in:19-29) fed
    through the C++ class, point cloud compared with the oracle applying the reference's GpuIntegrate
    rule to the same volume.  (The third-party package's own arithmetic is absent: parity unpinned.)"""
    exe = str(tmp_path / "dropin_tsdffusion")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "dropin_tsdffusion.cpp"), "-o", exe,
                           "-L", PKG, "-ltsdf_dropin", "-ltsdf_hip", f"-Wl,-rpath,{PKG}"])
    dims, vs = (500, 500, 500), 0.02
    origin = np.zeros(3, np.float32)
    # a camera inside the [0,10]^3 room looking along +z at a sphere + wall scene placed in front of it
    scene = synth.SurfScene((200, 200, 200), 0.02, np.array([3.0, 3.0, 3.0], np.float32))
    frames = []
    for k in range(2):
        pose = synth.make_pose(synth.rot_y(0.05 * k), [5.0 + 0.1 * k, 5.0, 0.5])
        frames.append((pose, scene.depth(pose, quantize=True)))
    inp = tmp_path / "frames.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for pose, d in frames:
            f.write(pose.astype(np.float32).tobytes())
            f.write(d.astype(np.float32).tobytes())
    out = tmp_path / "cloud.ply"
    mesh = tmp_path / "mesh.ply"
    subprocess.check_call([exe, str(inp), str(out), str(mesh)], cwd=str(tmp_path))
    t, w = oracle.init_grid(dims)
    for pose, d in frames:
        oracle.integrate(synth.TUM_K, pose, d, dims, origin, vs, float(np.float32(vs) * np.float32(5)), t, w)
    assert w.sum() > 10000
    oracle.save_ply(str(tmp_path / "want.ply"), t, w, dims, vs, origin)
    assert out.read_bytes() == (tmp_path / "want.ply").read_bytes()
    tri = oracle.mesh_triangles(t, w, dims[:2], 0, dims[2], vs, origin)
    raw = mesh.read_bytes()
    body = raw[raw.index(b"end_header\n") + len(b"end_header\n"):]
    assert len(tri) > 1000 and body[:36 * len(tri)] == tri.tobytes()


def test_python_mirror_of_class_tsdf(cuda, oracle, tmp_path):
    """semantic_slam_amd.tsdf.TSDF: same names and behaviour as the C++ class (default grid, files on close)."""
    from semantic_slam_amd.tsdf import TSDF
    dims, vs = (200, 200, 200), 0.004
    origin = np.array([-0.4, -0.4, 0.7], np.float32)
    scene = synth.SurfScene(dims, vs, origin)
    base = synth.identity_pose()
    t = TSDF(480, 640, 5, base, origin)
    assert t.voxel_grid_TSDF[0] == 1.0 and t.voxel_grid_weight.sum() == 0
    ref_t, ref_w = oracle.init_grid(dims)
    for k in range(2):
        c2w = scene.pose(k, n=8)
        d = scene.depth(c2w)
        t.Integrate(d, c2w)
        oracle.integrate(synth.TUM_K, c2w, d, dims, origin, vs, float(np.float32(vs) * np.float32(5)), ref_t, ref_w)
    t.Download()
    assert np.array_equal(t.voxel_grid_weight, ref_w) and np.array_equal(t.voxel_grid_TSDF, ref_t)
    t.close(str(tmp_path))
    oracle.save_bin(str(tmp_path / "want.bin"), ref_t, dims, origin, vs, float(np.float32(vs) * np.float32(5)))
    assert (tmp_path / "tsdf5.bin").read_bytes() == (tmp_path / "want.bin").read_bytes()
    assert (tmp_path / "tsdf5.ply").stat().st_size > 100
