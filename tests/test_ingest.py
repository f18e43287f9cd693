"""Keyframe pose / association ingest and BASELINE config[0], the CPU plumbing case."""
import os

import numpy as np

from semantic_slam_amd import ingest, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fr3_office_keyframes.npz")


def test_bundle_roundtrip(tmp_path):
    """Write a file in the layout of ref: src/ORB_SLAM2/System.cc:898-913 and read it back."""
    rng = np.random.default_rng(0)
    poses = [synth.random_pose(rng, 0.5, 1.0).reshape(4, 4) for _ in range(5)]
    p = tmp_path / "bundle.txt"
    with open(p, "w") as f:
        f.write(f"{len(poses)} 2\n")
        for T in poses:
            f.write("0 0 0\n")
            for r in range(3):
                f.write(" ".join(f"{T[r, c]:.6f}" for c in range(3)) + "\n")
            f.write(" ".join(f"{T[r, 3]:.6f}" for r in range(3)) + "\n")
        f.write("0.1 0.2 0.3\n255 255 255\n0\n")          # map points follow; ignored
    got = ingest.load_bundle_poses(str(p))
    assert got.shape == (5, 4, 4)
    assert np.allclose(got, np.stack(poses), atol=1e-6)
    inv = ingest.pose_inverse(got)
    for T, Ti in zip(got, inv):
        assert np.allclose(T.astype(np.float64) @ Ti, np.eye(4), atol=1e-5)
    a = tmp_path / "assoc.txt"
    a.write_text("1.5 rgb/1.5.png 1.5 depth/1.6.png\n2.5 rgb/2.5.png 2.5 depth/2.6.png\n")
    assert ingest.load_associations(str(a)) == [("1.5", "rgb/1.5.png", "depth/1.6.png"), ("2.5", "rgb/2.5.png", "depth/2.6.png")]


def test_fr3_office_fixture():
    z = np.load(GOLD, allow_pickle=False)
    Tcw = z["Tcw"]
    assert Tcw.shape == (194, 4, 4) and len(z["depth_names"]) == 194
    assert np.allclose(Tcw[0], np.eye(4), atol=1e-5)            # first keyframe is the world frame
    for T in Tcw:
        R = T[:3, :3].astype(np.float64)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-4) and abs(np.linalg.det(R) - 1) < 1e-4
    assert str(z["depth_names"][0]).startswith("depth/") and str(z["depth_names"][0]).endswith(".png")
    span = np.ptp(ingest.pose_inverse(Tcw)[:, :3, 3], axis=0)
    assert 0.5 < span.max() < 10.0                               # a desk-scale trajectory, metres


def test_config0_cpu_plumbing(oracle):
    """BASELINE configs[0]: one 640x480 frame of the fr3_office trajectory into 256^3 @ 1 cm on the
    CPU, with the caller-side steps of the reference in order: keyframe pose from bundle.txt
    (Twc = inverse of the stored Tcw), depth / 5000 with the labeller's 3x4 subsampling, per-
    instance mask, origin from the masked depth (ref: src/Object.cpp:23-49), then Integrate.
    (The TUM images are not in the reference tree: the frame is rendered, see synth.SurfScene.)"""
    Twc = ingest.pose_inverse(np.load(GOLD)["Tcw"])
    base2world = Twc[0].ravel()
    dims, vs = (256, 256, 256), 0.01
    scene = synth.SurfScene(dims, vs, np.array([-1.28, -1.28, 0.8], np.float32))
    cam2base = oracle.cam2base(base2world, Twc[0].ravel())
    raw = np.round(np.clip(scene.depth(cam2base), 0, 13.0) * 5000.0).astype(np.uint16)
    depth = oracle.depth_prep(raw, 5000.0)
    mask = np.full((480, 640), 255, np.uint8)
    masked = oracle.mask_depth(depth, mask)
    origin = oracle.object_origin(masked, synth.TUM_K)
    assert origin[2] > 0.5 and origin[0] < 0 and origin[1] < 0
    t, w = oracle.init_grid(dims)
    n = oracle.integrate(synth.TUM_K, cam2base, masked, dims, origin, vs, 0.05, t, w)
    assert n > 10000 and w.max() == 1.0
    assert np.count_nonzero(t < 1.0) > 1000 and t.min() >= -1.0 and t.max() <= 1.0
