"""Ingest of the SLAM front-end's saved map, as far as the TSDF path needs it.

The reference's ORB_SLAM2 runner saves keyframe poses in bundler format and the names of the
RGB/depth images of each keyframe (ref: src/ORB_SLAM2/System.cc:884-945 `SaveMap`, :981-1002
`SaveAssociations`); the offline labeller reloads them (ref: src/Utility.cpp:106-289,
examples/label_instance_rgbd.cpp:271-399) and hands `KeyFrame::GetPoseInverse()` (camera-to-
world) to the TSDF.  This module reads the same two files into arrays.

bundle.txt layout (ref: src/ORB_SLAM2/System.cc:898-913):
    <#keyframes> <#points>
    per keyframe:  "f k1 k2"          (zeros)
                   3 rows of Rcw      (world -> camera rotation)
                   tcw                (world -> camera translation)
    then the map points (not needed here).
"""
import numpy as np


def load_bundle_poses(path):
    """World-to-camera poses Tcw of all keyframes, float32 [n, 4, 4] (ref: src/Utility.cpp:137-172)."""
    with open(path) as f:
        n_kf, _n_pts = (int(x) for x in f.readline().split()[:2])
        out = np.zeros((n_kf, 4, 4), np.float32)
        for i in range(n_kf):
            f.readline()                                   # intrinsics line
            rows = [[float(x) for x in f.readline().split()] for _ in range(4)]
            out[i, :3, :3] = np.asarray(rows[:3], np.float32)
            out[i, :3, 3] = np.asarray(rows[3], np.float32)
            out[i, 3, 3] = 1.0
    return out


def pose_inverse(Tcw):
    """Camera-to-world Twc = [Rcw^T | -Rcw^T tcw] in fp32, what ORB_SLAM2's KeyFrame::GetPoseInverse
    returns and ref: src/Engine.cpp:222 / src/Object.cpp:23-29 feed to the TSDF.  (ORB_SLAM2 is not
    in the reference tree; the last bit of its OpenCV product is unpinned.)"""
    Tcw = np.asarray(Tcw, np.float32).reshape(-1, 4, 4)
    out = np.zeros_like(Tcw)
    for i, T in enumerate(Tcw):
        Rwc = T[:3, :3].T.copy()
        out[i, :3, :3] = Rwc
        out[i, :3, 3] = -(Rwc @ T[:3, 3]).astype(np.float32)
        out[i, 3, 3] = 1.0
    return out


def load_associations(path):
    """[(timestamp, rgb_name, depth_name)] per keyframe (ref: src/ORB_SLAM2/System.cc:981-1002)."""
    out = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if len(p) >= 4:
                out.append((p[0], p[1], p[3]))
    return out


def load_depth_png(path):
    """The 16-bit depth PNG of one TUM RGB-D frame as uint16 [H, W] -- the payload `tsdf_integrate_u16` takes (metres =
    value / 5000, ref: config/TUM3.yaml:34; the labeller reads the file unchanged, ref:
    examples/label_instance_rgbd.cpp:141-152 `cv::imread(..., CV_LOAD_IMAGE_UNCHANGED)`).  Read with Pillow (OpenCV is
    not in this image); an 8-bit or colour file is refused rather than silently rescaled."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode not in ("I;16", "I;16L", "I;16B", "I"):
            raise ValueError(f"{path}: expected a 16-bit single-channel depth image, got mode {im.mode!r}")
        a = np.array(im)
    if a.ndim != 2:
        raise ValueError(f"{path}: expected one channel, got shape {a.shape}")
    if a.dtype != np.uint16:
        if a.min() < 0 or a.max() > 65535:
            raise ValueError(f"{path}: values outside the 16-bit range")
        a = a.astype(np.uint16)
    return np.ascontiguousarray(a)


def save_depth_png(path, raw_u16):
    """Write a uint16 [H, W] frame as a 16-bit greyscale PNG (what the TUM dataset ships; used by tests and tools)."""
    from PIL import Image
    a = np.ascontiguousarray(raw_u16, dtype=np.uint16)
    Image.fromarray(a).save(path, format="PNG")


def iter_keyframes(bundle_path, associations_path, root):
    """The offline labeller's keyframe loop (ref: examples/label_instance_rgbd.cpp:78-110) as far as the TSDF needs it:
    yields (Twc [16] float32, raw depth uint16 [H, W], depth file name) per keyframe, poses from bundle.txt
    (Twc = inverse of the stored Tcw, as KeyFrame::GetPoseInverse), images named by associations.txt below `root`
    (the TUM sequence directory).  Feed to `tsdf_integrate_u16(raw, 5000, 4, 3, Twc)` for the labeller's own depth
    preparation (every 4th row / 3rd column, / 5000; ref: :89-100), or with steps (1, 1) for the whole frame."""
    import os
    Twc = pose_inverse(load_bundle_poses(bundle_path))
    names = load_associations(associations_path)
    if len(names) != len(Twc):
        raise ValueError(f"{bundle_path} holds {len(Twc)} keyframes, {associations_path} names {len(names)}")
    for T, (_, _rgb, depth_name) in zip(Twc, names):
        yield T.ravel().copy(), load_depth_png(os.path.join(root, depth_name)), depth_name
