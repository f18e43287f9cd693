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
