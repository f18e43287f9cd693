#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py

Every expected output here is produced by the reference's own GpuIntegrate body
(/root/reference/src/tsdf.cu:15-60) compiled for the host by `make -C oracle ref`
(oracle/_ref/libtsdf_ref.so) -- not by this project's restatement.  The fixtures are data only
(inputs + expected TSDF/weight arrays); no reference source text is stored.  Inputs are built
with semantic_slam_amd.synth and numpy's seeded generator, so the script is reproducible.

The reference has no golden vectors of its own for this path; these files
are what pins the oracle (tests/test_oracle_golden.py) and, through it, the HIP kernel.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.oracle import Ref, build  # noqa: E402
from semantic_slam_amd import synth  # noqa: E402


def small_camera(h, w):
    """TUM fr3 intrinsics scaled to an h x w image (keeps fixtures small)."""
    s = w / 640.0
    return np.array([535.4 * s, 0, 320.1 * s, 0, 539.2 * s, 247.6 * s, 0, 0, 1], np.float32)


def run_case(ref, name, dims, vs, origin, K, frames, trunc=None, depth_u16=False):
    """frames: list of (cam2base[16], depth[h,w] float32 or uint16 raw)."""
    trunc = np.float32(vs) * np.float32(5) if trunc is None else np.float32(trunc)
    n = dims[0] * dims[1] * dims[2]
    t = np.ones(n, np.float32)
    w = np.zeros(n, np.float32)
    poses, depths = [], []
    for c2b, d in frames:
        d32 = (d.astype(np.float32) * np.float32(1.0 / 5000.0)).astype(np.float32) if depth_u16 else d
        ref.integrate(K, c2b, d32, dims, origin, float(vs), float(trunc), t, w, threads=1)
        poses.append(np.asarray(c2b, np.float32))
        depths.append(d)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, dims=np.array(dims, np.int32), voxel_size=np.float32(vs),
                        trunc=np.float32(trunc), origin=np.asarray(origin, np.float32), K=K,
                        cam2base=np.stack(poses), depth=np.stack(depths),
                        depth_is_u16=np.bool_(depth_u16), tsdf=t, weight=w)
    print(f"{name}: dims {dims} frames {len(frames)} updated-voxel-frames {int(w.sum())} "
          f"tsdf!=1: {int((t != 1).sum())}  {os.path.getsize(path) / 1024:.0f} KiB")
    assert w.sum() > 0 or name.endswith("nothing")


def main():
    build(ref=True)
    assert Ref.available(), "oracle/_ref not built (needs /root/reference)"
    ref = Ref()
    rng = np.random.default_rng(20261004)
    h, w = 60, 80
    K = small_camera(h, w)

    # g1: identity pose, sphere + wall, one frame
    dims, vs = (32, 32, 32), 0.02
    origin = synth.surf_volume(32, vs, 0.6)
    sc = synth.SurfScene(dims, vs, origin, K=K, h=h, w=w)
    run_case(ref, "g1_identity", dims, vs, origin, K, [(synth.identity_pose(), sc.depth(synth.identity_pose()))])

    # g2: rotated poses, 4 frames accumulate (weights up to 4, TSDF means of different dists)
    dims, vs = (40, 36, 28), 0.015
    origin = synth.surf_volume(40, vs, 0.5)
    sc = synth.SurfScene(dims, vs, origin, K=K, h=h, w=w)
    frames = []
    for k in range(4):
        p = synth.random_pose(rng, 0.3, 0.25)
        frames.append((p, sc.depth(p)))
    run_case(ref, "g2_rotated_multiframe", dims, vs, origin, K, frames)

    # g3: camera partly behind / beside the volume: pcz <= 0 and out-of-image branches
    dims, vs = (36, 36, 36), 0.03
    origin = np.array([-0.54, -0.54, -0.3], np.float32)  # volume straddles the camera plane
    sc = synth.SurfScene(dims, vs, origin, K=K, h=h, w=w)
    frames = []
    for ang, sh in ((0.0, (0, 0, 0)), (0.9, (0.3, 0, 0.1)), (-1.2, (-0.2, 0.1, 0.2))):
        p = synth.make_pose(synth.rot_y(ang), sh)
        d = np.full((h, w), 0.8, np.float32)
        frames.append((p, d))
    run_case(ref, "g3_behind_and_outside", dims, vs, origin, K, frames)

    # g4: depth with zeros, negatives, > 6 m, exactly 6 m, and step edges
    dims, vs = (48, 24, 24), 0.05
    origin = np.array([-1.2, -0.6, 4.6], np.float32)  # spans z 4.6 .. 5.8 m: near the 6 m cut-off
    d = np.full((h, w), 5.5, np.float32)
    d[:, 20:30] = 0.0
    d[:, 30:36] = -1.0
    d[:, 36:44] = 6.0
    d[:, 44:50] = 6.0000005
    d[:, 50:56] = 7.0
    d[10:20, :] = 4.9
    d[20:25, :] = 5.2
    frames = [(synth.identity_pose(), d), (synth.make_pose(synth.rot_z(0.1), (0.05, -0.02, 0.0)), d)]
    run_case(ref, "g4_invalid_depth_and_edges", dims, vs, origin, K, frames)

    # g5: the reference's compile-time defaults (row length 200, voxel 4 mm, trunc 20 mm, TUM K,
    #     640x480) on a thin slice of the grid, TUM-style uint16 depth / 5000
    dims, vs = (200, 12, 10), 0.004
    origin = np.array([-0.4, -0.02, 0.9], np.float32)
    sc = synth.SurfScene((200, 200, 200), vs, np.array([-0.4, -0.4, 0.6], np.float32))
    frames = []
    for k in range(2):
        p = sc.pose(3 * k, n=16)
        raw = np.round(np.clip(sc.depth(p), 0, 13.0) * 5000.0).astype(np.uint16)
        frames.append((p, raw))
    run_case(ref, "g5_reference_defaults_u16", dims, vs, origin, synth.TUM_K, frames, depth_u16=True)

    # g6: nothing to integrate (all depth invalid): grid must stay at TSDF 1 / weight 0
    dims, vs = (16, 16, 16), 0.02
    origin = synth.surf_volume(16, vs, 0.5)
    run_case(ref, "g6_nothing", dims, vs, origin, K, [(synth.identity_pose(), np.zeros((h, w), np.float32))])

    # g7: full coverage in miniature (S-full shape): every voxel updated every frame
    dims, vs = (32, 32, 32), 0.005
    origin = np.array([-0.08, -0.08, 3.2], np.float32)
    frames = [(synth.sfull_pose(k), np.full((h, w), 5.9, np.float32)) for k in range(3)]
    run_case(ref, "g7_full_coverage", dims, vs, origin, K, frames)


if __name__ == "__main__":
    main()
