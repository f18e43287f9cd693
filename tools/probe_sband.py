#!/usr/bin/env python3
"""S-band (every voxel in the truncation band: all 16 B per voxel move) through several kernel variants on one
GPU: ms per frame, GB/s of the 16 B/voxel model, beside the bare stream probes.

    python tools/probe_sband.py [--grid 512] [--variants 3,17,23,7,0]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--variants", default="3,17,23,19,7,0")
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--workload", default="sband", choices=["sband", "sfull"])
    args = ap.parse_args()
    import torch
    from semantic_slam_amd import capi, synth
    D = args.grid
    vs = {512: 0.005, 1024: 0.002}.get(D, 2.56 / D)
    dims = (D, D, D)
    if args.workload == "sband":
        origin, trunc, pose = synth.sband_volume(dims, vs), synth.SBAND_TRUNC, synth.sband_pose
    else:
        origin, trunc, pose = synth.sfull_volume(dims, vs), None, synth.sfull_pose
    depth = torch.from_numpy(synth.sfull_depth()).cuda()
    poses = np.stack([pose(k) for k in range(64)])
    blk = lambda s, n: np.stack([poses[(s + i) % 64] for i in range(n)])
    n = D ** 3
    with capi.Volume(capi.make_config(dims, vs, origin, trunc=trunc)) as vol:
        for nt in (0, 1):
            vol.probe_stream(bool(nt), 3)
            ms = vol.probe_stream(bool(nt), 20)
            print(f"stream_rmw nt={nt}: {ms:.4f} ms  {16.0 * n / ms / 1e6:.0f} GB/s", flush=True)
        for v in [int(x) for x in args.variants.split(",")]:
            vol.set_kernel_variant(v)
            vol.reset()
            vol.integrate_sequence_timed(depth.data_ptr(), blk(0, 64))
            best = 1e9
            for rep in range(3):
                ms = vol.integrate_sequence_timed(depth.data_ptr(), blk(64, args.frames)) / args.frames
                best = min(best, ms)
            print(f"variant {v:3d}: {best:.4f} ms/frame  {n / best / 1e3:.0f} Mvox/s  {16.0 * n / best / 1e6:.0f} GB/s(16B model)  "
                  f"frames/launch {vol.frames_per_launch}", flush=True)


if __name__ == "__main__":
    main()
