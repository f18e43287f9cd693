"""Scratch timing probe (not a test): 512^3 S-full kernel time."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_slam_amd import capi, synth
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vs = 0.005 if D == 512 else 0.002
origin = synth.sfull_volume(D, vs)
cfg = capi.make_config((D, D, D), vs, origin)
vol = capi.Volume(cfg)
d = torch.from_numpy(synth.sfull_depth()).cuda()
poses = np.stack([synth.sfull_pose(k) for k in range(20)])
vol.integrate_sequence_timed(d.data_ptr(), poses[:3])
for rep in range(3):
    ms = vol.integrate_sequence_timed(d.data_ptr(), poses)
    per = ms / len(poses)
    print(f"D={D} {per:.4f} ms/frame  {D**3/per/1e3:.0f} Mvox/s  {16*D**3/per/1e6:.1f} GB/s  frac {16*D**3/per/1e6/8000:.3f}")
