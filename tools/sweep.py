"""Development probe (not a test, not the bench): time every Integrate kernel variant and the
bare RMW stream ceiling on one GPU.   python tools/sweep.py [grid] [workload]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_slam_amd import capi, synth  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
workload = sys.argv[2] if len(sys.argv) > 2 else "sfull"
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 17, 23, 39, 43] + [b + c for b in (80, 96) for c in (3, 7, 11)]
vs = {512: 0.005, 1024: 0.002}.get(D, 2.56 / D)
dims = (D, D, D)
if workload == "sfull":
    origin = synth.sfull_volume(D, vs)
    depth = synth.sfull_depth()
    poses = np.stack([synth.sfull_pose(k) for k in range(50)])
else:
    origin = synth.surf_volume(D, vs, 1.0)
    scene = synth.SurfScene(dims, vs, origin)
    poses = np.stack([scene.pose(k, 64) for k in range(50)])
    depth = scene.depth(poses[0], quantize=True)
cfg = capi.make_config(dims, vs, origin)
vol = capi.Volume(cfg)
d = torch.from_numpy(depth).cuda()
N = D ** 3
for nt in (0, 1):
    vol.probe_stream(nt, 3)
    ms = min(vol.probe_stream(nt, 20) for _ in range(3))
    print(f"stream_rmw nt={nt}: {ms:.4f} ms/pass  {16 * N / ms / 1e6:.0f} GB/s")
res = {}
for rnd in range(3):
    for v in variants:
        vol.set_kernel_variant(v)
        vol.reset()   # same starting state for every variant (and a valid free-space summary)
        vol.integrate_sequence_timed(d.data_ptr(), poses[:5])
        ms = vol.integrate_sequence_timed(d.data_ptr(), poses) / len(poses)
        res.setdefault(v, []).append(ms)
_, w = vol.download()
upd = float(w.astype(np.float64).sum()) / (len(poses) + 5)
print(f"workload {workload} D={D}: updated fraction {upd / N:.3f}")
for v in variants:
    ms = min(res[v]); med = sorted(res[v])[1]
    c = (v - 16) & 15
    desc = "rows<4> v0" if v == 2 else f"tile R={[1, 2, 4][c >> 2]} elide={(c >> 1) & 1} nt={c & 1} sum={int(32 <= v < 64 or 80 <= v < 96)} early={int(48 <= v < 80)} fast={int(v >= 80)} lds={int(v >= 112)}"
    print(f"variant {v:2d} {desc:28s} min {ms:.4f} med {med:.4f} ms  {N / ms / 1e3:9.0f} Mvox/s  alg {(16 * upd + 1.2e6) / ms / 1e6:7.0f} GB/s")
