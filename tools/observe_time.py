"""Development probe: what observing a 512^3 volume costs -- download, surface points, the two files ~TSDF writes, a checkpoint,
a device-to-device copy (python tools/observe_time.py)."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_slam_amd import capi, synth
D, vs = 512, 0.005
dims = (D, D, D)
origin = synth.surf_volume(D, vs, 1.0)
cfg = capi.make_config(dims, vs, origin)
scene = synth.SurfScene(dims, vs, origin)
vol = capi.Volume(cfg)
for k in range(4):
    p = scene.pose(k * 4, 64)
    vol.integrate(scene.depth(p, quantize=True), p)
vol.sync()
def T(name, fn, n=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt*1e3:.1f} ms", flush=True)
    return r
d = tempfile.mkdtemp(dir="/tmp")
T("download (2 x 537 MB to pageable numpy)", vol.download)
T("extract_surface (39 M points -> host)", vol.extract_surface)
T("save_ply", lambda: vol.save_ply(os.path.join(d, "a.ply")))
T("save_bin", lambda: vol.save_bin(os.path.join(d, "a.bin")))
T("save_state", lambda: vol.save_state(os.path.join(d, "a.state")))
t = torch.empty(D**3, dtype=torch.float32, device="cuda"); w = torch.empty_like(t)
T("copy_slices_to_device (D2D)", lambda: vol.copy_slices_to_device(0, D, t.data_ptr(), w.data_ptr()))
