"""Development probe: tsdf_download of a 512^3 volume into touched host pages, from one, two and four host threads."""
import os, sys, time, threading, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semantic_slam_amd import capi, synth
D, vs = 512, 0.005
cfg = capi.make_config((D, D, D), vs, synth.surf_volume(D, vs, 1.0))
vol = capi.Volume(cfg)
n = D ** 3
t = np.empty(n, np.float32); w = np.empty(n, np.float32)
t[:] = 0; w[:] = 0      # touch the pages
lib = vol.lib
def dl(tt, ww):
    return lib.tsdf_download(vol._h, tt.ctypes.data if tt is not None else None, ww.ctypes.data if ww is not None else None)
dl(t, w)
for rep in range(3):
    t0 = time.perf_counter(); dl(t, w); t1 = time.perf_counter()
    a = threading.Thread(target=dl, args=(t, None)); b = threading.Thread(target=dl, args=(None, w))
    a.start(); b.start(); a.join(); b.join()
    t2 = time.perf_counter()
    # four threads, halves of each array through copy_slices
    def cs(z0, nz, tt, ww):
        lib.tsdf_copy_slices(vol._h, z0, nz, tt, ww)
    s = D * D
    th = [threading.Thread(target=cs, args=(z0, 128, t.ctypes.data + 4 * s * z0, w.ctypes.data + 4 * s * z0)) for z0 in (0, 128, 256, 384)]
    for x in th: x.start()
    for x in th: x.join()
    t3 = time.perf_counter()
    print(f"sequential {1e3*(t1-t0):.1f} ms   two threads {1e3*(t2-t1):.1f} ms   four threads (copy_slices) {1e3*(t3-t2):.1f} ms", flush=True)
