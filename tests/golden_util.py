"""Loader for the fixtures in tests/golden/ (made by tests/golden/make_golden.py from the
reference's own kernel body; data only)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# g<N>_*.npz are the Integrate vectors; other files in the directory are other fixtures
NAMES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "g[0-9]_*.npz")))


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.name = name
        self.dims = tuple(int(x) for x in z["dims"])
        self.vs = float(z["voxel_size"])
        self.trunc = float(z["trunc"])
        self.origin = z["origin"].astype(np.float32)
        self.K = z["K"].astype(np.float32)
        self.cam2base = z["cam2base"].astype(np.float32)
        raw = z["depth"]
        if bool(z["depth_is_u16"]):
            # TUM convention: uint16 / 5000 in fp32 (ref: examples/label_instance_rgbd.cpp:99-100)
            self.depth = (raw.astype(np.float32) * np.float32(1.0 / 5000.0)).astype(np.float32)
        else:
            self.depth = raw.astype(np.float32)
        self.tsdf = z["tsdf"]
        self.weight = z["weight"]

    @property
    def frames(self):
        return list(zip(self.cam2base, self.depth))
