"""Python mirror of the reference's `class TSDF` (ref: include/tsdf.hpp:22-43) over the C ABI -- the
same names and argument meaning as the C++ drop-in in include/tsdf.hpp, for scripts and tests.

    tsdf = TSDF(480, 640, obj_id, base2world_16, origin_3)     # 200^3 @ 4 mm, TUM intrinsics
    tsdf.Integrate(depth_hw_float32_metres, cam2world_16)
    tsdf.Download(); tsdf.voxel_grid_TSDF, tsdf.voxel_grid_weight
    tsdf.close()          # what the C++ destructor does: tsdf<id>.ply and tsdf<id>.bin in the CWD
"""
import os

import numpy as np

from . import capi


class TSDF:
    def __init__(self, h, w, id, base2world_vec, origin, cfg=None, save_on_close=True):
        if cfg is None:
            cfg = capi.default_config(h, w)                      # ref: include/tsdf.hpp:63-67,96
            cfg.origin[:] = [float(x) for x in np.asarray(origin, np.float32).ravel()[:3]]
            cfg.base2world[:] = [float(x) for x in np.asarray(base2world_vec, np.float32).ravel()[:16]]
            cfg.id = int(id)
        self._vol = capi.Volume(cfg)
        self.cfg = cfg
        self.save_on_close = save_on_close
        n = self._vol.n_voxels
        # ref: src/tsdf.cu:77-81 -- host mirrors exist from construction, TSDF = 1, weight = 0
        self.voxel_grid_TSDF = np.ones(n, np.float32)
        self.voxel_grid_weight = np.zeros(n, np.float32)

    def Integrate(self, depth_im, cam2world_vec):
        """ref: src/tsdf.cu:135-168 -- depth in metres (h*w floats), 16-float row-major camera pose."""
        self._vol.integrate(depth_im, cam2world_vec)

    def Download(self):
        self.voxel_grid_TSDF, self.voxel_grid_weight = self._vol.download()

    def close(self, directory="."):
        """ref: src/tsdf.cu:98-133 -- download, write tsdf<id>.ply (weight threshold 0.9) and tsdf<id>.bin."""
        if self._vol is None:
            return
        if self.save_on_close:
            self.Download()
            self._vol.save_ply(os.path.join(directory, f"tsdf{self.cfg.id}.ply"), 0.9)
            self._vol.save_bin(os.path.join(directory, f"tsdf{self.cfg.id}.bin"))
        self._vol.close()
        self._vol = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
