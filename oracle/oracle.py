"""ctypes loader for the CPU checker (oracle/liboracle.so, oracle/_ref/libtsdf_ref.so).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Importers allowed: tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  Nothing under semantic_slam_amd/ imports this module.

`Oracle` wraps this project's CPU restatement (tsdf_oracle.c; each C function cites the
reference lines it follows).  `Ref` wraps the reference's own kernel body compiled for the
host from /root/reference/src/tsdf.cu:15-60 (`make -C oracle ref`), when that build exists;
`RefHip` wraps the same function compiled by hipcc for gfx950 as it stands (`make -C oracle ref_hip`).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(ref=True):
    """Compile the checker.  The reference body is only buildable where /root/reference exists."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if ref and os.path.isfile("/root/reference/src/tsdf.cu"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref", "ref_hip"])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """CPU restatement (kind "port" in bench.py's cpu_baseline)."""

    def __init__(self):
        # TSDF_ORACLE_LIB: another build of the same sources (the sanitizer build, oracle/_asan/liboracle_asan.so)
        path = os.environ.get("TSDF_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
        if not os.path.isfile(path):
            build(ref=False)
        L = self.lib = C.CDLL(path)
        L.oracle_integrate.restype = C.c_int64
        L.oracle_integrate.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, _f32p, _f32p, C.c_int]
        L.oracle_init_grid.argtypes = [_f32p, _f32p, C.c_int64]
        L.oracle_multiply_matrix.argtypes = [_f32p, _f32p, _f32p]
        L.oracle_invert_matrix.restype = C.c_int
        L.oracle_invert_matrix.argtypes = [_f32p, _f32p]
        L.oracle_cam2base.argtypes = [_f32p, _f32p, _f32p]
        L.oracle_surface_points.restype = C.c_int64
        L.oracle_surface_points.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float,
                                            C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                            C.c_void_p]
        L.oracle_zero_crossings.restype = C.c_int64
        L.oracle_zero_crossings.argtypes = [_f32p, _f32p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.oracle_mesh_triangles.restype = C.c_int64
        L.oracle_mesh_triangles.argtypes = L.oracle_zero_crossings.argtypes
        L.oracle_save_ply.restype = C.c_int
        L.oracle_save_ply.argtypes = [C.c_char_p, _f32p, _f32p, C.c_int, C.c_int, C.c_int,
                                      C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.c_float]
        L.oracle_save_bin.restype = C.c_int
        L.oracle_save_bin.argtypes = [C.c_char_p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.c_float, C.c_float, C.c_float, C.c_float]
        L.oracle_object_origin.argtypes = [_f32p, C.c_int, C.c_int, _f32p, _f32p]
        L.oracle_depth_prep.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS"),
                                        C.c_int, C.c_int, C.c_float, _f32p]
        L.oracle_mask_depth.argtypes = [_f32p, np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS"),
                                        C.c_int, _f32p]
        u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
        L.oracle_integrate_labels.restype = C.c_int64
        L.oracle_integrate_labels.argtypes = [_f32p, _f32p, _f32p, u16p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                              C.c_float, C.c_float, u16p, _f32p, _f32p]
        L.oracle_compose_labels.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS"), u16p, _f32p,
                                            C.c_int, C.c_int, u16p, _f32p]
        L.oracle_max_threads.restype = C.c_int
        u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
        L.oracle_integrate_colour.restype = C.c_int64
        L.oracle_integrate_colour.argtypes = [_f32p, _f32p, _f32p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _f32p, u32p]

    # --- grid ---------------------------------------------------------------------------
    def init_grid(self, dims, z_begin=0, z_end=None):
        dx, dy, dz = dims
        z_end = dz if z_end is None else z_end
        n = dx * dy * (z_end - z_begin)
        t = np.empty(n, np.float32)
        w = np.empty(n, np.float32)
        self.lib.oracle_init_grid(t, w, n)
        return t, w

    def integrate(self, K, cam2base, depth, dims, origin, voxel_size, trunc, tsdf, weight,
                  z_begin=0, z_end=None, max_depth=6.0, threads=0):
        """In-place update of tsdf/weight (slab-local arrays).  Returns voxels updated."""
        dx, dy, dz = dims
        z_end = dz if z_end is None else z_end
        h, w = depth.shape
        assert tsdf.size == dx * dy * (z_end - z_begin) == weight.size
        return int(self.lib.oracle_integrate(_f32(K).ravel(), _f32(cam2base).ravel(), _f32(depth),
                                             h, w, dx, dy, dz, z_begin, z_end,
                                             origin[0], origin[1], origin[2], voxel_size, trunc,
                                             max_depth, tsdf, weight, threads))

    # --- pose math ----------------------------------------------------------------------
    def multiply(self, a, b):
        out = np.empty(16, np.float32)
        self.lib.oracle_multiply_matrix(_f32(a).ravel(), _f32(b).ravel(), out)
        return out

    def invert(self, m):
        out = np.zeros(16, np.float32)
        ok = self.lib.oracle_invert_matrix(_f32(m).ravel(), out)
        return bool(ok), out

    def cam2base(self, base2world, cam2world):
        out = np.empty(16, np.float32)
        self.lib.oracle_cam2base(_f32(base2world).ravel(), _f32(cam2world).ravel(), out)
        return out

    # --- outputs ------------------------------------------------------------------------
    def surface_points(self, tsdf, weight, dims, voxel_size, origin, weight_thresh=0.9):
        dx, dy, dz = dims
        args = (tsdf, weight, dx, dy, dz, voxel_size, origin[0], origin[1], origin[2], 1.2,
                weight_thresh)
        n = self.lib.oracle_surface_points(*args, None)
        xyz = np.empty((n, 3), np.float32)
        self.lib.oracle_surface_points(*args, xyz.ctypes.data)
        return xyz

    def zero_crossings(self, tsdf, weight, dims_xy, z_begin, z_end, voxel_size, origin, halo=None,
                       weight_thresh=0.9):
        """Zero-crossing vertices of slab [z_begin, z_end) (project-defined rule, see tsdf_oracle.c).
        halo = (tsdf, weight) of slice z_end or None."""
        ht = hw = None
        if halo is not None:
            ht_a, hw_a = _f32(halo[0]), _f32(halo[1])
            ht, hw = ht_a.ctypes.data, hw_a.ctypes.data
        args = (tsdf, weight, ht, hw, dims_xy[0], dims_xy[1], z_begin, z_end, voxel_size,
                origin[0], origin[1], origin[2], weight_thresh)
        n = self.lib.oracle_zero_crossings(*args, None)
        xyz = np.empty((n, 3), np.float32)
        self.lib.oracle_zero_crossings(*args, xyz.ctypes.data)
        return xyz

    def mesh_triangles(self, tsdf, weight, dims_xy, z_begin, z_end, voxel_size, origin, halo=None,
                       weight_thresh=0.9):
        """Marching-tetrahedra triangles of slab [z_begin, z_end): array [n, 3, 3] (project-defined rule)."""
        ht = hw = None
        if halo is not None:
            ht_a, hw_a = _f32(halo[0]), _f32(halo[1])
            ht, hw = ht_a.ctypes.data, hw_a.ctypes.data
        args = (tsdf, weight, ht, hw, dims_xy[0], dims_xy[1], z_begin, z_end, voxel_size,
                origin[0], origin[1], origin[2], weight_thresh)
        n = self.lib.oracle_mesh_triangles(*args, None)
        tri = np.empty((n, 3, 3), np.float32)
        self.lib.oracle_mesh_triangles(*args, tri.ctypes.data)
        return tri

    def save_ply(self, path, tsdf, weight, dims, voxel_size, origin, weight_thresh=0.9):
        dx, dy, dz = dims
        rc = self.lib.oracle_save_ply(os.fsencode(path), tsdf, weight, dx, dy, dz, voxel_size,
                                      origin[0], origin[1], origin[2], 1.2, weight_thresh)
        assert rc == 0

    def save_bin(self, path, tsdf, dims, origin, voxel_size, trunc):
        dx, dy, dz = dims
        rc = self.lib.oracle_save_bin(os.fsencode(path), tsdf, dx, dy, dz, origin[0], origin[1],
                                      origin[2], voxel_size, trunc)
        assert rc == 0

    # --- caller-side adapters -----------------------------------------------------------
    def object_origin(self, depth, K):
        out = np.empty(3, np.float32)
        h, w = depth.shape
        self.lib.oracle_object_origin(_f32(depth), h, w, _f32(K).ravel(), out)
        return out

    def depth_prep(self, raw_u16, factor=5000.0):
        raw = np.ascontiguousarray(raw_u16, dtype=np.uint16)
        out = np.empty(raw.shape, np.float32)
        self.lib.oracle_depth_prep(raw, raw.shape[0], raw.shape[1], factor, out)
        return out

    def mask_depth(self, depth, mask_u8):
        d = _f32(depth)
        out = np.empty_like(d)
        self.lib.oracle_mask_depth(d, np.ascontiguousarray(mask_u8, dtype=np.uint8), d.size, out)
        return out

    # --- per-voxel label fusion (project-defined rule, see tsdf_oracle.c) ----------------------
    def integrate_labels(self, K, cam2base, depth, label_im, score_im, dims, origin, voxel_size, trunc,
                         label, fp, bp, z_begin=0, z_end=None, max_depth=6.0, prob_thd=0.5):
        dx, dy, dz = dims
        z_end = dz if z_end is None else z_end
        h, w = depth.shape
        return int(self.lib.oracle_integrate_labels(
            _f32(K).ravel(), _f32(cam2base).ravel(), _f32(depth), np.ascontiguousarray(label_im, np.uint16),
            _f32(score_im), h, w, dx, dy, z_begin, z_end, origin[0], origin[1], origin[2], voxel_size, trunc,
            max_depth, prob_thd, label, fp, bp))

    # --- per-voxel colour fusion (tsdf-fusion-python's published rule, see tsdf_oracle.c) ----------
    def integrate_colour(self, K, cam2base, depth, rgb, dims, origin, voxel_size, trunc, weight, colour,
                         z_begin=0, z_end=None, max_depth=6.0):
        """Colour pass of a frame whose oracle.integrate has already run (weight holds w_new).  colour: uint32 in place."""
        dx, dy, dz = dims
        z_end = dz if z_end is None else z_end
        h, w = depth.shape
        return int(self.lib.oracle_integrate_colour(
            _f32(K).ravel(), _f32(cam2base).ravel(), _f32(depth), np.ascontiguousarray(rgb, np.uint8), h, w, dx, dy,
            z_begin, z_end, origin[0], origin[1], origin[2], voxel_size, trunc, max_depth, weight, colour))

    def compose_labels(self, masks, labels, scores):
        masks = np.ascontiguousarray(masks, np.uint8)
        k, h, w = masks.shape
        lab = np.empty((h, w), np.uint16)
        sc = np.empty((h, w), np.float32)
        self.lib.oracle_compose_labels(masks, np.ascontiguousarray(labels, np.uint16), _f32(scores), k, h * w, lab, sc)
        return lab, sc

    def max_threads(self):
        return int(self.lib.oracle_max_threads())


class Ref:
    """The reference's own GpuIntegrate body on the host (kind "reference").

    Whole-grid only (the reference has no slabs) and max depth fixed at 6 m (tsdf.cu:46).
    """

    path = os.path.join(_HERE, "_ref", "libtsdf_ref.so")

    @classmethod
    def available(cls):
        return os.path.isfile(cls.path)

    def __init__(self):
        L = self.lib = C.CDLL(self.path)
        L.ref_integrate.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                    _f32p, _f32p, C.c_int]
        L.ref_max_threads.restype = C.c_int

    def integrate(self, K, cam2base, depth, dims, origin, voxel_size, trunc, tsdf, weight, threads=0):
        dx, dy, dz = dims
        h, w = depth.shape
        assert tsdf.size == dx * dy * dz == weight.size
        self.lib.ref_integrate(_f32(K).ravel(), _f32(cam2base).ravel(), _f32(depth), h, w,
                               dx, dy, dz, origin[0], origin[1], origin[2], voxel_size, trunc,
                               tsdf, weight, threads)

    def max_threads(self):
        return int(self.lib.ref_max_threads())


class RefHost:
    """The reference's own host functions that compile as they stand: multiply_matrix / invert_matrix (src/tsdf.cu:253-403)
    and the .ply writer SaveVoxelGrid2SurfacePointCloud (src/tsdf.cu:170-218).  oracle/Makefile, ref_host;
    oracle/ref_host_driver.cpp says what is and is not the reference's in that build."""

    path = os.path.join(_HERE, "_ref", "libtsdf_ref_host.so")

    @classmethod
    def available(cls):
        return os.path.isfile(cls.path)

    def __init__(self):
        L = self.lib = C.CDLL(self.path)
        L.ref_multiply_matrix.argtypes = [_f32p, _f32p, _f32p]
        L.ref_multiply_matrix.restype = None
        L.ref_invert_matrix.argtypes = [_f32p, _f32p]
        L.ref_invert_matrix.restype = C.c_int
        L.ref_save_ply.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                   _f32p, _f32p, C.c_float, C.c_float]
        L.ref_save_ply.restype = None

    def multiply(self, a, b):
        out = np.empty(16, np.float32)
        self.lib.ref_multiply_matrix(_f32(a).ravel(), _f32(b).ravel(), out)
        return out

    def invert(self, m):
        """(ok, inverse); as in the reference the output is left untouched when det == 0 (returned as NaNs here)."""
        out = np.full(16, np.nan, np.float32)
        ok = self.lib.ref_invert_matrix(_f32(m).ravel(), out)
        return bool(ok), out

    def save_ply(self, path, tsdf, weight, dims, voxel_size, origin, tsdf_thresh=1.2, weight_thresh=0.9):
        """tsdf<id>.ply as ~TSDF writes it (ref: src/tsdf.cu:110-112 passes 1.2f, 0.9f)."""
        dx, dy, dz = (int(d) for d in dims)
        t, w = _f32(tsdf).ravel(), _f32(weight).ravel()
        assert t.size == dx * dy * dz == w.size
        self.lib.ref_save_ply(os.fsencode(path), dx, dy, dz, voxel_size, origin[0], origin[1], origin[2], t, w, tsdf_thresh, weight_thresh)


class RefHip:
    """The reference's own GpuIntegrate compiled for gfx950 by hipcc exactly as it stands
    (`make -C oracle ref_hip`: no stand-in for anything) and launched with the reference's shape
    <<<dim_z, dim_y>>> on the GPU.  Device pointers in, device arrays updated in place; whole grid
    only, dim_y <= 1024 (the reference's block size), max depth fixed at 6 m (tsdf.cu:46)."""

    path = os.path.join(_HERE, "_ref", "libtsdf_ref_hip.so")

    @classmethod
    def available(cls):
        return os.path.isfile(cls.path)

    def __init__(self):
        L = self.lib = C.CDLL(self.path)
        vp = C.c_void_p
        L.ref_hip_integrate.restype = C.c_int
        L.ref_hip_integrate.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp]

    def integrate(self, K_ptr, cam2base_ptr, depth_ptr, h, w, dims, origin, voxel_size, trunc, tsdf_ptr, weight_ptr):
        dx, dy, dz = dims
        rc = self.lib.ref_hip_integrate(K_ptr, cam2base_ptr, depth_ptr, h, w, dx, dy, dz, origin[0], origin[1],
                                        origin[2], voxel_size, trunc, tsdf_ptr, weight_ptr)
        assert rc == 0, f"reference kernel launch failed ({rc})"
