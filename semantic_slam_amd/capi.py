"""ctypes binding of the C ABI in include/tsdf_hip.h (libtsdf_hip.so).

This is plumbing for tests and bench.py: the same entry points a C++ caller reaches through
include/tsdf.hpp.  There is no fallback of any kind here: if the shared library is missing
or a call fails, a TsdfError is raised.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# TSDF_HIP_LIB overrides the library path (A/B timing of two builds on one GPU box; tools/sweep.py)
LIB_PATH = os.environ.get("TSDF_HIP_LIB") or os.path.join(_PKG, "libtsdf_hip.so")

# every symbol include/tsdf_hip.h declares (tests check the .so exports exactly these)
ABI_SYMBOLS = [
    "tsdf_config_default", "tsdf_create", "tsdf_destroy", "tsdf_reset", "tsdf_integrate",
    "tsdf_integrate_u16", "tsdf_convert_depth_u16", "tsdf_set_deferral",
    "tsdf_integrate_device", "tsdf_integrate_cam2base", "tsdf_integrate_masked_device",
    "tsdf_integrate_frames_device",
    "tsdf_sync", "tsdf_download", "tsdf_copy_slices", "tsdf_upload", "tsdf_refresh_summary", "tsdf_device_ptrs", "tsdf_slab_voxels", "tsdf_frames_per_launch", "tsdf_shortcut_stats", "tsdf_brick_list_stats", "tsdf_classification_info",
    "tsdf_get_config", "tsdf_last_cam2base", "tsdf_set_stream", "tsdf_get_stream",
    "tsdf_count_surface", "tsdf_extract_surface", "tsdf_extract_crossings", "tsdf_extract_mesh", "tsdf_save_mesh_ply", "tsdf_save_mesh_welded_ply", "tsdf_save_ply", "tsdf_save_bin", "tsdf_load_bin", "tsdf_save_state", "tsdf_load_state",
    "tsdf_integrate_sequence_timed", "tsdf_integrate_frames_timed", "tsdf_probe_graph_replay", "tsdf_probe_stream", "tsdf_selftest_fastdiv", "tsdf_selftest_fastdiv_band", "tsdf_selftest_round", "tsdf_selftest_tile_tables", "tsdf_set_kernel_variant", "tsdf_set_brick_shape", "tsdf_brick_shape", "tsdf_default_brick_shape", "tsdf_last_error",
    "tsdf_version", "tsdf_multiply_matrix", "tsdf_invert_matrix",
    "tsdf_labels_enable", "tsdf_compose_labels", "tsdf_integrate_labels_device", "tsdf_integrate_frames_labels_device",
    "tsdf_download_labels",
    "tsdf_colour_enable", "tsdf_integrate_colour_device", "tsdf_integrate_rgbd", "tsdf_download_colour",
    "tsdf_object_origin", "tsdf_batch_create", "tsdf_batch_destroy", "tsdf_batch_size", "tsdf_batch_volume",
    "tsdf_batch_integrate_device", "tsdf_batch_sync",
    "tsdf_group_create", "tsdf_group_destroy", "tsdf_group_size", "tsdf_group_voxels", "tsdf_group_volume",
    "tsdf_group_integrate", "tsdf_group_integrate_frames", "tsdf_group_set_deferral", "tsdf_group_sync", "tsdf_group_reset", "tsdf_group_download",
    "tsdf_group_extract_surface", "tsdf_group_extract_crossings", "tsdf_group_extract_mesh",
    "tsdf_group_save_ply", "tsdf_group_save_mesh_ply", "tsdf_group_save_bin",
]


class TsdfError(RuntimeError):
    pass


# kernel variants of the library as shipped (tsdf_set_kernel_variant; csrc/tsdf_capi.hip); every other number exists only in
# the measurement build (`make -C semantic_slam_amd/csrc experiments`, loaded through TSDF_HIP_LIB)
SHIPPED_VARIANTS = (0, 1, 3, 7, 8)


def experiments_build():
    """True when the loaded library is the measurement build (-DTSDF_EXPERIMENTS)."""
    return b"+experiments" in load().tsdf_version()


def variants(*wanted):
    """Those of `wanted` the loaded library knows: the shipped ones, and the experiments when that build is loaded (tests
    parametrise over this, so the product's test matrix is exactly what ships)."""
    exp = experiments_build()
    return [v for v in wanted if v in SHIPPED_VARIANTS or exp]


class TsdfConfig(C.Structure):
    """Mirror of `struct tsdf_config` (include/tsdf_hip.h)."""
    _fields_ = [
        ("im_height", C.c_int32), ("im_width", C.c_int32),
        ("dim_x", C.c_int32), ("dim_y", C.c_int32), ("dim_z", C.c_int32),
        ("z_begin", C.c_int32), ("z_end", C.c_int32),
        ("voxel_size", C.c_float), ("trunc_margin", C.c_float), ("max_depth", C.c_float),
        ("origin", C.c_float * 3), ("cam_K", C.c_float * 9), ("base2world", C.c_float * 16),
        ("device", C.c_int32), ("id", C.c_int32),
    ]


_lib = None


def load():
    """Load libtsdf_hip.so (built by `make -C semantic_slam_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise TsdfError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)")
    # One HIP runtime per process: PyTorch ships its own libamdhip64, this library links ROCm's.  Whichever is loaded
    # first serves both (same SONAME); loaded the other way round the process ends up with two runtimes, and the second
    # one to initialise sees no device (tsdf_create: "no HIP device visible").  Callers hand torch device pointers to this
    # library anyway, so torch -- when it is installed -- goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, f32p, i64p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int64)
    L.tsdf_config_default.argtypes = [C.POINTER(TsdfConfig), C.c_int32, C.c_int32]
    L.tsdf_create.argtypes = [C.POINTER(TsdfConfig), C.POINTER(vp)]
    L.tsdf_destroy.argtypes = [vp]
    L.tsdf_reset.argtypes = [vp]
    L.tsdf_integrate.argtypes = [vp, vp, vp]
    L.tsdf_set_deferral.argtypes = [vp, C.c_int32]
    L.tsdf_integrate_u16.argtypes = [vp, vp, C.c_float, C.c_int32, C.c_int32, vp]
    L.tsdf_convert_depth_u16.argtypes = [vp, vp, vp, C.c_float, C.c_int32, C.c_int32]
    L.tsdf_integrate_device.argtypes = [vp, vp, vp]
    L.tsdf_integrate_cam2base.argtypes = [vp, vp, vp]
    L.tsdf_integrate_masked_device.argtypes = [vp, vp, vp, vp]
    L.tsdf_integrate_frames_device.argtypes = [vp, vp, vp, vp, C.c_int32]
    L.tsdf_sync.argtypes = [vp]
    L.tsdf_download.argtypes = [vp, vp, vp]
    L.tsdf_upload.argtypes = [vp, vp, vp]
    L.tsdf_copy_slices.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    L.tsdf_refresh_summary.argtypes = [vp]
    L.tsdf_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.tsdf_slab_voxels.argtypes = [vp]
    L.tsdf_slab_voxels.restype = C.c_int64
    L.tsdf_frames_per_launch.argtypes = [vp]
    L.tsdf_frames_per_launch.restype = C.c_int32
    L.tsdf_shortcut_stats.argtypes = [vp, C.c_int32, vp]
    L.tsdf_classification_info.argtypes = [vp, vp]
    L.tsdf_brick_list_stats.argtypes = [vp, vp]
    L.tsdf_get_config.argtypes = [vp, C.POINTER(TsdfConfig)]
    L.tsdf_last_cam2base.argtypes = [vp, vp]
    L.tsdf_set_stream.argtypes = [vp, vp]
    L.tsdf_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.tsdf_count_surface.argtypes = [vp, C.c_float, i64p]
    L.tsdf_extract_surface.argtypes = [vp, C.c_float, vp, C.c_int64, i64p]
    L.tsdf_extract_crossings.argtypes = [vp, vp, vp, C.c_float, vp, C.c_int64, i64p]
    L.tsdf_extract_mesh.argtypes = [vp, vp, vp, C.c_float, vp, C.c_int64, i64p]
    L.tsdf_save_mesh_ply.argtypes = [vp, C.c_char_p, C.c_float]
    L.tsdf_save_mesh_welded_ply.argtypes = [vp, C.c_char_p, C.c_float]
    L.tsdf_save_ply.argtypes = [vp, C.c_char_p, C.c_float]
    L.tsdf_save_bin.argtypes = [vp, C.c_char_p]
    L.tsdf_load_bin.argtypes = [vp, C.c_char_p]
    L.tsdf_save_state.argtypes = [vp, C.c_char_p]
    L.tsdf_load_state.argtypes = [vp, C.c_char_p]
    L.tsdf_integrate_sequence_timed.argtypes = [vp, vp, vp, C.c_int32, f32p]
    L.tsdf_integrate_frames_timed.argtypes = [vp, vp, vp, vp, C.c_int32, f32p]
    L.tsdf_probe_graph_replay.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, f32p, f32p]
    L.tsdf_probe_stream.argtypes = [vp, C.c_int32, C.c_int32, f32p]
    L.tsdf_selftest_fastdiv.argtypes = [C.c_int32, C.c_uint64, C.c_uint64, C.c_float, C.c_float, C.POINTER(C.c_uint64), f32p]
    L.tsdf_selftest_fastdiv_band.argtypes = [C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), f32p]
    L.tsdf_selftest_round.argtypes = [C.c_int32, C.POINTER(C.c_uint64), f32p]
    L.tsdf_selftest_tile_tables.argtypes = [C.c_int32, vp, vp, C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_uint64)]
    L.tsdf_set_kernel_variant.argtypes = [vp, C.c_int32]
    L.tsdf_set_brick_shape.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    L.tsdf_brick_shape.argtypes = [vp, C.POINTER(C.c_int32)]
    L.tsdf_default_brick_shape.argtypes = [C.POINTER(TsdfConfig), C.POINTER(C.c_int32)]
    L.tsdf_last_error.restype = C.c_char_p
    L.tsdf_version.restype = C.c_char_p
    L.tsdf_multiply_matrix.argtypes = [vp, vp, vp]
    L.tsdf_multiply_matrix.restype = None
    L.tsdf_invert_matrix.argtypes = [vp, vp]
    L.tsdf_labels_enable.argtypes = [vp, C.c_float]
    L.tsdf_compose_labels.argtypes = [vp, vp, vp, vp, C.c_int32, vp, vp]
    L.tsdf_integrate_labels_device.argtypes = [vp, vp, vp, vp, vp]
    L.tsdf_download_labels.argtypes = [vp, vp, vp, vp]
    L.tsdf_integrate_frames_labels_device.argtypes = [vp, vp, vp, vp, vp, C.c_int32]
    L.tsdf_colour_enable.argtypes = [vp]
    L.tsdf_integrate_colour_device.argtypes = [vp, vp, vp, vp]
    L.tsdf_integrate_rgbd.argtypes = [vp, vp, vp, vp]
    L.tsdf_download_colour.argtypes = [vp, vp]
    L.tsdf_object_origin.argtypes = [C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp]
    L.tsdf_batch_create.argtypes = [C.POINTER(TsdfConfig), C.c_int32, C.POINTER(vp)]
    L.tsdf_batch_destroy.argtypes = [vp]
    L.tsdf_batch_size.argtypes = [vp]
    L.tsdf_batch_volume.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.tsdf_batch_integrate_device.argtypes = [vp, vp, vp, vp]
    L.tsdf_batch_sync.argtypes = [vp]
    L.tsdf_group_create.argtypes = [C.POINTER(TsdfConfig), C.POINTER(C.c_int32), C.c_int32, C.POINTER(vp)]
    L.tsdf_group_destroy.argtypes = [vp]
    L.tsdf_group_size.argtypes = [vp]
    L.tsdf_group_voxels.argtypes = [vp]
    L.tsdf_group_voxels.restype = C.c_int64
    L.tsdf_group_volume.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.tsdf_group_integrate.argtypes = [vp, vp, vp]
    L.tsdf_group_integrate_frames.argtypes = [vp, vp, vp, C.c_int32]
    L.tsdf_group_set_deferral.argtypes = [vp, C.c_int32]
    L.tsdf_group_sync.argtypes = [vp]
    L.tsdf_group_reset.argtypes = [vp]
    L.tsdf_group_download.argtypes = [vp, vp, vp]
    for name in ("tsdf_group_extract_surface", "tsdf_group_extract_crossings", "tsdf_group_extract_mesh"):
        getattr(L, name).argtypes = [vp, C.c_float, vp, C.c_int64, i64p]
    L.tsdf_group_save_ply.argtypes = [vp, C.c_char_p, C.c_float]
    L.tsdf_group_save_mesh_ply.argtypes = [vp, C.c_char_p, C.c_float]
    L.tsdf_group_save_bin.argtypes = [vp, C.c_char_p]
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        raise TsdfError(f"{what} failed ({rc}): {load().tsdf_last_error().decode()}")


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32).ravel()
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} floats, got {a.size}")
    return a


def default_config(im_height=480, im_width=640):
    cfg = TsdfConfig()
    check(load().tsdf_config_default(C.byref(cfg), im_height, im_width), "tsdf_config_default")
    return cfg


def make_config(dims, voxel_size, origin, trunc=None, K=None, base2world=None, z_begin=0, z_end=None,
                im_height=480, im_width=640, max_depth=None, device=0, vol_id=0):
    """Reference defaults (include/tsdf.hpp:63-67,96) with the given overrides."""
    cfg = default_config(im_height, im_width)
    cfg.dim_x, cfg.dim_y, cfg.dim_z = (int(d) for d in dims)
    cfg.z_begin = int(z_begin)
    cfg.z_end = int(cfg.dim_z if z_end is None else z_end)
    cfg.voxel_size = voxel_size
    # the reference's member initialiser: trunc_margin = voxel_size * 5 in fp32
    cfg.trunc_margin = float(np.float32(voxel_size) * np.float32(5)) if trunc is None else trunc
    if max_depth is not None:
        cfg.max_depth = max_depth
    cfg.origin[:] = [float(x) for x in _f32(origin, 3)]
    if K is not None:
        cfg.cam_K[:] = [float(x) for x in _f32(K, 9)]
    if base2world is not None:
        cfg.base2world[:] = [float(x) for x in _f32(base2world, 16)]
    cfg.device = device
    cfg.id = vol_id
    return cfg


def multiply_matrix(a, b):
    out = np.empty(16, np.float32)
    a, b = _f32(a, 16), _f32(b, 16)
    load().tsdf_multiply_matrix(a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def invert_matrix(m):
    out = np.zeros(16, np.float32)
    m = _f32(m, 16)
    ok = load().tsdf_invert_matrix(m.ctypes.data, out.ctypes.data)
    return bool(ok), out


def selftest_fastdiv(n_samples, seed=1, device=0, fx=535.4, cx=320.1):
    """Returns (mismatches, first_bad[4]) of the device division self-test."""
    cnt = C.c_uint64()
    bad = (C.c_float * 4)()
    check(load().tsdf_selftest_fastdiv(device, seed, n_samples, fx, cx, C.byref(cnt), bad), "tsdf_selftest_fastdiv")
    return cnt.value, list(bad)


def selftest_fastdiv_band(n_samples, seed=1, device=0):
    """Returns (mismatches, first_bad[4]) of the device self-test of diff / trunc through the shared reciprocal."""
    cnt = C.c_uint64()
    bad = (C.c_float * 4)()
    check(load().tsdf_selftest_fastdiv_band(device, seed, n_samples, C.byref(cnt), bad), "tsdf_selftest_fastdiv_band")
    return cnt.value, list(bad)


def object_origin(depth_ptr, mask_ptr, h, w, K, device=0):
    """Origin of a new object volume from its first masked depth frame (ref: src/Object.cpp:37-49), on the device."""
    k = _f32(K, 9)
    out = np.empty(3, np.float32)
    check(load().tsdf_object_origin(device, depth_ptr, mask_ptr, h, w, k.ctypes.data, out.ctypes.data), "tsdf_object_origin")
    return out


def selftest_round(device=0):
    """Returns (mismatches, first_bad[4]) of the exhaustive pixel-rounding self-test."""
    cnt = C.c_uint64()
    bad = (C.c_float * 4)()
    check(load().tsdf_selftest_round(device, C.byref(cnt), bad), "tsdf_selftest_round")
    return cnt.value, list(bad)


def default_brick_shape(cfg):
    """(quads, rows, slices) of the wavefront brick tsdf_create would choose for this grid; host arithmetic only."""
    out = (C.c_int32 * 3)()
    check(load().tsdf_default_brick_shape(C.byref(cfg), out), "tsdf_default_brick_shape")
    return tuple(out)


def selftest_tile_tables(depth_ptr, mask_ptr, im_height, im_width, max_depth=6.0, device=0):
    """Entries of one frame's depth tile table that differ between the launched kernels and the plain ones (must be 0)."""
    cnt = C.c_uint64()
    check(load().tsdf_selftest_tile_tables(device, depth_ptr, mask_ptr, im_height, im_width, max_depth, C.byref(cnt)),
          "tsdf_selftest_tile_tables")
    return cnt.value


class Volume:
    """One z-slab of a TSDF grid in HBM: thin object wrapper over the opaque C handle."""

    def __init__(self, cfg, _borrowed_handle=None):
        self.lib = load()
        self.cfg = cfg
        self._owned = _borrowed_handle is None
        self._h = C.c_void_p()
        if self._owned:
            check(self.lib.tsdf_create(C.byref(cfg), C.byref(self._h)), "tsdf_create")
        else:
            self._h = _borrowed_handle

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if self._h:
            if self._owned:
                self.lib.tsdf_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- properties -------------------------------------------------------------------------
    @property
    def n_voxels(self):
        return int(self.lib.tsdf_slab_voxels(self._h))

    def shortcut_stats(self, enable=True):
        """(per-voxel, free-space, skipped) wavefront-frame counts since the counters were enabled; then (re)arm or stop."""
        out = (C.c_uint64 * 3)()
        check(self.lib.tsdf_shortcut_stats(self._h, 1 if enable else 0, out), "tsdf_shortcut_stats")
        return tuple(int(x) for x in out)

    def brick_list_stats(self):
        """(super-bricks every frame skipped, bricks on the work list, of those left to classify themselves, of those skipped
        after all) since shortcut_stats(True)."""
        out = (C.c_uint64 * 4)()
        check(self.lib.tsdf_brick_list_stats(self._h, out), "tsdf_brick_list_stats")
        return tuple(int(x) for x in out)

    def classification_info(self):
        """(fraction of workgroup-frames the last counted launch claimed or -1, launches since one classified)."""
        out = (C.c_double * 2)()
        check(self.lib.tsdf_classification_info(self._h, out), "tsdf_classification_info")
        return float(out[0]), int(out[1])

    @property
    def frames_per_launch(self):
        """Frames a sequence call applies per pass over this slab (1 when the kernel variant does not fuse)."""
        return int(self.lib.tsdf_frames_per_launch(self._h))

    @property
    def slab_shape(self):
        c = self.cfg
        return (c.z_end - c.z_begin, c.dim_y, c.dim_x)

    def device_ptrs(self):
        t, w = C.c_void_p(), C.c_void_p()
        check(self.lib.tsdf_device_ptrs(self._h, C.byref(t), C.byref(w)), "tsdf_device_ptrs")
        return t.value, w.value

    # -- integrate --------------------------------------------------------------------------
    def integrate(self, depth_host, cam2world):
        """TSDF::Integrate semantics: host depth (borrowed for the call), camera pose."""
        d = _f32(depth_host, self.cfg.im_height * self.cfg.im_width)
        p = _f32(cam2world, 16)
        check(self.lib.tsdf_integrate(self._h, d.ctypes.data, p.ctypes.data), "tsdf_integrate")

    def set_deferral(self, n_frames):
        """Host frames collected per launch by integrate() (0 / 1: every call launches; default 32)."""
        check(self.lib.tsdf_set_deferral(self._h, n_frames), "tsdf_set_deferral")

    def integrate_u16(self, raw_u16, cam2world, depth_factor=5000.0, row_step=1, col_step=1):
        """Raw 16-bit frame: half-size H2D copy, conversion (and optional subsampling) on the device."""
        r = np.ascontiguousarray(raw_u16, dtype=np.uint16).ravel()
        if r.size != self.cfg.im_height * self.cfg.im_width:
            raise ValueError("raw frame size does not match the configured image")
        p = _f32(cam2world, 16)
        check(self.lib.tsdf_integrate_u16(self._h, r.ctypes.data, depth_factor, row_step, col_step, p.ctypes.data),
              "tsdf_integrate_u16")

    def convert_depth_u16(self, raw_ptr, depth_ptr, depth_factor=5000.0, row_step=1, col_step=1):
        check(self.lib.tsdf_convert_depth_u16(self._h, raw_ptr, depth_ptr, depth_factor, row_step, col_step),
              "tsdf_convert_depth_u16")

    def integrate_device(self, depth_ptr, cam2world):
        p = _f32(cam2world, 16)
        check(self.lib.tsdf_integrate_device(self._h, depth_ptr, p.ctypes.data), "tsdf_integrate_device")

    def integrate_cam2base(self, depth_ptr, cam2base):
        p = _f32(cam2base, 16)
        check(self.lib.tsdf_integrate_cam2base(self._h, depth_ptr, p.ctypes.data), "tsdf_integrate_cam2base")

    def integrate_masked_device(self, depth_ptr, mask_ptr, cam2world):
        p = _f32(cam2world, 16)
        check(self.lib.tsdf_integrate_masked_device(self._h, depth_ptr, mask_ptr, p.ctypes.data),
              "tsdf_integrate_masked_device")

    def integrate_frames_device(self, depth_ptrs, poses, mask_ptrs=None):
        """A known sequence of frames == that many integrate_device calls, possibly fused per launch."""
        p = _f32(np.asarray(poses, dtype=np.float32))
        n = p.size // 16
        assert len(depth_ptrs) == n
        d = (C.c_void_p * n)(*[C.c_void_p(x) for x in depth_ptrs])
        m = None
        if mask_ptrs is not None:
            m = (C.c_void_p * n)(*[C.c_void_p(x) if x else C.c_void_p() for x in mask_ptrs])
        check(self.lib.tsdf_integrate_frames_device(self._h, d, m, p.ctypes.data, n), "tsdf_integrate_frames_device")

    def integrate_sequence_timed(self, depth_ptr, poses):
        """Queue len(poses) frames back to back; returns device milliseconds (HIP events)."""
        p = _f32(np.asarray(poses, dtype=np.float32))
        n = p.size // 16
        ms = C.c_float()
        check(self.lib.tsdf_integrate_sequence_timed(self._h, depth_ptr, p.ctypes.data, n, C.byref(ms)),
              "tsdf_integrate_sequence_timed")
        return ms.value

    def integrate_frames_timed(self, depth_ptrs, poses, mask_ptrs=None):
        """integrate_frames_device bracketed by HIP events on the handle's stream; returns device milliseconds."""
        p = _f32(np.asarray(poses, dtype=np.float32))
        n = p.size // 16
        assert len(depth_ptrs) == n
        d = (C.c_void_p * n)(*[C.c_void_p(x) for x in depth_ptrs])
        m = None
        if mask_ptrs is not None:
            m = (C.c_void_p * n)(*[C.c_void_p(x) if x else C.c_void_p() for x in mask_ptrs])
        ms = C.c_float()
        check(self.lib.tsdf_integrate_frames_timed(self._h, d, m, p.ctypes.data, n, C.byref(ms)),
              "tsdf_integrate_frames_timed")
        return ms.value

    def frames_timed_call(self, depth_ptrs, poses, mask_ptrs=None):
        """integrate_frames_timed with the argument marshalling done ahead of time: returns a callable that queues the
        sequence and returns its device milliseconds (for timed regions that must not include building 10 000 pointers)."""
        p = _f32(np.asarray(poses, dtype=np.float32))
        n = p.size // 16
        assert len(depth_ptrs) == n
        d = (C.c_void_p * n)(*depth_ptrs)
        m = None
        if mask_ptrs is not None:
            m = (C.c_void_p * n)(*[x if x else None for x in mask_ptrs])
        ms = C.c_float()

        def call():
            check(self.lib.tsdf_integrate_frames_timed(self._h, d, m, p.ctypes.data, n, C.byref(ms)), "tsdf_integrate_frames_timed")
            return ms.value
        return call

    def probe_graph_replay(self, depth_ptr, poses, iters=20):
        """(ms per repetition queued call by call, ms per repetition replayed from a captured hipGraph)."""
        p = _f32(np.asarray(poses, dtype=np.float32))
        a, b = C.c_float(), C.c_float()
        check(self.lib.tsdf_probe_graph_replay(self._h, depth_ptr, p.ctypes.data, p.size // 16, iters, C.byref(a), C.byref(b)),
              "tsdf_probe_graph_replay")
        return a.value, b.value

    def probe_stream(self, non_temporal=False, iters=20):
        """Bare RMW stream over the slab (ceiling probe); returns milliseconds per pass."""
        ms = C.c_float()
        check(self.lib.tsdf_probe_stream(self._h, int(non_temporal), iters, C.byref(ms)), "tsdf_probe_stream")
        return ms.value / iters

    def last_cam2base(self):
        out = np.empty(16, np.float32)
        check(self.lib.tsdf_last_cam2base(self._h, out.ctypes.data), "tsdf_last_cam2base")
        return out

    # -- state ------------------------------------------------------------------------------
    def sync(self):
        check(self.lib.tsdf_sync(self._h), "tsdf_sync")

    def reset(self):
        check(self.lib.tsdf_reset(self._h), "tsdf_reset")

    def download(self):
        n = self.n_voxels
        t, w = np.empty(n, np.float32), np.empty(n, np.float32)
        check(self.lib.tsdf_download(self._h, t.ctypes.data, w.ctypes.data), "tsdf_download")
        return t, w

    def copy_slices(self, z_local, n_slices):
        """Host copies of n_slices whole z-slices starting at slab-local z_local."""
        n = n_slices * self.cfg.dim_x * self.cfg.dim_y
        t, w = np.empty(n, np.float32), np.empty(n, np.float32)
        check(self.lib.tsdf_copy_slices(self._h, z_local, n_slices, t.ctypes.data, w.ctypes.data), "tsdf_copy_slices")
        return t, w

    def copy_slices_to_device(self, z_local, n_slices, tsdf_ptr, weight_ptr):
        """Same, into device buffers (e.g. an RCCL send buffer): no host hop."""
        check(self.lib.tsdf_copy_slices(self._h, z_local, n_slices, tsdf_ptr, weight_ptr), "tsdf_copy_slices")

    def upload(self, tsdf, weight):
        t, w = _f32(tsdf, self.n_voxels), _f32(weight, self.n_voxels)
        check(self.lib.tsdf_upload(self._h, t.ctypes.data, w.ctypes.data), "tsdf_upload")

    def set_stream(self, stream_ptr):
        check(self.lib.tsdf_set_stream(self._h, stream_ptr), "tsdf_set_stream")

    def set_kernel_variant(self, v):
        check(self.lib.tsdf_set_kernel_variant(self._h, v), "tsdf_set_kernel_variant")

    def set_brick_shape(self, quads=0, rows=0, slices=0):
        """The box a wavefront owns in classified launches (tuning only; (0, 0, 0) = the library's choice)."""
        check(self.lib.tsdf_set_brick_shape(self._h, quads, rows, slices), "tsdf_set_brick_shape")

    def brick_shape(self):
        out = (C.c_int32 * 3)()
        check(self.lib.tsdf_brick_shape(self._h, out), "tsdf_brick_shape")
        return tuple(out)

    # -- per-voxel label fusion ----------------------------------------------------------------
    def labels_enable(self, prob_threshold=0.5):
        check(self.lib.tsdf_labels_enable(self._h, prob_threshold), "tsdf_labels_enable")

    def compose_labels(self, masks_ptr, labels, scores, label_im_ptr, score_im_ptr):
        lab = np.ascontiguousarray(labels, np.uint16)
        sc = _f32(scores)
        check(self.lib.tsdf_compose_labels(self._h, masks_ptr, lab.ctypes.data, sc.ctypes.data, lab.size,
                                           label_im_ptr, score_im_ptr), "tsdf_compose_labels")

    def integrate_labels_device(self, depth_ptr, label_im_ptr, score_im_ptr, cam2world):
        p = _f32(cam2world, 16)
        check(self.lib.tsdf_integrate_labels_device(self._h, depth_ptr, label_im_ptr, score_im_ptr, p.ctypes.data),
              "tsdf_integrate_labels_device")

    def integrate_frames_labels_device(self, depth_ptrs, label_im_ptrs, score_im_ptrs, poses):
        """Integrate + label fusion of a known sequence in the same passes == integrate_device + integrate_labels_device
        per frame."""
        p = _f32(np.asarray(poses, dtype=np.float32))
        n = p.size // 16
        assert len(depth_ptrs) == len(label_im_ptrs) == len(score_im_ptrs) == n
        arr = lambda xs: (C.c_void_p * n)(*[C.c_void_p(x) for x in xs])
        check(self.lib.tsdf_integrate_frames_labels_device(self._h, arr(depth_ptrs), arr(label_im_ptrs), arr(score_im_ptrs),
                                                           p.ctypes.data, n), "tsdf_integrate_frames_labels_device")

    def download_labels(self):
        n = self.n_voxels
        lab, fp, bp = np.empty(n, np.uint16), np.empty(n, np.float32), np.empty(n, np.float32)
        check(self.lib.tsdf_download_labels(self._h, lab.ctypes.data, fp.ctypes.data, bp.ctypes.data), "tsdf_download_labels")
        return lab, fp, bp

    # -- per-voxel colour fusion ----------------------------------------------------------------
    def colour_enable(self):
        check(self.lib.tsdf_colour_enable(self._h), "tsdf_colour_enable")

    def integrate_colour_device(self, depth_ptr, rgb_ptr, cam2world):
        """Colour pass of the frame just integrated (queue right after integrate*_device of the same frame)."""
        p = _f32(cam2world, 16)
        check(self.lib.tsdf_integrate_colour_device(self._h, depth_ptr, rgb_ptr, p.ctypes.data), "tsdf_integrate_colour_device")

    def integrate_rgbd(self, depth_host, rgb_host, cam2world):
        d = _f32(depth_host, self.cfg.im_height * self.cfg.im_width)
        c = np.ascontiguousarray(rgb_host, dtype=np.uint8).ravel()
        if c.size != 3 * self.cfg.im_height * self.cfg.im_width:
            raise ValueError("colour image must be height x width x 3 bytes")
        p = _f32(cam2world, 16)
        check(self.lib.tsdf_integrate_rgbd(self._h, d.ctypes.data, c.ctypes.data, p.ctypes.data), "tsdf_integrate_rgbd")

    def download_colour(self):
        out = np.empty(self.n_voxels, np.uint32)
        check(self.lib.tsdf_download_colour(self._h, out.ctypes.data), "tsdf_download_colour")
        return out

    # -- outputs ----------------------------------------------------------------------------
    def count_surface(self, weight_thresh=0.9):
        n = C.c_int64()
        check(self.lib.tsdf_count_surface(self._h, weight_thresh, C.byref(n)), "tsdf_count_surface")
        return n.value

    def extract_surface(self, weight_thresh=0.9):
        n = self.count_surface(weight_thresh)
        xyz = np.empty((n, 3), np.float32)
        got = C.c_int64()
        check(self.lib.tsdf_extract_surface(self._h, weight_thresh, xyz.ctypes.data, n, C.byref(got)),
              "tsdf_extract_surface")
        assert got.value == n
        return xyz

    def extract_crossings(self, halo=None, weight_thresh=0.9):
        """Zero-crossing vertices of the slab; halo = (tsdf, weight) of slice z_end as numpy arrays,
        or a pair of device pointers (ints), or None on the top slab."""
        ht = hw = None
        keep = None
        if halo is not None:
            if isinstance(halo[0], (int, np.integer)):
                ht, hw = int(halo[0]), int(halo[1])
            else:
                keep = (_f32(halo[0]), _f32(halo[1]))
                ht, hw = keep[0].ctypes.data, keep[1].ctypes.data
        n = C.c_int64()
        check(self.lib.tsdf_extract_crossings(self._h, ht, hw, weight_thresh, None, 0, C.byref(n)), "tsdf_extract_crossings")
        xyz = np.empty((n.value, 3), np.float32)
        if n.value:
            got = C.c_int64()
            check(self.lib.tsdf_extract_crossings(self._h, ht, hw, weight_thresh, xyz.ctypes.data, n.value, C.byref(got)),
                  "tsdf_extract_crossings")
            assert got.value == n.value
        return xyz

    def extract_mesh(self, halo=None, weight_thresh=0.9):
        """Marching-tetrahedra triangles of the slab, array [n, 3, 3]; halo as for extract_crossings."""
        ht = hw = None
        keep = None
        if halo is not None:
            if isinstance(halo[0], (int, np.integer)):
                ht, hw = int(halo[0]), int(halo[1])
            else:
                keep = (_f32(halo[0]), _f32(halo[1]))
                ht, hw = keep[0].ctypes.data, keep[1].ctypes.data
        n = C.c_int64()
        check(self.lib.tsdf_extract_mesh(self._h, ht, hw, weight_thresh, None, 0, C.byref(n)), "tsdf_extract_mesh")
        tri = np.empty((n.value, 3, 3), np.float32)
        if n.value:
            got = C.c_int64()
            check(self.lib.tsdf_extract_mesh(self._h, ht, hw, weight_thresh, tri.ctypes.data, n.value, C.byref(got)),
                  "tsdf_extract_mesh")
            assert got.value == n.value
        return tri

    def save_mesh_ply(self, path, weight_thresh=0.9):
        check(self.lib.tsdf_save_mesh_ply(self._h, os.fsencode(path), weight_thresh), "tsdf_save_mesh_ply")

    def save_mesh_welded_ply(self, path, weight_thresh=0.9):
        check(self.lib.tsdf_save_mesh_welded_ply(self._h, os.fsencode(path), weight_thresh), "tsdf_save_mesh_welded_ply")

    def save_ply(self, path, weight_thresh=0.9):
        check(self.lib.tsdf_save_ply(self._h, os.fsencode(path), weight_thresh), "tsdf_save_ply")

    def save_bin(self, path):
        check(self.lib.tsdf_save_bin(self._h, os.fsencode(path)), "tsdf_save_bin")

    def load_bin(self, path):
        check(self.lib.tsdf_load_bin(self._h, os.fsencode(path)), "tsdf_load_bin")

    def save_state(self, path):
        check(self.lib.tsdf_save_state(self._h, os.fsencode(path)), "tsdf_save_state")

    def load_state(self, path):
        check(self.lib.tsdf_load_state(self._h, os.fsencode(path)), "tsdf_load_state")


class Batch:
    """n per-object volumes integrated by one launch per frame (tsdf_batch_*)."""

    def __init__(self, cfgs):
        self.lib = load()
        self.cfgs = list(cfgs)
        arr = (TsdfConfig * len(self.cfgs))(*self.cfgs)
        self._h = C.c_void_p()
        check(self.lib.tsdf_batch_create(arr, len(self.cfgs), C.byref(self._h)), "tsdf_batch_create")
        self.volumes = []
        for i, cfg in enumerate(self.cfgs):
            h = C.c_void_p()
            check(self.lib.tsdf_batch_volume(self._h, i, C.byref(h)), "tsdf_batch_volume")
            self.volumes.append(Volume(cfg, _borrowed_handle=h))

    def integrate_device(self, depth_ptr, mask_ptrs, cam2world):
        """mask_ptrs: list of device pointers (or None entries), or None for no masks at all."""
        p = _f32(cam2world, 16)
        masks = None
        if mask_ptrs is not None:
            masks = (C.c_void_p * len(self.cfgs))(*[C.c_void_p(m) if m else C.c_void_p() for m in mask_ptrs])
        check(self.lib.tsdf_batch_integrate_device(self._h, depth_ptr, masks, p.ctypes.data), "tsdf_batch_integrate_device")

    def sync(self):
        check(self.lib.tsdf_batch_sync(self._h), "tsdf_batch_sync")

    def close(self):
        if self._h:
            for v in self.volumes:
                v.close()
            self.lib.tsdf_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Group:
    """One grid cut into z-slabs over several devices in one process (tsdf_group_*)."""

    def __init__(self, cfg, devices):
        self.lib = load()
        self.cfg = cfg
        self.devices = [int(d) for d in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        check(self.lib.tsdf_group_create(C.byref(cfg), arr, len(self.devices), C.byref(self._h)), "tsdf_group_create")
        self.slabs = []
        for i in range(len(self.devices)):
            h = C.c_void_p()
            check(self.lib.tsdf_group_volume(self._h, i, C.byref(h)), "tsdf_group_volume")
            c = TsdfConfig()
            check(self.lib.tsdf_get_config(h, C.byref(c)), "tsdf_get_config")
            self.slabs.append(Volume(c, _borrowed_handle=h))

    @property
    def n_voxels(self):
        return int(self.lib.tsdf_group_voxels(self._h))

    def integrate(self, depth_host, cam2world):
        d = _f32(depth_host, self.cfg.im_height * self.cfg.im_width)
        p = _f32(cam2world, 16)
        check(self.lib.tsdf_group_integrate(self._h, d.ctypes.data, p.ctypes.data), "tsdf_group_integrate")

    def integrate_frames(self, depths_host, poses):
        keep = [_f32(d, self.cfg.im_height * self.cfg.im_width) for d in depths_host]
        p = _f32(np.asarray(poses, dtype=np.float32))
        n = p.size // 16
        assert len(keep) == n
        ptrs = (C.c_void_p * n)(*[C.c_void_p(d.ctypes.data) for d in keep])
        check(self.lib.tsdf_group_integrate_frames(self._h, ptrs, p.ctypes.data, n), "tsdf_group_integrate_frames")

    def set_deferral(self, n_frames):
        check(self.lib.tsdf_group_set_deferral(self._h, n_frames), "tsdf_group_set_deferral")

    def sync(self):
        check(self.lib.tsdf_group_sync(self._h), "tsdf_group_sync")

    def reset(self):
        check(self.lib.tsdf_group_reset(self._h), "tsdf_group_reset")

    def download(self):
        n = self.n_voxels
        t, w = np.empty(n, np.float32), np.empty(n, np.float32)
        check(self.lib.tsdf_group_download(self._h, t.ctypes.data, w.ctypes.data), "tsdf_group_download")
        return t, w

    def _list(self, fn, shape, weight_thresh):
        n = C.c_int64()
        check(fn(self._h, weight_thresh, None, 0, C.byref(n)), "tsdf_group_extract")
        out = np.empty((n.value,) + shape, np.float32)
        if n.value:
            got = C.c_int64()
            check(fn(self._h, weight_thresh, out.ctypes.data, n.value, C.byref(got)), "tsdf_group_extract")
            assert got.value == n.value
        return out

    def extract_surface(self, weight_thresh=0.9):
        return self._list(self.lib.tsdf_group_extract_surface, (3,), weight_thresh)

    def extract_crossings(self, weight_thresh=0.9):
        return self._list(self.lib.tsdf_group_extract_crossings, (3,), weight_thresh)

    def extract_mesh(self, weight_thresh=0.9):
        return self._list(self.lib.tsdf_group_extract_mesh, (3, 3), weight_thresh)

    def save_ply(self, path, weight_thresh=0.9):
        check(self.lib.tsdf_group_save_ply(self._h, os.fsencode(path), weight_thresh), "tsdf_group_save_ply")

    def save_mesh_ply(self, path, weight_thresh=0.9):
        check(self.lib.tsdf_group_save_mesh_ply(self._h, os.fsencode(path), weight_thresh), "tsdf_group_save_mesh_ply")

    def save_bin(self, path):
        check(self.lib.tsdf_group_save_bin(self._h, os.fsencode(path)), "tsdf_group_save_bin")

    def close(self):
        if self._h:
            for v in self.slabs:
                v.close()
            self.lib.tsdf_group_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
