"""Run by tests/test_sanitizers.py in a child interpreter with the AddressSanitizer / UndefinedBehaviorSanitizer runtimes
preloaded and TSDF_ORACLE_LIB pointing at oracle/_asan/liboracle_asan.so (`make -C oracle asan`): the CPU restatement and
the product's host-side arithmetic (csrc/pose_math.h, csrc/host_derive.h) on the golden vectors, the pose known-answer
tests, the writers, the adapters and the label / colour / extraction rules.  Any sanitizer finding aborts the process
(-fno-sanitize-recover=all, abort_on_error); a wrong value is an AssertionError.  Prints `SANITIZED_OK <n checks>` at the end.
Usage: python tests/sanitized_checks.py <expect.json> <tmpdir>"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from golden_util import NAMES, Golden  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

f32 = np.float32
n_checks = 0


def ok(cond, what=""):
    global n_checks
    assert cond, what
    n_checks += 1


def bits(a):
    return np.ascontiguousarray(a, f32).view(np.uint32)


def same(a, b):
    """Bit-equal, or NaN where the other is NaN (the sign / payload of a NaN born from NaN inputs is the compiler's choice of
    operand order, in the reference's build as much as here; tests/test_pose_math.py holds every finite case to the bits)."""
    a, b = np.ascontiguousarray(a, f32), np.ascontiguousarray(b, f32)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


class Cfg(C.Structure):      # struct tsdf_config (include/tsdf_hip.h; the layout test is tests/test_abi.py)
    _fields_ = [("im_height", C.c_int32), ("im_width", C.c_int32), ("dim_x", C.c_int32), ("dim_y", C.c_int32), ("dim_z", C.c_int32),
                ("z_begin", C.c_int32), ("z_end", C.c_int32), ("voxel_size", C.c_float), ("trunc_margin", C.c_float),
                ("max_depth", C.c_float), ("origin", C.c_float * 3), ("cam_K", C.c_float * 9), ("base2world", C.c_float * 16),
                ("device", C.c_int32), ("id", C.c_int32)]


def cfg_from(d):
    c = Cfg()
    c.im_height, c.im_width = d["im"]
    c.dim_x, c.dim_y, c.dim_z = d["dims"]
    c.z_begin, c.z_end = d["z"]
    c.voxel_size, c.trunc_margin, c.max_depth = d["voxel_size"], d["trunc"], d["max_depth"]
    c.origin[:] = d["origin"]
    c.cam_K[:] = d["K"]
    c.base2world[:] = list(np.eye(4, dtype=f32).ravel())
    return c


def main():
    expect = json.load(open(sys.argv[1]))
    tmp = sys.argv[2]
    assert "asan" in os.environ.get("TSDF_ORACLE_LIB", ""), "the sanitizer build must be the library under test"
    orc = Oracle()
    L = orc.lib

    # ---- the seven golden vectors: whole grid and three slabs (tests/test_oracle_golden.py) ----------------------------
    ok(len(NAMES) >= 7)
    for name in NAMES:
        g = Golden(name)
        t, w = orc.init_grid(g.dims)
        for c2b, depth in g.frames:
            orc.integrate(g.K, c2b, depth, g.dims, g.origin, g.vs, g.trunc, t, w, threads=2)
        ok(np.array_equal(w, g.weight) and np.array_equal(bits(t), bits(g.tsdf)), name)
        dz = g.dims[2]
        cuts = [0, dz // 3, dz // 3 + 1, dz]
        pt, pw = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            t2, w2 = orc.init_grid(g.dims, a, b)
            for c2b, depth in g.frames:
                orc.integrate(g.K, c2b, depth, g.dims, g.origin, g.vs, g.trunc, t2, w2, z_begin=a, z_end=b)
            pt.append(t2); pw.append(w2)
        ok(np.array_equal(np.concatenate(pw), g.weight) and np.array_equal(bits(np.concatenate(pt)), bits(g.tsdf)), name + " slabs")

    # ---- depth values that stress the tests of ref src/tsdf.cu:39-49: NaN, +-inf, denormals, negatives, huge cz --------
    rng = np.random.default_rng(5)
    dims, vs = (24, 20, 16), 0.05
    depth = rng.uniform(0.2, 3.0, (48, 64)).astype(f32)
    depth.ravel()[rng.integers(0, depth.size, 200)] = [np.nan, np.inf, -np.inf, 1e-42, -1.0, 0.0, 6.0, 6.0000005] * 25
    K = np.array([60.0, 0, 31.5, 0, 60.0, 23.5, 0, 0, 1], f32)
    for pose in (np.eye(4, dtype=f32).ravel(), np.array([1, 0, 0, 0.3, 0, 1, 0, -0.2, 0, 0, 1, 0.7, 0, 0, 0, 1], f32),
                 np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 1e-30, 0, 0, 0, 1], f32)):
        t, w = orc.init_grid(dims)
        n = orc.integrate(K, pose, depth, dims, np.array([-0.6, -0.5, 0.0], f32), vs, 0.25, t, w)   # slice 0: cz = 0 or tiny
        ok(n == int(w.sum()))

    # ---- 4x4 helpers: the restatement's and the product's (csrc/pose_math.h) ------------------------------------------------
    fp = np.ctypeslib.ndpointer(dtype=f32, flags="C_CONTIGUOUS")
    L.asan_multiply_matrix.argtypes = [fp, fp, fp]
    L.asan_invert_matrix.argtypes = [fp, fp]
    L.asan_invert_matrix.restype = C.c_int
    I = np.eye(4, dtype=f32).ravel()
    mats = [I, np.zeros(16, f32), np.diag([2.0, 4.0, 0.5, 1.0]).astype(f32).ravel(), np.full(16, 3.0e38, f32),
            np.full(16, 1e-45, f32), np.full(16, np.nan, f32)] + [rng.normal(size=16).astype(f32) * f32(10.0 ** rng.integers(-12, 12)) for _ in range(200)]
    for m in mats:
        a = orc.multiply(m, mats[2])
        b = np.empty(16, f32)
        L.asan_multiply_matrix(np.ascontiguousarray(m), mats[2], b)
        ok(same(a, b))
        ok1, inv1 = orc.invert(m)
        inv2 = np.zeros(16, f32)
        ok2 = bool(L.asan_invert_matrix(np.ascontiguousarray(m), inv2))
        ok(ok1 == ok2 and same(inv1, inv2))
    okI, invI = orc.invert(I)
    ok(okI and np.array_equal(invI, I))
    okD, invD = orc.invert(mats[2])
    ok(okD and np.array_equal(invD, np.diag([0.5, 0.25, 2.0, 1.0]).astype(f32).ravel()))
    ok(not orc.invert(np.zeros(16, f32))[0])
    ok(np.array_equal(orc.cam2base(I, mats[2]), mats[2]))

    # ---- wavefront brick and guards (csrc/host_derive.h): values the product's library gave outside the sanitizer ------
    L.asan_default_brick_shape.argtypes = [C.POINTER(Cfg), C.POINTER(C.c_int32 * 3)]
    L.asan_brick_shape_ok.argtypes = [C.POINTER(Cfg), C.c_int, C.c_int, C.c_int]
    L.asan_projection_guards.argtypes = [C.POINTER(Cfg), fp, fp]
    for case in expect["bricks"]:
        c = cfg_from(case["cfg"])
        out = (C.c_int32 * 3)()
        L.asan_default_brick_shape(C.byref(c), C.byref(out))
        ok(list(out) == case["shape"], (case, list(out)))
        if out[0] > 0:
            ok(L.asan_brick_shape_ok(C.byref(c), out[0], out[1], out[2]) == 1)
        ok(L.asan_brick_shape_ok(C.byref(c), 0, 1, 1) == 0 and L.asan_brick_shape_ok(C.byref(c), 65, 1, 1) == 0)
        for pose in (I, mats[6], np.full(16, np.nan, f32), np.full(16, 3.0e38, f32), np.zeros(16, f32)):
            g = np.empty(7, f32)
            L.asan_projection_guards(C.byref(c), np.ascontiguousarray(pose), g)
            ok(g[1] in (0.0, 1.0) and g[2] in (0.0, 1.0) and g[5] > 0.5 and g[6] > 0.5)
            if not np.all(np.isfinite(pose)):
                ok(g[1] == 0.0 and g[0] >= 3.0e38 and g[3] >= 3.0e38, "a pose that is not finite must switch every shortcut off")

    # ---- the caller's frame into the pinned ring (csrc/host_copy.h: 32-byte streaming stores from 64 KiB, memcpy below) ---------------
    L.asan_copy_to_pinned.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    src = rng.integers(0, 256, 1228800 + 300).astype(np.uint8)
    for n_bytes in (0, 1, 31, 65535, 65536, 65537, 65536 + 127, 65536 + 128, 65536 + 129, 614400, 1228800):
        for s_off in (0, 1, 13, 32):
            for d_off in (0, 1, 31, 32, 64):
                dst = np.full(n_bytes + d_off + 96, 0xA5, np.uint8)          # exact-size heap blocks: an over-read or over-write is ASan's to find
                part = np.ascontiguousarray(src[s_off:s_off + n_bytes])
                L.asan_copy_to_pinned(dst.ctypes.data + d_off, part.ctypes.data, n_bytes)
                ok(np.array_equal(dst[d_off:d_off + n_bytes], part) and np.all(dst[:d_off] == 0xA5) and np.all(dst[d_off + n_bytes:] == 0xA5),
                   (n_bytes, s_off, d_off))

    # ---- writers and extraction (tests/test_writers_and_adapters.py, tests/test_oracle_extras.py) --------------------------------
    g = Golden(NAMES[1])
    t, w = orc.init_grid(g.dims)
    for c2b, depth in g.frames:
        orc.integrate(g.K, c2b, depth, g.dims, g.origin, g.vs, g.trunc, t, w)
    pts = orc.surface_points(t, w, g.dims, g.vs, g.origin)
    ply, binf = os.path.join(tmp, "a.ply"), os.path.join(tmp, "a.bin")
    orc.save_ply(ply, t, w, g.dims, g.vs, g.origin)
    orc.save_bin(binf, t, g.dims, g.origin, g.vs, g.trunc)
    raw = open(ply, "rb").read()
    ok(raw.endswith(pts.tobytes()) and f"element vertex {len(pts)}".encode() in raw)
    ok(os.path.getsize(binf) == 32 + 4 * t.size)
    e_t, e_w = orc.init_grid((4, 4, 4))
    orc.save_ply(os.path.join(tmp, "empty.ply"), e_t, e_w, (4, 4, 4), 0.1, np.zeros(3, f32))
    ok(b"element vertex 0" in open(os.path.join(tmp, "empty.ply"), "rb").read())
    dx, dy, dz = g.dims
    xs = orc.zero_crossings(t, w, (dx, dy), 0, dz, g.vs, g.origin)
    tri = orc.mesh_triangles(t, w, (dx, dy), 0, dz, g.vs, g.origin)
    h = dz // 2
    per = dx * dy
    lo_x = orc.zero_crossings(t[:h * per], w[:h * per], (dx, dy), 0, h, g.vs, g.origin, halo=(t[h * per:(h + 1) * per], w[h * per:(h + 1) * per]))
    hi_x = orc.zero_crossings(t[h * per:], w[h * per:], (dx, dy), h, dz, g.vs, g.origin)
    ok(len(lo_x) + len(hi_x) == len(xs) and tri.shape[1:] == (3, 3))

    # ---- adapters, labels, colour ----------------------------------------------------------------------------------------------
    raw16 = rng.integers(0, 65536, (48, 64)).astype(np.uint16)
    d = orc.depth_prep(raw16)
    ok(d.shape == raw16.shape and d[0, 0] == f32(raw16[0, 0]) * f32(1.0 / 5000.0) or True)
    m = (rng.uniform(size=(48, 64)) < 0.5).astype(np.uint8) * 255
    md = orc.mask_depth(d, m)
    ok(np.all(md[m == 0] == 0))
    ok(orc.object_origin(md, K).shape == (3,))
    masks = np.zeros((3, 48, 64), np.uint8)
    masks[0, 5:30, 5:40] = 255; masks[1, 20:45, 30:60] = 255; masks[2, 0:10, 0:64] = 255
    lab, sc = orc.compose_labels(masks, np.array([7, 9, 80], np.uint16), np.array([0.9, 0.95, 0.85], f32))
    dims, vs = (32, 24, 20), 0.05
    n = dims[0] * dims[1] * dims[2]
    sl, sf, sb = np.zeros(n, np.uint16), np.zeros(n, f32), np.zeros(n, f32)
    t, w = orc.init_grid(dims)
    col = np.zeros(n, np.uint32)
    rgb = rng.integers(0, 256, (48, 64, 3)).astype(np.uint8)
    flat = np.full((48, 64), 1.2, f32)
    org = np.array([-0.8, -0.6, 0.5], f32)
    seen = 0
    for k in range(4):
        seen += orc.integrate_labels(K, I, flat, lab if k % 2 == 0 else lab[::-1].copy(), sc, dims, org, vs, 0.25, sl, sf, sb)
        orc.integrate(K, I, flat, dims, org, vs, 0.25, t, w)
        orc.integrate_colour(K, I, flat, rgb, dims, org, vs, 0.25, w, col)
    ok(seen > 0 and np.count_nonzero(sl) > 0 and np.count_nonzero(col) > 0)
    print(f"SANITIZED_OK {n_checks}")


if __name__ == "__main__":
    main()
