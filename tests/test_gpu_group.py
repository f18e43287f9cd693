"""tsdf_group_*: one grid cut into z-slabs over several devices in ONE process (the C++ host's way to span the
node).  The GPU box has one card, so every slab handle lives on device 0 -- separate allocations, separate streams,
the same fan-out, peer-copy halo and gather code paths -- and everything must equal one handle holding the whole
grid, bit for bit and byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "semantic_slam_amd")


def scene_frames(dims, vs, origin, n):
    sc = synth.SurfScene(dims, vs, origin)
    return [(sc.pose(k % 7, 9), sc.depth(sc.pose(k % 7, 9), quantize=True)) for k in range(n)]


@pytest.mark.parametrize("dims,vs,n_slabs", [((256, 72, 43), 0.01, 2), ((200, 60, 31), 0.012, 3), ((64, 64, 5), 0.02, 5)])
def test_group_equals_whole_grid(cuda, oracle, tmp_path, dims, vs, n_slabs):
    origin = synth.surf_volume(dims[0], vs, 0.9)
    cfg = capi.make_config(dims, vs, origin, vol_id=3)
    frames = scene_frames(dims, vs, origin, 36)
    ref_t, ref_w = oracle.init_grid(dims)
    for pose, depth in frames:
        oracle.integrate(cfg.cam_K, pose, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, threads=8)
    with capi.Volume(cfg) as whole, capi.Group(cfg, [0] * n_slabs) as grp:
        assert [(-v.cfg.z_begin + v.cfg.z_end) for v in grp.slabs] == [(i + 1) * dims[2] // n_slabs - i * dims[2] // n_slabs
                                                                       for i in range(n_slabs)]
        for pose, depth in frames[:3]:               # frame by frame: pinned ring + fan-out
            whole.integrate(depth, pose)
            grp.integrate(depth, pose)
        devs = [cuda.from_numpy(d).cuda() for _, d in frames[3:]]
        whole.integrate_frames_device([d.data_ptr() for d in devs], np.stack([p for p, _ in frames[3:]]))
        grp.integrate_frames([d for _, d in frames[3:]], np.stack([p for p, _ in frames[3:]]))   # 33 frames: two passes
        grp.sync()
        t0, w0 = whole.download()
        t1, w1 = grp.download()
        assert np.array_equal(w0, ref_w) and np.array_equal(t0.view(np.uint32), ref_t.view(np.uint32))
        assert np.array_equal(w1, ref_w) and np.array_equal(t1.view(np.uint32), ref_t.view(np.uint32))
        for name in ("extract_surface", "extract_crossings", "extract_mesh"):
            a, b = getattr(whole, name)(), getattr(grp, name)()
            if name == "extract_surface":
                assert len(a) > 100
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), name
        for ext, fa, fb in (("ply", whole.save_ply, grp.save_ply), ("bin", whole.save_bin, grp.save_bin),
                            ("mesh.ply", whole.save_mesh_ply, grp.save_mesh_ply)):
            pa, pb = str(tmp_path / f"a.{ext}"), str(tmp_path / f"b.{ext}")
            fa(pa)
            fb(pb)
            assert open(pa, "rb").read() == open(pb, "rb").read(), ext
        # and the .bin / .ply are the oracle's (the reference's formats)
        oracle.save_bin(str(tmp_path / "o.bin"), ref_t, dims, origin, vs, cfg.trunc_margin)
        assert open(str(tmp_path / "o.bin"), "rb").read() == open(str(tmp_path / "b.bin"), "rb").read()
        oracle.save_ply(str(tmp_path / "o.ply"), ref_t, ref_w, dims, vs, origin)
        assert open(str(tmp_path / "o.ply"), "rb").read() == open(str(tmp_path / "b.ply"), "rb").read()
        grp.reset()
        t2, w2 = grp.download()
        assert np.all(t2 == 1.0) and np.all(w2 == 0.0)


def test_group_rejects_bad_arguments(cuda):
    cfg = capi.make_config((16, 16, 4), 0.01, [0, 0, 1])
    with pytest.raises(capi.TsdfError, match="slabs"):
        capi.Group(cfg, [0] * 5)                     # more slabs than slices
    with pytest.raises(capi.TsdfError, match="device"):
        capi.Group(cfg, [0, 99])


def test_cpp_class_over_a_device_list(cuda, oracle, tmp_path):
    """`TSDF(cfg, devices)` from a C++ program: the reference's call sequence (ctor, Integrate per frame, delete) over
    three slabs and over one handle -- identical host mirrors after Download(), identical tsdf<id>.ply / .bin."""
    exe = str(tmp_path / "dropin_multidevice")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "dropin_multidevice.cpp"), "-o", exe,
                           "-L", PKG, "-ltsdf_dropin", "-ltsdf_hip", f"-Wl,-rpath,{PKG}"])
    dims, vs = (200, 96, 47), 0.006
    origin = synth.surf_volume(dims[0], vs, 0.8)
    frames = scene_frames(dims, vs, origin, 4)
    inp = tmp_path / "frames.bin"
    with open(inp, "wb") as f:
        f.write(np.array(list(dims) + [len(frames)], np.int32).tobytes())
        f.write(np.array([vs], np.float32).tobytes())
        f.write(np.asarray(origin, np.float32).tobytes())
        for pose, depth in frames:
            f.write(np.asarray(pose, np.float32).tobytes())
            f.write(depth.astype(np.float32).tobytes())
    out = subprocess.check_output([exe, str(inp)], cwd=str(tmp_path)).decode()
    assert "mirrors identical" in out, out
    cfg = capi.make_config(dims, vs, origin)
    ref_t, ref_w = oracle.init_grid(dims)
    for pose, depth in frames:
        oracle.integrate(cfg.cam_K, pose, depth, dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, threads=8)
    for ext in ("ply", "bin"):
        a = open(str(tmp_path / f"tsdf1.{ext}"), "rb").read()     # single device
        b = open(str(tmp_path / f"tsdf2.{ext}"), "rb").read()     # three slabs
        assert a == b and len(a) > 1000, ext
    oracle.save_bin(str(tmp_path / "o.bin"), ref_t, dims, origin, vs, cfg.trunc_margin)
    assert open(str(tmp_path / "o.bin"), "rb").read() == open(str(tmp_path / "tsdf2.bin"), "rb").read()


def test_borrowed_slab_handle_sees_the_groups_collected_frames_in_call_order(cuda, oracle):
    """tsdf_group_integrate collects frames (deferred integration, 32 per pass).  A slab handle borrowed with
    tsdf_group_volume must see them: observing through it (download, extraction, device pointers) applies them first,
    and a frame integrated through it AFTER group frames is applied after them -- the float bits depend on the order."""
    dims, vs = (128, 48, 30), 0.012
    origin = synth.surf_volume(dims[0], vs, 0.8)
    cfg = capi.make_config(dims, vs, origin)
    frames = scene_frames(dims, vs, origin, 8)
    with capi.Group(cfg, [0, 0]) as grp:            # (capi.Group borrows every slab in its constructor, before any frame)
        lo = grp.slabs[0]
        zb, ze = lo.cfg.z_begin, lo.cfg.z_end
        rt, rw = oracle.init_grid(dims, zb, ze)

        def ref(k):
            oracle.integrate(cfg.cam_K, frames[k][0], frames[k][1], dims, origin, vs, cfg.trunc_margin, rt, rw, z_begin=zb, z_end=ze)

        for k in range(3):                           # collected by the group, not yet launched
            grp.integrate(frames[k][1], frames[k][0])
            ref(k)
        t, w = lo.download()                         # observation through the borrowed handle
        assert rw.max() >= 3
        assert np.array_equal(w, rw) and np.array_equal(t.view(np.uint32), rt.view(np.uint32))
        grp.integrate(frames[3][1], frames[3][0])    # group frame A (collected) ...
        ref(3)
        d_dev = cuda.from_numpy(frames[4][1]).cuda()
        lo.integrate_device(d_dev.data_ptr(), frames[4][0])   # ... then frame B through the slab handle: A first, then B
        ref(4)
        grp.integrate(frames[5][1], frames[5][0])    # ... then group frame C: B (collected by the slab) first, then C
        ref(5)
        assert lo.count_surface() > 100              # an extraction through the handle applies all three
        t, w = lo.download()
        assert np.array_equal(w, rw) and np.array_equal(t.view(np.uint32), rt.view(np.uint32))
        # the other slab saw the group's frames only (0, 1, 2, 3, 5)
        hi = grp.slabs[1]
        ht, hw = oracle.init_grid(dims, hi.cfg.z_begin, hi.cfg.z_end)
        for k in (0, 1, 2, 3, 5):
            oracle.integrate(cfg.cam_K, frames[k][0], frames[k][1], dims, origin, vs, cfg.trunc_margin, ht, hw,
                             z_begin=hi.cfg.z_begin, z_end=hi.cfg.z_end)
        t, w = hi.download()
        assert np.array_equal(w, hw) and np.array_equal(t.view(np.uint32), ht.view(np.uint32))
