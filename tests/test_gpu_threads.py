"""Distinct handles driven from distinct host threads at the same time -- the reference's caller loops over its objects
under an (inert) `omp parallel for` (ref: src/Engine.cpp:170-172), one TSDF per object, never sharing one.  Each thread
creates its own volume, integrates its own masked frames (host frames: deferred, fused; device frames: per launch) and
reads it back; every result must equal the oracle's for that object.  ctypes releases the GIL for the duration of a call,
so the library's entry points really do run concurrently."""
import threading

import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_distinct_handles_from_distinct_threads(cuda, oracle, tmp_path):
    dims, vs = (200, 96, 64), 0.004
    n_threads, n_frames = 6, 5
    scene = synth.SurfScene((200, 200, 200), 0.004, np.array([-0.4, -0.4, 0.7], np.float32))
    poses = [scene.pose(k, 7) for k in range(n_frames)]
    depths = [scene.depth(p, quantize=True) for p in poses]
    jobs = []
    rng = np.random.default_rng(8)
    for i in range(n_threads):
        origin = np.array([-0.4 + 0.05 * i, -0.2 + 0.02 * i, 0.75 + 0.04 * i], np.float32)
        mask = np.zeros((480, 640), np.uint8)
        r0, c0 = int(rng.integers(40, 160)), int(rng.integers(60, 240))
        mask[r0:r0 + 260, c0:c0 + 330] = 255
        cfg = capi.make_config(dims, vs, origin, vol_id=i)
        ref_t, ref_w = oracle.init_grid(dims)
        for p, d in zip(poses, depths):
            oracle.integrate(cfg.cam_K, p, oracle.mask_depth(d, mask), dims, origin, vs, cfg.trunc_margin, ref_t, ref_w, threads=4)
        jobs.append((cfg, mask, ref_t, ref_w))
    results, errors = [None] * n_threads, []
    start = threading.Barrier(n_threads)

    def work(i):
        try:
            cfg, mask, _, _ = jobs[i]
            cuda.cuda.set_device(0)
            start.wait()
            with capi.Volume(cfg) as vol:
                if i % 3 == 0:        # host frames (the reference's call shape): masked on the host, collected, fused
                    for p, d in zip(poses, depths):
                        vol.integrate(oracle.mask_depth(d, mask), p)
                elif i % 3 == 1:      # device frames + device mask, deferred
                    m_dev = cuda.from_numpy(mask).cuda()
                    keep = [cuda.from_numpy(d).cuda() for d in depths]
                    for p, d in zip(poses, keep):
                        vol.integrate_masked_device(d.data_ptr(), m_dev.data_ptr(), p)
                else:                 # one kernel per call, classification forced on
                    vol.set_deferral(0)
                    vol.set_kernel_variant(8)
                    m_dev = cuda.from_numpy(mask).cuda()
                    keep = [cuda.from_numpy(d).cuda() for d in depths]
                    for p, d in zip(poses, keep):
                        vol.integrate_masked_device(d.data_ptr(), m_dev.data_ptr(), p)
                results[i] = vol.download()
                # ... and writes its two files as the object's destructor does (the writers share two staging buffers)
                vol.save_bin(str(tmp_path / f"tsdf{i}.bin"))
                vol.save_ply(str(tmp_path / f"tsdf{i}.ply"))
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for i, (cfg, _, ref_t, ref_w) in enumerate(jobs):
        t, w = results[i]
        assert np.array_equal(w, ref_w), f"thread {i}: weights differ"
        assert np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)), f"thread {i}: TSDF differs"
        assert ref_w.sum() > 1000
        oracle.save_bin(str(tmp_path / "want.bin"), ref_t, dims, cfg.origin, vs, cfg.trunc_margin)
        oracle.save_ply(str(tmp_path / "want.ply"), ref_t, ref_w, dims, vs, cfg.origin)
        assert (tmp_path / f"tsdf{i}.bin").read_bytes() == (tmp_path / "want.bin").read_bytes(), f"thread {i}: .bin differs"
        assert (tmp_path / f"tsdf{i}.ply").read_bytes() == (tmp_path / "want.ply").read_bytes(), f"thread {i}: .ply differs"
