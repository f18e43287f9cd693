"""The reference's own kernel, compiled for gfx950 by hipcc exactly as it stands and launched with its own
shape <<<dim_z, dim_y>>> (oracle/_ref/libtsdf_ref_hip.so, `make -C oracle ref_hip`: nothing substituted, see
oracle/Makefile), against the CPU restatement and the product kernels on the same inputs: the 7 golden
vectors and random grids / poses / depth images.  Bit-exact three ways -- this is what pins the oracle.

The .so is built in the build container (where /root/reference exists) and travels to the GPU box with the
snapshot; nothing here reads /root/reference."""
import numpy as np
import pytest

from golden_util import NAMES, Golden
from oracle.oracle import RefHip
from semantic_slam_amd import capi, synth

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not RefHip.available(), reason="oracle/_ref/libtsdf_ref_hip.so not built")]


def _ref_run(torch, ref, K, frames, dims, origin, vs, trunc):
    """frames: [(cam2base[16], depth[h, w])] -> (tsdf, weight) after the reference kernel saw them in order."""
    n = dims[0] * dims[1] * dims[2]
    t = torch.ones(n, dtype=torch.float32, device="cuda")        # ref: src/tsdf.cu:79-81
    w = torch.zeros(n, dtype=torch.float32, device="cuda")
    k_dev = torch.from_numpy(np.ascontiguousarray(K, np.float32)).cuda()
    for c2b, depth in frames:
        p_dev = torch.from_numpy(np.ascontiguousarray(c2b, np.float32)).cuda()
        d_dev = torch.from_numpy(np.ascontiguousarray(depth, np.float32)).cuda()
        torch.cuda.synchronize()
        ref.integrate(k_dev.data_ptr(), p_dev.data_ptr(), d_dev.data_ptr(), depth.shape[0], depth.shape[1], dims,
                      origin, vs, trunc, t.data_ptr(), w.data_ptr())
    return t.cpu().numpy(), w.cpu().numpy()


def _product_run(torch, K, frames, dims, origin, vs, trunc, h, w, variant, fused):
    cfg = capi.make_config(dims, vs, origin, trunc=trunc, K=K, im_height=h, im_width=w)
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(variant)
        devs = [torch.from_numpy(np.ascontiguousarray(d, np.float32)).cuda() for _, d in frames]
        if fused:   # base2world is the identity in make_config: cam2world == cam2base
            vol.integrate_frames_device([d.data_ptr() for d in devs], np.stack([c for c, _ in frames]))
        else:
            for (c2b, _), d in zip(frames, devs):
                vol.integrate_cam2base(d.data_ptr(), c2b)
        vol.sync()
        return vol.download()


@pytest.mark.parametrize("name", NAMES)
def test_reference_kernel_on_device_reproduces_golden_and_product(cuda, oracle, name):
    g = Golden(name)
    ref = RefHip()
    h, w = g.depth.shape[1:]
    rt, rw = _ref_run(cuda, ref, g.K, g.frames, g.dims, g.origin, g.vs, g.trunc)
    # the fixture itself came from the host build of the same function: device == host, bit for bit
    assert np.array_equal(rw, g.weight)
    assert np.array_equal(rt.view(np.uint32), g.tsdf.view(np.uint32))
    # the CPU restatement
    ot, ow = oracle.init_grid(g.dims)
    for c2b, depth in g.frames:
        oracle.integrate(g.K, c2b, depth, g.dims, g.origin, g.vs, g.trunc, ot, ow)
    assert np.array_equal(ow, rw) and np.array_equal(ot.view(np.uint32), rt.view(np.uint32))
    # the product: default per-frame kernel and the fused sequence path
    for variant, fused in ((3, False), (0, False), (0, True), (8, True)):   # 3: one kernel per frame; 0: collected
        pt, pw = _product_run(cuda, g.K, g.frames, g.dims, g.origin, g.vs, g.trunc, h, w, variant, fused)
        assert np.array_equal(pw, rw), (variant, fused)
        assert np.array_equal(pt.view(np.uint32), rt.view(np.uint32)), (variant, fused)


@pytest.mark.parametrize("seed", range(8))
def test_reference_kernel_oracle_and_product_agree_on_random_inputs(cuda, oracle, seed):
    ref = RefHip()
    rng = np.random.default_rng(100 + seed)
    dims = (int(rng.integers(2, 40)) * 4, int(rng.integers(8, 120)), int(rng.integers(8, 60)))
    vs = float(rng.choice([0.004, 0.01, 0.02, 0.05]))
    trunc = float(np.float32(vs) * np.float32(5))
    origin = synth.surf_volume(max(dims), vs, z0=float(rng.uniform(-0.2, 2.0)))
    h, w = int(rng.integers(30, 200)), int(rng.integers(40, 260))
    K = np.array([rng.uniform(50, 250), 0, w / 2 + rng.uniform(-3, 3), 0, rng.uniform(50, 250),
                  h / 2 + rng.uniform(-3, 3), 0, 0, 1], np.float32)
    sc = synth.SurfScene(dims, vs, origin, K=K, h=h, w=w)
    odd = np.array([0.0, -1.0, 6.5, 6.0, np.nan, np.inf, -np.inf, 1e-42], np.float32)
    frames = []
    for k in range(5):
        c2b = synth.random_pose(rng, 0.5, 0.5)
        depth = sc.depth(c2b, quantize=bool(k & 1))
        depth[rng.integers(0, h, 60), rng.integers(0, w, 60)] = rng.choice(odd, 60)
        frames.append((c2b, depth))
    rt, rw = _ref_run(cuda, ref, K, frames, dims, origin, vs, trunc)
    ot, ow = oracle.init_grid(dims)
    for c2b, depth in frames:
        oracle.integrate(K, c2b, depth, dims, origin, vs, trunc, ot, ow, threads=4)
    assert rw.sum() > 0
    assert np.array_equal(ow, rw) and np.array_equal(ot.view(np.uint32), rt.view(np.uint32))
    for variant, fused in ((3, False), (0, False), (0, True), (8, True), (7, True), (1, False)) + (((2, False), (23, False)) if capi.experiments_build() else ()):
        pt, pw = _product_run(cuda, K, frames, dims, origin, vs, trunc, h, w, variant, fused)
        assert np.array_equal(pw, rw), (variant, fused)
        assert np.array_equal(pt.view(np.uint32), rt.view(np.uint32)), (variant, fused)
