"""Randomised parity: grids of every shape class (row-mapped, flat, scalar fallback, partial chunks),
image sizes and intrinsics that are not TUM's, cameras inside / behind / beside the volume (the
camera-plane guard of the fast projection), fused sequences with optional masks -- always bit-exact
against the oracle."""
import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    shape_class = seed % 5
    if shape_class == 0:
        dims = (256 * int(rng.integers(1, 3)), int(rng.integers(3, 20)), int(rng.integers(2, 12)))   # row-mapped kernels
    elif shape_class == 1:
        dims = (4 * int(rng.integers(1, 90)), int(rng.integers(1, 40)), int(rng.integers(1, 20)))    # flat mapping
    elif shape_class == 2:
        dims = (int(rng.integers(1, 70)) | 1, int(rng.integers(1, 30)), int(rng.integers(1, 20)))    # odd rows: scalar kernel
    elif shape_class == 3:
        dims = (4, int(rng.integers(1, 5)), int(rng.integers(1, 5)))                                 # less than one chunk
    else:
        dims = (int(rng.integers(8, 40)) * 4, int(rng.integers(8, 40)), int(rng.integers(8, 30)))
    h, w = int(rng.integers(24, 200)), int(rng.integers(32, 260))
    K = np.array([rng.uniform(30, 400), 0, w / 2 + rng.uniform(-10, 10), 0, rng.uniform(30, 400),
                  h / 2 + rng.uniform(-10, 10), 0, 0, 1], np.float32)
    vs = float(rng.choice([0.003, 0.01, 0.02, 0.05]))
    ext = np.array(dims) * vs
    # volume placed so that cameras can end up inside, behind or beside it
    origin = (rng.uniform(-1.0, 0.2, 3) * ext + np.array([0, 0, rng.uniform(-0.5, 1.5)])).astype(np.float32)
    trunc = float(np.float32(vs) * np.float32(rng.choice([2, 5, 9])))
    max_depth = float(rng.choice([6.0, 2.5, 10.0]))
    return rng, dims, h, w, K, vs, origin, trunc, max_depth


# 0: classification decided per launch, 7: never, 8: always (bricks per wavefront), 11: always (rows per workgroup)
# -- all inside the driver's single pytest run (no env knob)
@pytest.mark.parametrize("variant", capi.variants(0, 7, 8, 11, 13))
@pytest.mark.parametrize("seed", range(40))
def test_random_configuration(cuda, oracle, seed, variant):
    run_case(cuda, oracle, seed, variant)


@pytest.mark.parametrize("seed", range(40, 52))
def test_random_long_sequences_and_brick_shapes(cuda, oracle, seed):
    """More frames than one pass holds (runs of claimed frames, a second launch on the same volume, the per-launch
    decision) over a random valid brick shape."""
    dims = random_case(seed)[1]
    run_case(cuda, oracle, seed, 8 if seed % 2 else 0, shape=random_brick_shape(seed, dims), long=True)


def random_brick_shape(seed, dims):
    if dims[0] % 4:
        return None
    rs = np.random.default_rng(seed)
    quads = dims[0] // 4
    q = int(rs.choice([d for d in range(1, min(quads, 64) + 1) if quads % d == 0]))
    r = int(rs.integers(1, 64 // q + 1))
    return q, r, int(rs.integers(1, 64 // (q * r) + 1))


def run_case(cuda, oracle, seed, variant, shape=None, long=False):
    """One random configuration under one kernel variant (tools/fuzz_stress.py runs many more seeds)."""
    rng, dims, h, w, K, vs, origin, trunc, max_depth = random_case(seed)
    base = synth.random_pose(rng, 0.5, 0.5) if seed % 3 else synth.identity_pose()
    cfg = capi.make_config(dims, vs, origin, trunc=trunc, K=K, base2world=base, im_height=h, im_width=w,
                           max_depth=max_depth)
    scene = synth.SurfScene(dims, vs, origin, K=K, h=h, w=w)
    n_frames = int(rng.integers(1, 7))
    if long:
        n_frames += 30 + seed % 9
    frames = []
    for k in range(n_frames):
        # mostly cameras that look at the volume from outside, from its boundary or from inside it
        centre = origin.astype(np.float64) + np.array(dims) * vs / 2.0
        dist = float(rng.choice([0.0, 0.3, 1.0, 2.5])) * float(max(dims) * vs) + float(rng.uniform(0.0, 0.5))
        c2b_want = synth.look_at_pose(rng, centre, dist) if k % 4 else synth.random_pose(rng, 0.8, 0.5)
        c2w = oracle.multiply(base, c2b_want)             # so that inverse(base) * c2w is (nearly) c2b_want
        c2b = oracle.cam2base(base, c2w)
        mode = int(rng.integers(0, 4))
        if mode == 0:
            depth = scene.depth(c2b, quantize=bool(rng.integers(0, 2)))
        elif mode == 1:
            depth = np.full((h, w), float(rng.uniform(0.2, max_depth)), np.float32)
        elif mode == 2:
            depth = rng.uniform(-0.5, max_depth * 1.2, (h, w)).astype(np.float32)       # noise incl. invalid values
        else:
            depth = scene.depth(c2b)
            depth[rng.integers(0, h, 20), rng.integers(0, w, 20)] = 0.0
        mask = None
        if dims[0] % 4 == 0 and rng.integers(0, 3) == 0:
            mask = (rng.uniform(0, 1, (h, w)) < 0.7).astype(np.uint8) * 255
        frames.append((c2w, c2b, depth, mask))
    ref_t, ref_w = oracle.init_grid(dims)
    for _, c2b, depth, mask in frames:
        d = depth if mask is None else oracle.mask_depth(depth, mask)
        oracle.integrate(K, c2b, d, dims, origin, vs, trunc, ref_t, ref_w, max_depth=max_depth)
    with capi.Volume(cfg) as vol:
        vol.set_kernel_variant(variant)
        if shape is not None:
            vol.set_brick_shape(*shape)
        keep = [(cuda.from_numpy(np.ascontiguousarray(d)).cuda(), None if m is None else cuda.from_numpy(m).cuda())
                for _, _, d, m in frames]
        if (seed % 2 == 0 or long) and dims[0] % 4 == 0:      # as one fused sequence
            vol.integrate_frames_device([d.data_ptr() for d, _ in keep], np.stack([f[0] for f in frames]),
                                        [None if m is None else m.data_ptr() for _, m in keep])
        else:                                       # frame by frame
            for (c2w, _, _, _), (d, m) in zip(frames, keep):
                if m is None:
                    vol.integrate_device(d.data_ptr(), c2w)
                else:
                    vol.integrate_masked_device(d.data_ptr(), m.data_ptr(), c2w)
        t, wgt = vol.download()
        n_surface = vol.count_surface()
    assert np.array_equal(wgt, ref_w), f"seed {seed} dims {dims}: weights differ"
    assert np.array_equal(t.view(np.uint32), ref_t.view(np.uint32)), f"seed {seed} dims {dims}: TSDF differs"
    assert n_surface == len(oracle.surface_points(ref_t, ref_w, dims, vs, origin))
