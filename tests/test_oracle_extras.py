"""Project-defined rules of the CPU oracle that have no reference counterpart (zero-crossing
vertices, label fusion, mask composition): hand-checkable known-answer cases, so the GPU parity
tests against the oracle mean something."""
import numpy as np

f32 = np.float32


def test_zero_crossing_known_answers(oracle):
    dims, vs = (3, 2, 2), 0.5
    origin = np.array([1.0, 2.0, 3.0], f32)
    t = np.ones(12, f32)
    w = np.ones(12, f32)
    t[0] = -0.25            # voxel (0,0,0); neighbours +x (1,0,0)=1, +y (0,1,0)=1, +z (0,0,1)=1 are positive
    w[3] = 0.0              # (0,1,0) unobserved: its edge is skipped
    pts = oracle.zero_crossings(t, w, dims[:2], 0, 2, vs, origin)
    s = f32(-0.25) / (f32(-0.25) - f32(1.0))           # 0.2
    want = np.array([[f32(1.0) + s * f32(vs), 2.0, 3.0],        # x edge
                     [1.0, 2.0, f32(3.0) + s * f32(vs)]], f32)  # z edge (y edge skipped: weight 0)
    assert np.array_equal(pts, want)
    # the same grid as two slabs: the z edge needs the halo
    lower = oracle.zero_crossings(t[:6], w[:6], dims[:2], 0, 1, vs, origin, halo=(t[6:], w[6:]))
    assert np.array_equal(lower, want)
    assert len(oracle.zero_crossings(t[:6], w[:6], dims[:2], 0, 1, vs, origin, halo=None)) == 1
    # exact zero counts as positive: -0.25 | 0.0 crosses, 0.0 | 1.0 does not
    t2 = np.array([-0.25, 0.0, 1.0], f32)
    p2 = oracle.zero_crossings(t2, np.ones(3, f32), (3, 1), 0, 1, 1.0, np.zeros(3, f32))
    assert len(p2) == 1 and p2[0, 0] == f32(1.0)


def test_compose_labels_rule(oracle):
    masks = np.zeros((3, 2, 4), np.uint8)
    masks[0, :, :3] = 255
    masks[1, :, 1:] = 255
    masks[2, 0, :] = 255
    lab, sc = oracle.compose_labels(masks, [5, 7, 9], [0.9, 0.95, 0.95])
    # pixel (0,0): instances 0 and 2 -> 2 has the higher score; (0,1): 0,1,2 -> 1 (0.95, lower index than 2)
    assert lab.tolist() == [[9, 7, 7, 7], [5, 7, 7, 7]]
    assert sc[0, 0] == f32(0.95) and sc[1, 0] == f32(0.9)
    lab0, sc0 = oracle.compose_labels(np.zeros((1, 2, 2), np.uint8), [3], [0.9])
    assert not lab0.any() and not sc0.any()


def test_label_fusion_rule(oracle):
    """One voxel in front of the camera, inside the band, fed a sequence of (label, score) observations."""
    K = np.array([100, 0, 32, 0, 100, 24, 0, 0, 1], f32)
    dims, vs = (4, 1, 1), 0.01
    origin = np.array([0.0, 0.0, 1.0], f32)           # voxel 0 projects to pixel (32, 24)
    depth = np.full((48, 64), 1.0, f32)               # surface exactly at the voxel: diff = 0, in band
    pose = np.eye(4, dtype=f32).ravel()
    lab, fp, bp = np.zeros(4, np.uint16), np.zeros(4, f32), np.zeros(4, f32)

    def see(l, s):
        li = np.full((48, 64), l, np.uint16)
        si = np.full((48, 64), s, f32)
        return oracle.integrate_labels(K, pose, depth, li, si, dims, origin, vs, 0.05, lab, fp, bp, prob_thd=0.5)

    assert see(0, 0.9) == 0 and lab[0] == 0                       # no instance: nothing
    assert see(7, 0.9) == 4 and lab[0] == 7 and fp[0] == f32(0.9)   # adopt
    see(7, 0.8)
    assert fp[0] == f32(0.9) + f32(0.8) and bp[0] == 0            # same label: Fp += s
    see(3, 0.9)
    assert lab[0] == 7 and bp[0] == f32(0.9)                      # other label: Bp += s, P = 1.7/2.6 >= 0.5
    see(3, 0.9)
    assert lab[0] == 3 and fp[0] == f32(0.9) and bp[0] == 0       # P = 1.7/3.5 < 0.5: re-adopt
    # outside the band (free space in front of the surface) nothing happens
    far = np.full((48, 64), 3.0, f32)
    li, si = np.full((48, 64), 9, np.uint16), np.full((48, 64), 0.9, f32)
    assert oracle.integrate_labels(K, pose, far, li, si, dims, origin, vs, 0.05, lab, fp, bp) == 0


def test_mesh_rule_known_answers(oracle):
    """One cube with one corner inside: three tetrahedra touch corner 0 ... the surface is a closed fan."""
    t = np.ones(8, f32)
    t[0] = -1.0                                   # corner (0,0,0) inside, all others outside at +1: s = 0.5 on every edge
    w = np.ones(8, f32)
    tri = oracle.mesh_triangles(t, w, (2, 2), 0, 2, 1.0, np.zeros(3, f32))
    assert tri.shape == (6, 3, 3)                 # all six tetrahedra contain corner 0
    assert np.allclose(np.abs(tri).max(), 0.5) and np.all((tri == 0) | (tri == 0.5))
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert np.all(np.einsum("ij,ij->i", n, tri.mean(1)) > 0)        # wound away from the inside corner
    # an unobserved corner removes the cube; no upper slice and no halo removes it too
    w2 = w.copy(); w2[5] = 0.0
    assert len(oracle.mesh_triangles(t, w2, (2, 2), 0, 2, 1.0, np.zeros(3, f32))) == 0
    assert len(oracle.mesh_triangles(t[:4], w[:4], (2, 2), 0, 1, 1.0, np.zeros(3, f32))) == 0
    assert np.array_equal(oracle.mesh_triangles(t[:4], w[:4], (2, 2), 0, 1, 1.0, np.zeros(3, f32), halo=(t[4:], w[4:])), tri)


def test_colour_rule_known_answers(oracle):
    """tsdf-fusion-python's colour rule as restated in oracle_integrate_colour: per channel min(255, round((c*w_old + new)/w_new)),
    only on voxels the frame updates, channels packed B << 16 | G << 8 | R with image channel 0 = R."""
    dims, vs = (8, 8, 8), 0.05
    origin = np.array([-0.2, -0.2, 1.0], np.float32)
    K = np.array([100, 0, 32, 0, 100, 24, 0, 0, 1], np.float32)
    pose = np.eye(4, dtype=np.float32).ravel()
    depth = np.full((48, 64), 1.6, np.float32)            # behind the whole volume: every voxel in view is updated
    t, w = oracle.init_grid(dims)
    col = np.zeros(t.size, np.uint32)
    seq = [(10, 20, 30), (11, 20, 255), (12, 21, 0), (200, 0, 7)]
    acc = np.zeros(3)
    for k, rgbv in enumerate(seq):
        rgb = np.empty((48, 64, 3), np.uint8)
        rgb[:] = rgbv
        n_upd = oracle.integrate(K, pose, depth, dims, origin, vs, 0.25, t, w)
        n_col = oracle.integrate_colour(K, pose, depth, rgb, dims, origin, vs, 0.25, w, col)
        assert n_col == n_upd == t.size
        # running mean with rounding at every step, as the package does it (round-half-away like CUDA's roundf)
        acc = np.minimum(255, np.floor((acc * k + np.array(rgbv)) / (k + 1) + 0.5))
        want = (int(acc[2]) << 16) | (int(acc[1]) << 8) | int(acc[0])
        assert np.all(col == want), (k, hex(int(col[0])), hex(want))
    # a frame that updates nothing colours nothing
    before = col.copy()
    none = np.zeros((48, 64), np.float32)
    assert oracle.integrate(K, pose, none, dims, origin, vs, 0.25, t, w) == 0
    assert oracle.integrate_colour(K, pose, none, rgb, dims, origin, vs, 0.25, w, col) == 0
    assert np.array_equal(col, before)
