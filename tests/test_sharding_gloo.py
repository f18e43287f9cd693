"""The N > 1 path on CPU: two (and three) processes, gloo backend.  semantic_slam_amd.sharded
drives a stand-in slab (the oracle behind the Volume interface -- there is no GPU here) through
the same partition / halo / gather code the GPU job runs over RCCL, and the gathered result
must equal the unsharded grid bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from semantic_slam_amd import synth  # noqa: E402
from semantic_slam_amd.sharded import ShardedVolume, slab_range  # noqa: E402

DIMS, VS = (40, 24, 21), 0.02        # 21 slices: uneven slabs


class OracleSlab:
    """Test stand-in with the slab interface of capi.Volume, computing with the CPU oracle."""

    def __init__(self, zb, ze):
        from oracle.oracle import Oracle
        self.o = Oracle()
        self.zb, self.ze = zb, ze
        self.origin = synth.surf_volume(40, VS, 0.5)
        self.t, self.w = self.o.init_grid(DIMS, zb, ze)

    def integrate(self, depth, cam2world):
        self.o.integrate(synth.TUM_K, cam2world, depth, DIMS, self.origin, VS, 0.1, self.t, self.w,
                         z_begin=self.zb, z_end=self.ze)

    def sync(self):
        pass

    def download(self):
        return self.t.copy(), self.w.copy()

    def copy_slices(self, z_local, n):
        s = DIMS[0] * DIMS[1]
        return self.t[z_local * s:(z_local + n) * s].copy(), self.w[z_local * s:(z_local + n) * s].copy()

    def extract_crossings(self, halo=None, thr=0.9):
        return self.o.zero_crossings(self.t, self.w, DIMS[:2], self.zb, self.ze, VS, self.origin, halo, thr)

    def extract_mesh(self, halo=None, thr=0.9):
        return self.o.mesh_triangles(self.t, self.w, DIMS[:2], self.zb, self.ze, VS, self.origin, halo, thr)

    def extract_surface(self, thr=0.9):
        dims = (DIMS[0], DIMS[1], self.ze - self.zb)
        pts = self.o.surface_points(self.t, self.w, dims, VS, self.origin, thr)
        # slab-local z -> global z exactly as the device kernel does: origin + (z_begin + lz) * size
        keep = (np.abs(self.t) != 0) & (self.w > thr)
        lz = np.flatnonzero(keep) // (DIMS[0] * DIMS[1])
        pts[:, 2] = self.origin[2] + (self.zb + lz).astype(np.float32) * np.float32(VS)
        return pts


def frames():
    origin = synth.surf_volume(40, VS, 0.5)
    sc = synth.SurfScene(DIMS, VS, origin)
    return [(sc.pose(k, 8), sc.depth(sc.pose(k, 8))) for k in range(3)]


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        vol = ShardedVolume(DIMS, OracleSlab, dist=dist, comm_device="cpu")
        assert (vol.z_begin, vol.z_end) == slab_range(DIMS[2], rank, world)
        for pose, depth in frames():
            vol.integrate(depth, pose)
        halo = vol.halo_exchange()
        ht, hw = halo if halo is not None else (None, None)
        t, w = vol.gather(dst=0)
        pts = vol.gather_surface(dst=0)
        xing = vol.gather_crossings(dst=0)
        mesh = vol.gather_mesh(dst=0)
        q.put((rank, ht, hw, t, w, pts, xing, mesh))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_whole(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        r = q.get(timeout=120)
        out[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    whole = OracleSlab(0, DIMS[2])
    for pose, depth in frames():
        whole.integrate(depth, pose)
    _, _, t, w, pts, xing, mesh = out[0]
    assert whole.w.sum() > 0
    assert np.array_equal(w, whole.w) and np.array_equal(t.view(np.uint32), whole.t.view(np.uint32))
    assert np.array_equal(pts.view(np.uint32), whole.extract_surface().view(np.uint32))
    want_m = whole.extract_mesh(None)
    assert len(want_m) > 100 and np.array_equal(mesh.view(np.uint32), want_m.view(np.uint32))
    want_x = whole.extract_crossings(None)
    assert len(want_x) > 100 and np.array_equal(xing.view(np.uint32), want_x.view(np.uint32))
    s = DIMS[0] * DIMS[1]
    for r in range(world):
        zb, ze = slab_range(DIMS[2], r, world)
        ht, hw = out[r][0], out[r][1]
        if r == world - 1:
            assert ht is None and hw is None          # top slab has no upper neighbour
        else:
            assert np.array_equal(ht, whole.t[ze * s:(ze + 1) * s])   # = first slice of the next slab
            assert np.array_equal(hw, whole.w[ze * s:(ze + 1) * s])
        if r != 0:
            assert out[r][2] is None and out[r][4] is None   # only dst holds the gathered results


def test_slab_ranges_tile_the_grid():
    for dz in (1, 7, 512, 1000):
        for world in (1, 2, 3, 8):
            cuts = [slab_range(dz, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == dz
            assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:]))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        slab_range(10, 3, 3)
