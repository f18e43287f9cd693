"""RCCL on the wire, on the one GPU the box has: a world of ONE rank whose halo goes to itself.

RCCL refuses two ranks on one device, so the 2-rank RCCL case of tests/test_gpu_sharding.py is skipped on a one-GPU box
and the halo has only ever travelled through gloo.  This test runs the product's wire function
(`sharded.exchange_slices`: one grouped isend + irecv of device buffers) over a real RCCL communicator with the rank as
its own lower and upper neighbour -- RCCL serves a grouped send/recv to oneself with its ordinary point-to-point kernel
-- between two slab handles of one grid held by the same process: slab B's first slice is copied device to device into
the send buffer, comes back through RCCL into the receive buffer, and its device addresses feed slab A's extraction
kernels.  Crossings and mesh of A (with that halo) + B must equal the whole grid's, bit for bit, and the slice must
arrive unchanged.  What this does not cover: two devices, xGMI, more than one rank (the driver's 8-GPU run).

The child process is given 150 s; a communicator that cannot be created is reported as a skip with its reason.  (The file
sorts last on purpose: a wire that hangs should not take other tests with it.)"""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from semantic_slam_amd import capi, synth  # noqa: E402
from semantic_slam_amd.sharded import exchange_slices  # noqa: E402
from test_gpu_sharding import DIMS, VS, ORIGIN, frames, free_port, make_slab, whole_grid  # noqa: E402

pytestmark = pytest.mark.gpu
SPLIT = 19


def child(port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        x = torch.full((4,), 3.0, device="cuda")
        dist.all_reduce(x)                      # creates the communicator
        torch.cuda.synchronize()
        assert float(x.sum()) == 12.0
    except Exception as e:   # noqa: BLE001 -- reported to the parent, which skips
        q.put(("unsupported", repr(e)[:300]))
        return
    try:
        a, b = make_slab(0, SPLIT), make_slab(SPLIT, DIMS[2])
        fr = frames()
        devs = [torch.from_numpy(d).cuda() for _, d in fr]
        for v in (a, b):
            v.integrate_frames_device([d.data_ptr() for d in devs], np.stack([p for p, _ in fr]))
        n = DIMS[0] * DIMS[1]
        send = torch.empty(2 * n, dtype=torch.float32, device="cuda")
        recv = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
        b.copy_slices_to_device(0, 1, send.data_ptr(), send.data_ptr() + 4 * n)      # B's first slice = A's halo
        exchange_slices(dist, None, send, 0, recv, 0)                                # ... through RCCL, to this very rank
        halo = (recv.data_ptr(), recv.data_ptr() + 4 * n)
        out = {"backend": dist.get_backend(), "slice": recv.cpu().numpy(), "sent": send.cpu().numpy(),
               "xa": a.extract_crossings(halo), "xb": b.extract_crossings(None),
               "ma": a.extract_mesh(halo), "mb": b.extract_mesh(None)}
        q.put(("ok", out))
        a.close()
        b.close()
    finally:
        dist.destroy_process_group()


def test_rccl_carries_the_halo_slice_between_two_slabs(cuda, oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=child, args=(free_port(), q))
    p.start()
    try:
        r = q.get(timeout=150)
    except Exception:   # noqa: BLE001 -- queue.Empty: the wire did not answer
        r = ("hung", None)
    finally:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()     # this exact child, nothing else
    if r[0] == "unsupported":
        pytest.skip("no RCCL communicator for one rank on this box: " + r[1])
    assert r[0] == "ok", "the grouped send/recv to the own rank did not complete within 150 s"
    out = r[1]
    assert out["backend"] == "nccl"
    t0, w0, _, xing0, mesh0 = whole_grid(cuda)
    n = DIMS[0] * DIMS[1]
    assert np.array_equal(out["slice"].view(np.uint32), out["sent"].view(np.uint32))
    assert np.array_equal(out["slice"][:n].view(np.uint32), t0[SPLIT * n:(SPLIT + 1) * n].view(np.uint32))
    assert np.array_equal(out["slice"][n:], w0[SPLIT * n:(SPLIT + 1) * n]) and out["slice"][n:].sum() > 100
    xing = np.concatenate([out["xa"], out["xb"]])
    mesh = np.concatenate([out["ma"], out["mb"]])
    assert len(xing0) > 100 and np.array_equal(xing.view(np.uint32), xing0.view(np.uint32))
    assert len(mesh0) > 100 and np.array_equal(mesh.view(np.uint32), mesh0.view(np.uint32))
    # and without the halo the boundary's crossings are missing: the slice matters
    with make_slab(0, SPLIT) as a:
        fr = frames()
        for pose, depth in fr:
            a.integrate(depth, pose)
        assert len(a.extract_crossings(None)) < len(out["xa"])
