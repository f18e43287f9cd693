"""z-slab sharding of one TSDF grid over the ranks of a torch.distributed job (one process
per GPU, RCCL over xGMI on MI355X; gloo on CPU for tests).

Why z: the reference's layout is z-major (index = z*dim_y*dim_x + y*dim_x + x,
ref: src/tsdf.cu:52), so rank r of W owns the contiguous range z in [r*D/W, (r+1)*D/W) and
the gathered volume is the plain concatenation of the slabs.

Data movement
  * Integrate: none between ranks.  Every voxel's update depends only on its own state, the
    shared depth frame and the pose (ref: src/tsdf.cu:27-57), so each rank integrates its
    slab independently; the 1.2 MB depth frame reaches every GPU from its own host process.
    The kernel uses the GLOBAL z index, so a sharded grid is bit-identical to an unsharded one.
  * Extraction: the reference's own surface rule is per voxel (ref: src/tsdf.cu:179) and
    needs nothing from a neighbour; neighbourhood-based extraction (zero crossings, marching
    tetrahedra) needs slice z_end from the upper neighbour -- `halo_exchange()` is that one-voxel
    halo: one grouped send/recv per slab boundary, dim_y*dim_x*8 bytes (2 MiB at 512^2).
  * Gather: slabs (or extracted lists) to one rank in z order (for the .bin/.ply writers).

comm_device selects where the communication buffers live: "cuda" = in HBM, exchanged by RCCL
(the slab's slices are copied device-to-device into the send buffer and the received halo is
handed to the extraction kernels as device pointers: no host hop anywhere), "cpu" = host
buffers through gloo (tests; rehearsal of N ranks on fewer GPUs).

The slab object is anything with the small interface of `capi.Volume` (integrate*, download,
copy_slices, extract_*); tests drive the same code on CPU with a stand-in slab.
"""
import numpy as np

# largest single message of a gather, in floats (256 MiB): slabs are sent in whole-slice pieces no larger than this
_GATHER_PIECE = 64 * 1024 * 1024


def slab_range(dim_z, rank, world):
    """Contiguous, balanced split of [0, dim_z) -- slabs differ by at most one slice."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return rank * dim_z // world, (rank + 1) * dim_z // world


def exchange_slices(dist, group, send, lower, recv, upper):
    """The wire of the one-voxel halo: buffer `send` (or None) to global rank `lower` and buffer `recv` (or None) from
    global rank `upper`, as ONE grouped point-to-point call (RCCL for device buffers, gloo for host buffers).  Returns
    when `recv` may be read by any stream (the extraction kernels run on the slab's own stream, not on torch's)."""
    ops = []
    if send is not None:
        ops.append(dist.P2POp(dist.isend, send, lower, group))
    if recv is not None:
        ops.append(dist.P2POp(dist.irecv, recv, upper, group))
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if (recv is not None and recv.is_cuda) or (send is not None and send.is_cuda):
        import torch
        torch.cuda.current_stream().synchronize()


class ShardedVolume:
    def __init__(self, dims, make_slab, dist=None, group=None, comm_device="cpu"):
        """dims: global (dim_x, dim_y, dim_z).  make_slab(z_begin, z_end) -> slab object for this
        rank.  dist: the torch.distributed module when initialised (None = single process).
        comm_device: "cuda" to exchange through RCCL, "cpu" through gloo."""
        if comm_device not in ("cpu", "cuda"):
            raise ValueError("comm_device must be 'cpu' or 'cuda'")
        self.dims = tuple(int(d) for d in dims)
        self.dist = dist if (dist is not None and dist.is_initialized()) else None
        self.group = group
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.comm_device = comm_device
        self.z_begin, self.z_end = slab_range(self.dims[2], self.rank, self.world)
        self.slab = make_slab(self.z_begin, self.z_end)
        self._halo_buf = None   # keeps a device-resident halo alive while extraction kernels read it

    # ---- integrate: no collective ---------------------------------------------------------
    def integrate(self, depth_host, cam2world):
        self.slab.integrate(depth_host, cam2world)

    def integrate_device(self, depth_ptr, cam2world):
        self.slab.integrate_device(depth_ptr, cam2world)

    def sync(self):
        self.slab.sync()

    @property
    def slice_voxels(self):
        return self.dims[0] * self.dims[1]

    @property
    def n_slices(self):
        return self.z_end - self.z_begin

    def _torch_dev(self):
        import torch
        if self.comm_device == "cuda":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def _global_rank(self, group_rank):
        if self.group is None:
            return group_rank
        return self.dist.get_global_rank(self.group, group_rank)

    def _range_of(self, r):
        return slab_range(self.dims[2], r, self.world)

    # ---- one-voxel halo --------------------------------------------------------------------
    def halo_exchange(self):
        """Slice z_end -- the first slice of the next non-empty slab -- for this rank, or None on the
        last slab.  comm_device "cpu": a pair of float32 arrays (tsdf, weight) of dim_y*dim_x;
        comm_device "cuda": a pair of DEVICE ADDRESSES (ints) of the same two arrays inside a buffer
        this object keeps alive until the next exchange -- exactly what `extract_crossings` /
        `extract_mesh` take as `halo`, so the slice never visits the host.
        Every rank sends its first slice down and receives its upper neighbour's, all boundaries at
        once (point-to-point, so on xGMI each boundary uses its own link)."""
        self._halo_buf = None
        if self.world == 1:
            return None
        import torch
        dist = self.dist
        owners = [r for r in range(self.world) if self._range_of(r)[1] > self._range_of(r)[0]]
        if self.rank not in owners:
            return None
        i = owners.index(self.rank)
        lower = owners[i - 1] if i > 0 else None
        upper = owners[i + 1] if i + 1 < len(owners) else None
        n = self.slice_voxels
        dev = self._torch_dev()
        send = recv = None
        if lower is not None:
            send = torch.empty(2 * n, dtype=torch.float32, device=dev)
            self._slices_into(send, 0, 1)
        if upper is not None:
            recv = torch.empty(2 * n, dtype=torch.float32, device=dev)
        exchange_slices(dist, self.group, send, None if lower is None else self._global_rank(lower),
                        recv, None if upper is None else self._global_rank(upper))
        if recv is None:
            return None
        if recv.is_cuda:
            self._halo_buf = recv
            return recv.data_ptr(), recv.data_ptr() + 4 * n
        host = recv.numpy()
        return host[:n].copy(), host[n:].copy()

    def _slices_into(self, buf, z_local, n_slices):
        """n_slices whole slices from slab-local z_local into buf (tsdf values, then weights), on the comm device."""
        n = n_slices * self.slice_voxels
        if buf.is_cuda:
            # device to device, straight into the RCCL send buffer
            self.slab.copy_slices_to_device(z_local, n_slices, buf.data_ptr(), buf.data_ptr() + 4 * n)
        else:
            import torch
            t, w = self.slab.copy_slices(z_local, n_slices)
            buf[:n].copy_(torch.from_numpy(t))
            buf[n:2 * n].copy_(torch.from_numpy(w))

    # ---- gather in z order --------------------------------------------------------------------
    def gather(self, dst=0):
        """Whole grid (tsdf, weight) as host arrays on rank dst, None elsewhere.  Slabs may differ in size.
        Pieces of at most 256 MiB travel through buffers on the comm device (HBM + RCCL, or host + gloo)."""
        if self.world == 1:
            return self.slab.download()
        import torch
        dist = self.dist
        s = self.slice_voxels
        per_piece = max(1, _GATHER_PIECE // (2 * s))      # slices per message
        dev = self._torch_dev()
        if self.rank == dst:
            n_all = self.dims[2] * s
            out_t, out_w = np.empty(n_all, np.float32), np.empty(n_all, np.float32)
            for r in range(self.world):
                zb, ze = self._range_of(r)
                if r == dst:
                    if ze > zb:
                        t, w = self.slab.download()
                        out_t[zb * s:ze * s], out_w[zb * s:ze * s] = t, w
                    continue
                for z0 in range(zb, ze, per_piece):
                    k = min(per_piece, ze - z0)
                    buf = torch.empty(2 * k * s, dtype=torch.float32, device=dev)
                    dist.recv(buf, self._global_rank(r), group=self.group)
                    a = buf.cpu().numpy()
                    out_t[z0 * s:(z0 + k) * s] = a[:k * s]
                    out_w[z0 * s:(z0 + k) * s] = a[k * s:]
            return out_t, out_w
        for z0 in range(self.z_begin, self.z_end, per_piece):
            k = min(per_piece, self.z_end - z0)
            buf = torch.empty(2 * k * s, dtype=torch.float32, device=dev)
            self._slices_into(buf, z0 - self.z_begin, k)
            dist.send(buf, self._global_rank(dst), group=self.group)
        return None, None

    def _gather_rows(self, rows, width, dst):
        """Concatenate per-rank float32 arrays [n_r, *width] in rank (= z) order on rank dst.  Counts travel by
        all_gather, the rows point to point, both through buffers on the comm device."""
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1)
        if self.world == 1:
            return rows.reshape((-1,) + width)
        import torch
        dist = self.dist
        dev = self._torch_dev()
        mine = torch.tensor([rows.size], dtype=torch.int64, device=dev)
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(counts, mine, group=self.group)
        counts = [int(c.item()) for c in counts]
        if self.rank == dst:
            parts = []
            for r in range(self.world):
                if r == dst:
                    parts.append(rows)
                elif counts[r] > 0:
                    buf = torch.empty(counts[r], dtype=torch.float32, device=dev)
                    dist.recv(buf, self._global_rank(r), group=self.group)
                    parts.append(buf.cpu().numpy())
            return np.concatenate(parts).astype(np.float32).reshape((-1,) + width)
        if rows.size > 0:
            dist.send(torch.from_numpy(rows).to(dev), self._global_rank(dst), group=self.group)
        return None

    def gather_crossings(self, weight_thresh=0.9, dst=0):
        """Zero-crossing vertices of the whole grid in grid order on rank dst.  This is the
        neighbourhood-based extraction the halo exists for: every rank first receives slice z_end
        from its upper neighbour (halo_exchange, RCCL/gloo), extracts its slab's vertices on its
        device, and the lists are concatenated in z order -- identical to the unsharded list."""
        pts = self.slab.extract_crossings(self.halo_exchange(), weight_thresh)
        return self._gather_rows(pts, (3,), dst)

    def gather_mesh(self, weight_thresh=0.9, dst=0):
        """Marching-tetrahedra triangles of the whole grid in grid order on rank dst: halo exchange, per-slab
        extraction on each device, concatenation in z order (identical to the unsharded mesh)."""
        tri = self.slab.extract_mesh(self.halo_exchange(), weight_thresh)
        return self._gather_rows(tri, (3, 3), dst)

    def gather_surface(self, weight_thresh=0.9, dst=0):
        """Surface points of the whole grid in grid order on rank dst (ref rule: src/tsdf.cu:179).
        Per-voxel rule: each rank compacts its own slab on its GPU, lists are concatenated in z order."""
        return self._gather_rows(self.slab.extract_surface(weight_thresh), (3,), dst)
