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
    cubes) needs slice z_end from the upper neighbour -- `halo_exchange()` is that one-voxel
    halo: one grouped send/recv per slab boundary, dim_y*dim_x*8 bytes (2 MiB at 512^2).
  * Gather: slabs to one rank in z order (for the .bin/.ply writers).

The slab object is anything with the small interface of `capi.Volume` (integrate*, download,
copy_slices, extract_surface); tests drive the same code on CPU with a stand-in slab.
"""
import numpy as np


def slab_range(dim_z, rank, world):
    """Contiguous, balanced split of [0, dim_z) -- slabs differ by at most one slice."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return rank * dim_z // world, (rank + 1) * dim_z // world


class ShardedVolume:
    def __init__(self, dims, make_slab, dist=None, group=None, comm_device="cpu"):
        """dims: global (dim_x, dim_y, dim_z).  make_slab(z_begin, z_end) -> slab object for this
        rank.  dist: the torch.distributed module when initialised (None = single process).
        comm_device: "cuda" to exchange through RCCL, "cpu" through gloo."""
        self.dims = tuple(int(d) for d in dims)
        self.dist = dist if (dist is not None and dist.is_initialized()) else None
        self.group = group
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self.comm_device = comm_device
        self.z_begin, self.z_end = slab_range(self.dims[2], self.rank, self.world)
        self.slab = make_slab(self.z_begin, self.z_end)

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

    # ---- one-voxel halo --------------------------------------------------------------------
    def halo_exchange(self):
        """Returns (tsdf, weight) of global slice z_end -- the first slice of the next
        non-empty slab -- as float32 arrays of dim_y*dim_x, or (None, None) on the last slab.
        Every rank sends its first slice down and receives its upper neighbour's, all
        boundaries at once (point-to-point, so on xGMI each boundary uses its own link)."""
        if self.world == 1:
            return None, None
        import torch
        dist = self.dist
        owners = [r for r in range(self.world) if slab_range(self.dims[2], r, self.world)[1] >
                  slab_range(self.dims[2], r, self.world)[0]]       # ranks with a non-empty slab
        if self.rank not in owners:
            return None, None
        i = owners.index(self.rank)
        lower = owners[i - 1] if i > 0 else None
        upper = owners[i + 1] if i + 1 < len(owners) else None
        n = self.slice_voxels
        dev = torch.device("cuda", torch.cuda.current_device()) if self.comm_device == "cuda" else torch.device("cpu")
        ops, recv = [], None
        if lower is not None:
            send = torch.empty(2 * n, dtype=torch.float32, device=dev)
            self._first_slice_into(send)
            ops.append(dist.P2POp(dist.isend, send, self._global_rank(lower), self.group))
        if upper is not None:
            recv = torch.empty(2 * n, dtype=torch.float32, device=dev)
            ops.append(dist.P2POp(dist.irecv, recv, self._global_rank(upper), self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if recv is None:
            return None, None
        host = recv.cpu().numpy()
        return host[:n].copy(), host[n:].copy()

    def _global_rank(self, group_rank):
        if self.group is None:
            return group_rank
        return self.dist.get_global_rank(self.group, group_rank)

    def _first_slice_into(self, buf):
        """buf: 2*slice floats (tsdf then weight) on the comm device."""
        n = self.slice_voxels
        if buf.is_cuda and hasattr(self.slab, "copy_slices_to_device"):
            self.slab.copy_slices_to_device(0, 1, buf.data_ptr(), buf.data_ptr() + 4 * n)  # D2D, no host hop
        else:
            import torch
            t, w = self.slab.copy_slices(0, 1)
            buf[:n].copy_(torch.from_numpy(t))
            buf[n:].copy_(torch.from_numpy(w))

    # ---- gather in z order --------------------------------------------------------------------
    def gather(self, dst=0):
        """Whole grid (tsdf, weight) on rank dst, None elsewhere.  Slabs may differ in size."""
        t, w = self.slab.download()
        if self.world == 1:
            return t, w
        import torch
        dist = self.dist
        if self.rank == dst:
            parts_t, parts_w = [], []
            for r in range(self.world):
                zb, ze = slab_range(self.dims[2], r, self.world)
                n = (ze - zb) * self.slice_voxels
                if r == dst:
                    parts_t.append(t); parts_w.append(w)
                elif n > 0:
                    buf = torch.empty(2 * n, dtype=torch.float32)
                    dist.recv(buf, self._global_rank(r), group=self.group)
                    a = buf.numpy()
                    parts_t.append(a[:n].copy()); parts_w.append(a[n:].copy())
            return np.concatenate(parts_t), np.concatenate(parts_w)
        if t.size > 0:
            buf = torch.from_numpy(np.concatenate([t, w]))
            dist.send(buf, self._global_rank(dst), group=self.group)
        return None, None

    def gather_crossings(self, weight_thresh=0.9, dst=0):
        """Zero-crossing vertices of the whole grid in grid order on rank dst.  This is the
        neighbourhood-based extraction the halo exists for: every rank first receives slice z_end
        from its upper neighbour (halo_exchange, RCCL/gloo), extracts its slab's vertices on its
        device, and the lists are concatenated in z order -- identical to the unsharded list."""
        ht, hw = self.halo_exchange()
        pts = self.slab.extract_crossings(None if ht is None else (ht, hw), weight_thresh)
        if self.world == 1:
            return pts
        gathered = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(pts, gathered, dst=self._global_rank(dst), group=self.group)
        if self.rank != dst:
            return None
        return np.concatenate([g.reshape(-1, 3) for g in gathered]).astype(np.float32)

    def gather_mesh(self, weight_thresh=0.9, dst=0):
        """Marching-tetrahedra triangles of the whole grid in grid order on rank dst: halo exchange, per-slab
        extraction on each device, concatenation in z order (identical to the unsharded mesh)."""
        ht, hw = self.halo_exchange()
        tri = self.slab.extract_mesh(None if ht is None else (ht, hw), weight_thresh)
        if self.world == 1:
            return tri
        gathered = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(tri, gathered, dst=self._global_rank(dst), group=self.group)
        if self.rank != dst:
            return None
        return np.concatenate([g.reshape(-1, 3, 3) for g in gathered]).astype(np.float32)

    def gather_surface(self, weight_thresh=0.9, dst=0):
        """Surface points of the whole grid in grid order on rank dst (ref rule: src/tsdf.cu:179).
        Per-voxel rule: each rank compacts its own slab on its GPU, lists are concatenated in z order."""
        pts = self.slab.extract_surface(weight_thresh)
        if self.world == 1:
            return pts
        gathered = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(pts, gathered, dst=self._global_rank(dst), group=self.group)
        if self.rank != dst:
            return None
        return np.concatenate([g.reshape(-1, 3) for g in gathered]).astype(np.float32)
