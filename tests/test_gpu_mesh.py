"""Marching-tetrahedra mesh on the device (tsdf_extract_mesh) against the CPU restatement of the same
project-defined rule (no reference function exists for it: parity unpinned by reference output), the
z halo between slabs, topology of the result, and the .ply writer."""
from collections import Counter

import numpy as np
import pytest

from semantic_slam_amd import capi, synth

pytestmark = pytest.mark.gpu


def fuse(cuda, oracle, dims, vs, cuts, frames=4):
    origin = synth.surf_volume(max(dims), vs, 0.6)
    scene = synth.SurfScene(dims, vs, origin)
    vols = [capi.Volume(capi.make_config(dims, vs, origin, z_begin=a, z_end=b)) for a, b in zip(cuts[:-1], cuts[1:])]
    ref_t, ref_w = oracle.init_grid(dims)
    for k in range(frames):
        c2w = scene.pose(k, n=6)
        depth = scene.depth(c2w, quantize=True)
        for v in vols:
            v.integrate(depth, c2w)
        oracle.integrate(vols[0].cfg.cam_K, c2w, depth, dims, origin, vs, vols[0].cfg.trunc_margin, ref_t, ref_w)
    return origin, vols, ref_t, ref_w


def test_mesh_matches_oracle_and_slabs_use_the_halo(cuda, oracle):
    dims, vs = (80, 64, 48), 0.012
    cuts = [0, 17, 31, 48]
    origin, vols, ref_t, ref_w = fuse(cuda, oracle, dims, vs, cuts)
    whole = oracle.mesh_triangles(ref_t, ref_w, dims[:2], 0, dims[2], vs, origin)
    assert len(whole) > 3000
    parts, no_halo = [], 0
    for i, v in enumerate(vols):
        halo = vols[i + 1].copy_slices(0, 1) if i + 1 < len(vols) else None
        parts.append(v.extract_mesh(halo))
        no_halo += len(v.extract_mesh(None))
    got = np.concatenate(parts)
    assert got.shape == whole.shape and np.array_equal(got.view(np.uint32), whole.view(np.uint32))
    assert no_halo < len(whole), "cubes straddling slab boundaries need the halo slice"
    for v in vols:
        v.close()


def test_mesh_of_a_sphere_is_a_closed_manifold(cuda, oracle, tmp_path):
    """Upload the signed distance of a sphere: the extracted soup must be watertight (every edge in exactly
    two triangles, opposite directions), wound outwards, with Euler characteristic 2."""
    D, vs = 40, 0.05
    origin = np.zeros(3, np.float32)
    z, y, x = np.meshgrid(np.arange(D), np.arange(D), np.arange(D), indexing="ij")
    c = (D - 1) * vs / 2
    t = (np.sqrt((x * vs - c) ** 2 + (y * vs - c) ** 2 + (z * vs - c) ** 2) - 0.62).astype(np.float32).ravel()
    w = np.ones_like(t)
    with capi.Volume(capi.make_config((D, D, D), vs, origin)) as vol:
        vol.upload(t, w)
        tri = vol.extract_mesh()
        vol.save_mesh_ply(str(tmp_path / "sphere.ply"))
        vol.save_mesh_welded_ply(str(tmp_path / "sphere_welded.ply"))
    assert np.array_equal(tri.view(np.uint32), oracle.mesh_triangles(t, w, (D, D), 0, D, vs, origin).view(np.uint32))
    ids, edges, directed = {}, Counter(), Counter()
    for T in tri:
        k = [ids.setdefault(T[i].tobytes(), len(ids)) for i in range(3)]
        for a, b in ((0, 1), (1, 2), (2, 0)):
            edges[tuple(sorted((k[a], k[b])))] += 1
            directed[(k[a], k[b])] += 1
    assert set(edges.values()) == {2} and max(directed.values()) == 1
    assert len(ids) - len(edges) + len(tri) == 2
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert np.all(np.einsum("ij,ij->i", n, tri.mean(1) - c) > 0)
    raw = (tmp_path / "sphere.ply").read_bytes()
    head = (f"ply\nformat binary_little_endian 1.0\nelement vertex {3 * len(tri)}\nproperty float x\nproperty float y\n"
            f"property float z\nelement face {len(tri)}\nproperty list uchar int vertex_indices\nend_header\n").encode()
    assert raw.startswith(head) and len(raw) == len(head) + len(tri) * (36 + 13)
    assert raw[len(head):len(head) + 36 * len(tri)] == tri.tobytes()
    # the welded file: V - E + F = 2 with the file's own vertex count, outward unit normals
    rawm = (tmp_path / "sphere_welded.ply").read_bytes()
    headm = rawm[:rawm.index(b"end_header\n")].decode()
    nv = int(headm.split("element vertex ")[1].split()[0])
    assert nv == len(ids) and int(headm.split("element face ")[1].split()[0]) == len(tri) and "red" not in headm
    bodym = rawm[rawm.index(b"end_header\n") + len(b"end_header\n"):]
    assert len(bodym) == nv * 24 + len(tri) * 13 and nv - len(edges) + len(tri) == 2
    recm = np.frombuffer(bodym[:24 * nv], np.float32).reshape(nv, 6)
    assert np.allclose(np.linalg.norm(recm[:, 3:], axis=1), 1.0, atol=1e-5)
    assert np.all(np.einsum("ij,ij->i", recm[:, 3:], recm[:, :3] - c) > 0.9 * np.linalg.norm(recm[:, :3] - c, axis=1))
    fidx = np.frombuffer(bodym[24 * nv:], np.uint8).reshape(len(tri), 13)[:, 1:].copy().view(np.int32).reshape(-1, 3)
    assert np.array_equal(recm[:, :3][fidx].view(np.uint32), tri.view(np.uint32))
