"""bench.py's choice of grid at N GPUs (no GPU needed): N = 1 is BASELINE.json configs[1] (512^3 @ 5 mm); N > 1 cuts
configs[3] (1024^3 @ 2 mm) into N z-slabs unless --grid names another grid to cut; --scaling weak keeps the grid that grows
with N inside the same physical box.  The line's metric names the grid that ran."""
import argparse
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def args(**kw):
    a = argparse.Namespace(grid=512, grid_given=False, workload="sband", scaling="", voxel_mm=0.0, emulate_world=0, emulate_rank=0)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_one_gpu_is_configs1():
    dims, vs, part, scaling = bench.grid_for(args(), 1)
    assert dims == (512, 512, 512) and vs == 0.005 and part == 1 and scaling == "weak"
    assert bench.grid_label(dims) == "512³"


@pytest.mark.parametrize("world", [2, 4, 8])
def test_more_gpus_cut_configs3(world):
    dims, vs, part, scaling = bench.grid_for(args(), world)
    assert dims == (1024, 1024, 1024) and vs == 0.002 and part == world and scaling == "strong"
    assert dims[2] % world == 0 and (dims[0] * dims[1] * dims[2]) // world >= 512 ** 3        # N = 8: the N = 1 grid's voxels per rank
    assert bench.grid_label(dims) == "1024³"
    # the single-GPU rehearsal of one rank sees the same grid
    assert bench.grid_for(args(emulate_world=world), 1)[:2] == (dims, vs)


def test_a_named_grid_is_cut_as_it_is():
    dims, vs, part, scaling = bench.grid_for(args(grid=256, grid_given=True), 2)
    assert dims == (256, 256, 256) and vs == 0.01 and scaling == "strong"
    dims, vs, _, _ = bench.grid_for(args(grid=2048, grid_given=True, voxel_mm=2.0, workload="ssurf", emulate_world=8), 1)
    assert dims == (2048, 2048, 2048) and vs == 0.002          # BASELINE.json configs[4]


@pytest.mark.parametrize("world,want,vs_want", [(2, (512, 512, 1024), 0.0025), (4, (512, 1024, 1024), 0.0025), (8, (1024, 1024, 1024), 0.0025),
                                                (3, (512, 512, 1536), 0.005 / 3)])
def test_weak_scaling_keeps_per_rank_work(world, want, vs_want):
    dims, vs, part, scaling = bench.grid_for(args(scaling="weak"), world)
    assert dims == want and abs(vs - vs_want) < 1e-12 and scaling == "weak" and part == world
    assert dims[0] * dims[1] * dims[2] == world * 512 ** 3
    assert bench.grid_label(dims) == ("1024³" if world == 8 else f"{want[0]}×{want[1]}×{want[2]}")
