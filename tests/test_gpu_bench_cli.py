"""bench.py as the driver runs it: the one-line contract, the N-rank launch, and what a dead wire does to the exit code.

  * N = 1: one JSON line with the contract's keys, `roofline` (frac <= 1, kernel time <= step time) and `config.workload`.
  * N = 2 over gloo (two ranks share the box's one card; RCCL refuses that): `python -m torch.distributed.run` as the
    driver launches it, one line from rank 0, rc 0, `multi_gpu.world` = 2, halo + extraction ran.
  * one rank over RCCL (`--dist-world1`): the same code path -- process group on RCCL, barriers, all-reduces on device
    tensors, the halo step (no neighbour), extraction, `multi_gpu` -- on the real library.
  * a halo step that raises / never returns: the line is still printed, the process exits 4 / 3.
Short runs (--steps 5, no companion legs, no PMC passes, no CPU baseline)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
QUICK = ["--steps", "5", "--warmup", "2", "--no-extras", "--no-traffic", "--no-cpu-baseline"]


def run(cmd, env=None, timeout=300):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr.decode()[-2000:]


def test_one_gpu_line_keeps_the_contract(cuda):
    rc, lines, err = run([sys.executable, "bench.py"] + QUICK)
    assert rc == 0, err
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["unit"] == "Mvoxels/s" and d["dtype"] == "f32"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "512x512x512" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.3 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.001
    assert r["algorithmic_bytes_per_unit"] == 16 and r["units_per_launch"] == 512 ** 3
    # value = voxels x steps / time
    assert abs(d["value"] - 512 ** 3 / d["ms_per_step"] / 1e3) / d["value"] < 2e-3


def test_two_ranks_as_the_driver_launches_them(cuda):
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    rc, lines, err = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), "bench.py", "--gpus", "2", "--grid", "256", "--strong-leg"] + QUICK,
                         env={"TSDF_BENCH_BACKEND": "gloo"})
    assert rc == 0, err
    assert len(lines) == 1, "rank 0 alone prints"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    m = d["multi_gpu"]
    assert m["world"] == 2 and m["backend"] == "gloo" and m["data_path_collectives"] == 0 and m["halo_and_extraction_ran"] is True
    assert m["halo_bytes_per_boundary"] == 8 * d["config"]["grid"][0] * d["config"]["grid"][1]
    assert d["extraction_hung"] is False and d["extraction"]["vertices"] >= 0 and "error" not in d["extraction"]
    # strong scaling: the named grid itself cut in two; metric, workload and partition say what ran
    assert d["config"]["grid"] == [256, 256, 256] and d["config"]["voxel_size_m"] == 0.01
    assert "256³ grid" in d["metric"] and "256x256x256 @ 10 mm" in d["config"]["workload"]
    assert "2 z-slab(s) of 128 slices" in d["config"]["partition"] and "strong scaling" in d["config"]["partition"]
    assert m["global_grid"] == [256, 256, 256] and m["voxels_per_rank"] == 256 * 256 * 128
    assert abs(d["value"] - 256 ** 3 / d["ms_per_step"] / 1e3) / d["value"] < 2e-3
    assert 0.0 < m["hbm_frac_of_n_gpus"] <= 1.0
    assert abs(m["hbm_frac_of_n_gpus"] - m["hbm_GBps_all_ranks"] / (2 * 8000.0)) < 1e-3
    # north_star's 512^3 split beside it, and the N = 1 figure measured in the same job
    s5 = d["strong_512"]
    assert s5["grid"] == [512, 512, 512] and s5["voxel_size_m"] == 0.005 and s5["n_gpus"] == 2 and s5["scaling"] == "strong"
    assert "2 z-slab(s) of 256 slices" in s5["workload"] and s5["cache_resident"] is False and s5["slab_resident_bytes"] == 8 * 512 * 512 * 256
    assert abs(s5["value"] - 512 ** 3 / s5["ms_per_step"] / 1e3) / s5["value"] < 2e-3 and 0.0 < s5["hbm_frac_of_n_gpus"] <= 1.0
    n1 = d["n1_same_job"]
    assert "512x512x512 @ 5 mm" in n1["workload"] and n1["value"] > 0 and 0.3 < n1["hbm_frac"] <= 1.0


def test_one_rank_runs_the_multi_gpu_path_over_rccl(cuda):
    rc, lines, err = run([sys.executable, "bench.py", "--dist-world1", "--grid", "256"] + QUICK)
    assert rc == 0, err
    d = json.loads(lines[0])
    m = d["multi_gpu"]
    assert m["world"] == 1 and m["backend"].startswith("rccl") and m["comm_device"] == "cuda" and m["halo_and_extraction_ran"] is True
    assert d["extraction"]["backend"].startswith("rccl") and d["extraction_hung"] is False


@pytest.mark.parametrize("fault,code", [("raise", 4), ("hang", 3)])
def test_a_dead_wire_shows_in_the_exit_code(cuda, fault, code):
    rc, lines, err = run([sys.executable, "bench.py", "--dist-world1", "--grid", "256", "--inject-fault", fault, "--extract-deadline", "3"] + QUICK)
    assert rc == code, (rc, err)
    assert len(lines) == 1, "the line is printed all the same"
    d = json.loads(lines[0])
    assert "error" in d["extraction"] and d["extraction_hung"] is (fault == "hang")
    assert d["multi_gpu"]["halo_and_extraction_ran"] is False
    assert d["value"] > 0


@pytest.mark.parametrize("world,rank_", [(2, 1), (4, 3), (8, 0), (8, 5), (8, 7)])
def test_every_ranks_slab_of_configs3_keeps_the_headlines_promise(cuda, world, rank_):
    """The driver's N = 2, 4, 8 runs give every rank one z-slab of BASELINE.json configs[3], 1024^3 @ 2 mm.  The line's
    assertions -- every voxel of the slab updated by every frame, every TSDF value inside the truncation band -- must hold
    for ANY rank's slab, not only rank 0's: rehearsed here slab by slab on one GPU (`--emulate-world N --emulate-rank r`;
    an assertion that fails is a non-zero exit), and the line must name the grid, the voxel size and the partition it ran."""
    rc, lines, err = run([sys.executable, "bench.py", "--emulate-world", str(world), "--emulate-rank", str(rank_)] + QUICK)
    assert rc == 0, err
    d = json.loads(lines[0])
    per_rank = 1024 ** 3 // world
    assert d["roofline"]["units_per_launch"] == per_rank and 0.3 < d["roofline"]["frac"] <= 1.0
    assert d["config"]["grid"] == [1024, 1024, 1024] and d["config"]["voxel_size_m"] == 0.002 and d["scaling"] == "strong"
    assert "1024³ grid" in d["metric"] and "1024x1024x1024 @ 2 mm" in d["config"]["workload"]
    assert f"{world} z-slab(s) of {1024 // world} slices" in d["config"]["partition"] and "configs[3]" in d["config"]["partition"]
    assert f"rank {rank_} of {world}" in d["config"]["partition"]
    assert abs(d["value"] - per_rank / d["ms_per_step"] / 1e3) / d["value"] < 2e-3      # the rehearsal's value is the slab's rate


def test_weak_scaling_is_still_there_and_says_so(cuda):
    """`--scaling weak`: per-rank work fixed at the 512^3 grid, the global grid grows with N inside the same physical box."""
    rc, lines, err = run([sys.executable, "bench.py", "--scaling", "weak", "--emulate-world", "8", "--emulate-rank", "6"] + QUICK)
    assert rc == 0, err
    d = json.loads(lines[0])
    assert d["scaling"] == "weak" and d["config"]["grid"] == [1024, 1024, 1024] and d["config"]["voxel_size_m"] == 0.0025
    assert "1024x1024x1024 @ 2.5 mm" in d["config"]["workload"] and "weak scaling" in d["config"]["partition"]
    assert d["roofline"]["units_per_launch"] == 512 ** 3
