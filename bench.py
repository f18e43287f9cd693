#!/usr/bin/env python3
"""bench.py -- Mvoxels/s of the HIP TSDF Integrate path (BASELINE.json metric) with its HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 512] [--workload sband|sfull|ssurf|traj]
                    [--mode frame|fused]

One "step" = one Integrate of one 640x480 depth frame into the whole grid.  At N = 1 the workload is
BASELINE.json configs[1] -- 512^3 @ 5 mm, synthetic depth + pose stream -- as S-band (semantic_slam_amd/synth.py):
the volume wholly inside the frustum, every voxel updated by every frame AND inside the truncation band, so both
divisions run, every TSDF value changes every frame and all 16 B per voxel (TSDF + weight, read + write) have to
move: nothing about the update can be elided.  Mode "frame" (the headline) issues one kernel launch per step --
the reference's call shape, TSDF::Integrate once per frame (ref: src/tsdf.cu:135-168) -- so the SURVEY section 8(d)
byte model (16 B per updated voxel + one pass over the depth frame) is exactly what a launch must move and
roofline.frac is a true HBM fraction.  Mode "fused" hands the frames to tsdf_integrate_frames_device, which applies
up to 32 of them per pass over the volume with the voxels held in registers (bit-identical results, more voxel
updates per second, bound by instruction issue instead of HBM); it is reported beside the headline.

Timing: the depth frame(s) are resident in HBM before the timed region.  After the warm-up (at least --warmup steps
and two full launches) the K-step sequence is repeated until at least 0.3 s of device time has been timed, between
a barrier + torch.cuda.synchronize() on both sides; ms_per_step = wall / (K x repeats), max over ranks; HIP events
on the handle's stream give the kernel time.  For N > 1 every rank owns one z-slab (no data-path collective; by
default WEAK scaling: each rank's slab holds as many voxels as the whole N = 1 grid); after the timed region the
ranks run the one-voxel halo exchange (RCCL) and the halo-fed extraction once and report its time; if that step does
not come back within 120 s, or raises, the line is still printed (with `extraction_hung` / `extraction.error`) and every
rank exits non-zero (3 = hung, 4 = failed), so a dead wire shows in the record.  Launch with torch.distributed.run as
the driver does.  Rank 0 prints one JSON line.
"""
import argparse
import csv
import glob
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, HBM)
MIN_TIMED_MS = 300.0       # the K-step sequence is repeated until this much device time has been timed


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--grid", type=int, default=0, help="grid edge in voxels (default 512; 1024 for --workload traj)")
    ap.add_argument("--workload", default="sband", choices=["sband", "sfull", "ssurf", "traj"])
    ap.add_argument("--mode", default="", choices=["", "frame", "fused"],
                    help="frame: one launch per step (default for sband / sfull); fused: up to 32 frames per pass over "
                         "the volume through tsdf_integrate_frames_device (default for ssurf / traj)")
    ap.add_argument("--variant", type=int, default=-1, help="force a kernel variant (DESIGN.md; overrides --mode)")
    ap.add_argument("--scaling", default="", choices=["", "weak", "strong"],
                    help="N > 1 (default strong): strong = ONE global grid cut into N z-slabs -- by default BASELINE.json configs[3], "
                         "1024^3 @ 2 mm (at N = 8 a rank's slab holds the 134 M voxels of the N = 1 grid); --grid names another grid "
                         "to cut.  weak = every rank integrates a slab as large as the whole --grid (default 512^3) grid: the global "
                         "grid grows with N inside the same physical box (512x512x1024 @ 5 / 5 / 2.5 mm, 512x1024x1024, 1024^3 @ 2.5 mm)")
    ap.add_argument("--dist-world1", action="store_true",
                    help="with one rank: still create the process group and run the N > 1 code path (barriers, all-reduces, halo step, "
                         "multi_gpu object) -- the RCCL rehearsal a one-GPU box allows (tests/test_gpu_bench_cli.py)")
    ap.add_argument("--extract-deadline", type=float, default=120.0, help="seconds the post-run halo + extraction step may take")
    ap.add_argument("--inject-fault", choices=["none", "raise", "hang"], default="none",
                    help="tests only: make the halo + extraction step fail / never return (the line must still be printed and the "
                         "exit code must say so: 4 / 3)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single-GPU rehearsal of one rank of an N-GPU job: integrate only rank 0's z-slab of N")
    ap.add_argument("--emulate-rank", type=int, default=0, help="with --emulate-world N: which rank's z-slab (default 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the contract's timed region (for profiler runs: no companion legs, which launch more kernels)")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not measure roofline.traffic (two short rocprofv3 --pmc child runs of this script)")
    ap.add_argument("--no-extract", action="store_true", help="N > 1: skip the halo exchange + extraction after the timed region")
    ap.add_argument("--noise-mm", type=float, default=0.0, help="ssurf / traj: zero-mean Gaussian depth noise, sigma in mm (SURVEY.md 8d), rng seed 1234")
    ap.add_argument("--holes", type=float, default=0.0, help="ssurf / traj: fraction of the image lost to invalid (zero) 8 x 8 pixel blocks, like a real sensor's dropouts")
    ap.add_argument("--voxel-mm", type=float, default=0.0, help="voxel size in mm (default: 5 at 512, 2 at 1024, else 2560 / grid)")
    ap.add_argument("--tum-dir", default="", help="traj: directory of the TUM fr3_office sequence (holding depth/<stamp>.png as "
                    "result/rgbd/associations.txt names them); when it exists the keyframes' real 16-bit depth frames are integrated "
                    "(value / 5000 m) instead of the rendered ones.  No dataset ships with the repository and none is fetched.")
    ap.add_argument("--labels", type=int, default=0,
                    help="ssurf: per-voxel label fusion in the same passes (tsdf_integrate_frames_labels_device, BASELINE.json configs[4]) "
                         "with this many synthetic instance masks per frame in MaskRCNN's output format (uint8 {0,255}, label 1..80, "
                         "score in (0.8, 1]), composed on the device.  configs[4] as one rank sees it: --workload ssurf --grid 2048 "
                         "--voxel-mm 2 --emulate-world 8 --emulate-rank 1 --labels 6")
    ap.add_argument("--strong-leg", action="store_true",
                    help="N > 1 (or --dist-world1): run the strong_512 / n1_same_job legs even with --no-extras (tests)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--depth-cache", default="", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    a = ap.parse_args()
    a.grid_given = bool(a.grid)
    if not a.grid:
        a.grid = 1024 if a.workload == "traj" else 512
    if not a.mode:
        a.mode = "frame" if a.workload in ("sband", "sfull") else "fused"
    return a


# ------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------
class Workload:
    """origin, truncation margin, base pose, the cam2world stream and the depth frame(s) it comes with."""

    def __init__(self, name, dims, vs, noise_mm=0.0, holes=0.0, tum_dir="", cache_dir=""):
        from semantic_slam_amd import synth
        self.name, self.dims, self.vs = name, dims, vs
        # rendered frames are kept for the profiled child runs of this script (a rendering pass per child would cost more than
        # the child): <cache_dir>/<workload>_<grid>_<voxel>_<noise>_<holes>.npy, float32 [frames, H, W]
        cache = (os.path.join(cache_dir, f"tsdf_bench_{name}_{dims[0]}x{dims[1]}x{dims[2]}_{vs:g}_{noise_mm:g}_{holes:g}_{int(bool(tum_dir))}.npy")
                 if cache_dir and name in ("ssurf", "traj") else "")
        cached = np.load(cache, allow_pickle=False) if cache and os.path.isfile(cache) else None
        self.trunc = None            # None = the reference's 5 x voxel
        self.base2world = None
        self.full_coverage = False
        if name in ("sband", "sfull"):
            self.origin = synth.sband_volume(dims, vs) if name == "sband" else synth.sfull_volume(dims, vs)
            pose = synth.sband_pose if name == "sband" else synth.sfull_pose
            self.trunc = synth.SBAND_TRUNC if name == "sband" else None
            self.poses = np.stack([pose(k) for k in range(64)])
            self.depths = [synth.sfull_depth()]                      # one constant frame, shared by every pose
            self.full_coverage = True
            self.desc = ("every voxel updated every frame inside the truncation band (dist < 1, both divisions, every TSDF "
                         "value changes): all 16 B per voxel move" if name == "sband" else
                         "every voxel updated every frame with dist = 1 (free space)")
        elif name == "ssurf":
            self.origin = synth.surf_volume(max(dims), vs, 1.0)
            scene = synth.SurfScene(dims, vs, self.origin)
            self.poses = np.stack([scene.pose(k, 64) for k in range(64)])
            self.depths = list(cached) if cached is not None else [scene.depth(p, quantize=True) for p in self.poses]   # re-rendered per pose
            self.desc = "sphere + wall seen from a 64-pose orbit, depth re-rendered per pose and quantised at 1/5000 m"
        else:   # traj: BASELINE configs[2]
            from semantic_slam_amd import ingest
            gold = os.path.join(ROOT, "tests", "golden", "fr3_office_keyframes.npz")
            Twc = ingest.pose_inverse(np.load(gold, allow_pickle=False)["Tcw"])
            self.base2world = Twc[0].ravel().copy()
            half = max(dims[0], dims[1]) * vs / 2.0
            self.origin = np.array([-half, -half, 0.6], np.float32)
            scene = synth.SurfScene(dims, vs, self.origin)
            from semantic_slam_amd import capi
            ok, binv = capi.invert_matrix(self.base2world)
            self.poses = np.stack([T.ravel() for T in Twc])
            # what the camera sees from each keyframe: rendered from the relative pose the library will compose
            names = [str(x) for x in np.load(gold, allow_pickle=False)["depth_names"]]
            if cached is not None:
                self.depths = list(cached)
                self.desc = "(frames of the parent run)"
            elif tum_dir and all(os.path.isfile(os.path.join(tum_dir, nm)) for nm in names):
                # the sequence is there: the keyframes' own depth frames, raw 16-bit / 5000 (config/TUM3.yaml:34)
                scale = np.float32(1.0) / np.float32(5000.0)
                self.depths = [np.ascontiguousarray(ingest.load_depth_png(os.path.join(tum_dir, nm)).astype(np.float32) * scale) for nm in names]
                self.desc = (f"the {len(self.poses)} keyframes of the reference's saved fr3_office run (result/rgbd/bundle.txt, "
                             f"associations.txt): poses from the bundle file, base = first keyframe, REAL depth frames from {tum_dir}")
            else:
                self.depths = [scene.depth(capi.multiply_matrix(binv, p), quantize=True) for p in self.poses]
                self.desc = (f"the {len(self.poses)} keyframe poses of the reference's saved fr3_office run (result/rgbd/bundle.txt), "
                             "base = first keyframe, sphere + wall depth re-rendered per pose, quantised at 1/5000 m"
                             + ("" if not tum_dir else f" (--tum-dir {tum_dir} does not hold the named depth frames)"))
        if cached is None and name in ("ssurf", "traj") and (noise_mm > 0 or holes > 0):
            # a sensor's imperfections (Gaussian noise, sigma 2 mm; dropouts as zero blocks), then the
            # 1/5000 m quantisation again
            self.depths = synth.sensor_imperfections(self.depths, noise_mm, holes)
            self.desc += f"; Gaussian noise sigma {noise_mm:g} mm, {holes * 100:g} % of the image dropped in 8 x 8 blocks"
        self.depths = [np.ascontiguousarray(d, np.float32) for d in self.depths]
        if cache and cached is None:
            np.save(cache, np.stack(self.depths))
        self.n_pose = len(self.poses)

    def block(self, start, n):
        idx = [(start + i) % self.n_pose for i in range(n)]
        return self.poses[idx], idx


def default_voxel(D):
    return {512: 0.005, 1024: 0.002}.get(D, 2.56 / D)


def grid_for(args, world):
    """Global dims, voxel size, the number of slabs the grid is cut into and the scaling label of the line.

    N = 1: BASELINE.json configs[1], 512^3 @ 5 mm.  N > 1, default ("strong"): ONE grid cut into N z-slabs -- BASELINE.json
    configs[3], 1024^3 @ 2 mm, unless --grid names another grid to cut.  "weak": per-rank work fixed at the --grid grid,
    the global grid grows with N inside the same physical box."""
    part_world = args.emulate_world if (args.emulate_world > 1 and world == 1) else world
    scaling = args.scaling or ("strong" if part_world > 1 else "weak")     # at N = 1 the label says nothing; "weak" as before
    D = args.grid
    if part_world > 1 and scaling == "strong" and not args.grid_given and args.workload != "traj":
        D = 1024                                   # configs[3]
    vs = default_voxel(D)
    if args.voxel_mm > 0:
        vs = args.voxel_mm / 1000.0
    dims = [D, D, D]
    if scaling == "weak" and part_world > 1:
        # per-rank work fixed: z, then y, then x doubled in turn, voxels halved whenever z is doubled; a rank count that
        # is not a power of two just gets N x the slices
        k, n = 0, part_world
        while n % 2 == 0:
            n //= 2
            k += 1
        if n == 1:
            for i in range(k):
                dims[2 - i % 3] *= 2
            vs = vs / 2 ** ((k + 2) // 3)
        else:
            dims[2] *= part_world
            vs = vs / part_world
    return tuple(dims), vs, part_world, scaling


def grid_label(dims):
    return f"{dims[0]}³" if dims[0] == dims[1] == dims[2] else f"{dims[0]}×{dims[1]}×{dims[2]}"


# ------------------------------------------------------------------------------------------------------------
# CPU baseline (checker code timed as the baseline; never used to produce the GPU result)
# ------------------------------------------------------------------------------------------------------------
def cpu_baseline(args, W, cam_K, trunc):
    from oracle.oracle import Oracle, Ref
    orc = Oracle()
    dims, vs, origin = W.dims, W.vs, W.origin
    Dz = dims[2]
    nz = min(Dz, 64)                      # sample: the middle nz slices of the grid
    zb = (Dz - nz) // 2
    threads = orc.max_threads()
    base = W.base2world if W.base2world is not None else np.eye(4, dtype=np.float32).ravel()
    c2b = [orc.cam2base(base, p) for p in W.poses]
    depth = lambda i: W.depths[i % len(W.depths)]
    t, w = orc.init_grid(dims, zb, zb + nz)
    orc.integrate(cam_K, c2b[0], depth(0), dims, origin, vs, trunc, t, w, z_begin=zb, z_end=zb + nz, threads=threads)  # page in
    frames, t0, n_upd = 0, time.perf_counter(), 0
    while True:
        i = (frames + 1) % W.n_pose
        n_upd += orc.integrate(cam_K, c2b[i], depth(i), dims, origin, vs, trunc, t, w, z_begin=zb, z_end=zb + nz, threads=threads)
        frames += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or frames >= 200:
            break
    per_slice = dims[0] * dims[1]
    out = {"value": round(nz * per_slice * frames / el / 1e6, 1), "unit": "Mvoxels/s", "cores": threads, "kind": "port",
           "sample": f"oracle/tsdf_oracle.c (OpenMP over rows), {frames} frames of the same workload into z-slab "
                     f"[{zb},{zb + nz}) of the {dims[0]}x{dims[1]}x{dims[2]} grid, {el:.1f} s"}
    if W.full_coverage:
        assert n_upd == frames * nz * per_slice, "full-coverage workload must update every voxel (N_upd == N)"
    # the same at one thread (BASELINE.md section 4 asks for both), on an 8-slice slab, a few seconds
    n1 = min(Dz, 8)
    t1, w1 = orc.init_grid(dims, zb, zb + n1)
    orc.integrate(cam_K, c2b[0], depth(0), dims, origin, vs, trunc, t1, w1, z_begin=zb, z_end=zb + n1, threads=1)
    f1, t0 = 0, time.perf_counter()
    while True:
        i = (f1 + 1) % W.n_pose
        orc.integrate(cam_K, c2b[i], depth(i), dims, origin, vs, trunc, t1, w1, z_begin=zb, z_end=zb + n1, threads=1)
        f1 += 1
        el1 = time.perf_counter() - t0
        if el1 > 3.0 or f1 >= 50:
            break
    out["one_thread"] = {"value": round(n1 * per_slice * f1 / el1 / 1e6, 1), "unit": "Mvoxels/s", "cores": 1,
                         "sample": f"{f1} frames into an {n1}-slice slab, {el1:.1f} s"}
    ref = None
    if Ref.available() and max(dims) <= 1024:
        # the reference body has no slab form: give it a grid that IS the slab (origin shifted in z on the host;
        # timing only, values are not compared here)
        r = Ref()
        rd = (dims[0], dims[1], nz)
        ro = np.array([origin[0], origin[1], origin[2] + zb * vs], np.float32)
        t2, w2 = orc.init_grid(rd)
        r.integrate(cam_K, c2b[0], depth(0), rd, ro, vs, trunc, t2, w2, threads=threads)
        f2, t0 = 0, time.perf_counter()
        while True:
            i = (f2 + 1) % W.n_pose
            r.integrate(cam_K, c2b[i], depth(i), rd, ro, vs, trunc, t2, w2, threads=threads)
            f2 += 1
            el2 = time.perf_counter() - t0
            if el2 > args.cpu_seconds / 2 or f2 >= 200:
                break
        ref = {"value": round(nz * per_slice * f2 / el2 / 1e6, 1), "unit": "Mvoxels/s", "cores": threads, "kind": "reference",
               "sample": f"GpuIntegrate of the reference (src/tsdf.cu:15-60) built for the host by oracle/Makefile, "
                         f"{f2} frames into a {rd[0]}x{rd[1]}x{rd[2]} grid, {el2:.1f} s"}
    return out, ref


# ------------------------------------------------------------------------------------------------------------
# roofline.traffic, measured by this run: two short rocprofv3 --pmc passes over a child run of this script
# ------------------------------------------------------------------------------------------------------------
_PMC_BROKEN = None     # reason of the first failed pass: the remaining passes of the run are skipped (a profiler that does not work
                       # here must cost the line one time-out, not one per pass)


def _pmc_pass(counters, kernel_substr, child_args, keep, timeout=150):
    """One `rocprofv3 --pmc <counters>` pass over a short child run of this script (`--pmc-child`: the launches only).
    Returns ({counter: (mean over the kernel's last `keep` dispatches -- the child's timed part; its warm-up launch may have
    taken another kernel -- , dispatches)}, None) or (None, reason)."""
    global _PMC_BROKEN
    if _PMC_BROKEN is not None:
        return None, "skipped after an earlier pass failed: " + _PMC_BROKEN
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        _PMC_BROKEN = "rocprofv3 not found"
        return None, _PMC_BROKEN
    tmp = os.environ.get("TMPDIR") or "/tmp"
    d = tempfile.mkdtemp(prefix="bench_pmc_", dir=tmp)
    cmd = [exe, "--pmc"] + list(counters) + ["-d", d, "-o", "p", "--output-format", "csv", "--",
                                             sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child"] + child_args
    try:
        env = dict(os.environ, TMPDIR=tmp)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        p = subprocess.run(cmd, cwd=tmp, env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            _PMC_BROKEN = f"rocprofv3 --pmc {' '.join(counters)} failed (rc {p.returncode}): {p.stderr.decode(errors='replace')[-200:]}"
            return None, _PMC_BROKEN
        out = {}
        rows = [r for r in csv.DictReader(open(files[0])) if kernel_substr in r["Kernel_Name"]]
        if rows:
            out["kernel_name"] = rows[-1]["Kernel_Name"]      # as the profiler prints it: the name profiles/*_kernel_stats.csv carries
        for c in counters:
            v = [float(r["Counter_Value"]) for r in sorted((r for r in rows if r["Counter_Name"] == c), key=lambda r: int(r["Dispatch_Id"]))]
            v = v[-keep:]                # without the child's warm-up launches (they also settle the per-launch decisions)
            if not v:
                return None, f"no dispatch of {kernel_substr} in the {c} pass"
            out[c] = (float(np.mean(v)), len(v))
        return out, None
    except Exception as e:   # noqa: BLE001 -- the counters are optional, the bench line is not
        _PMC_BROKEN = f"rocprofv3 --pmc {' '.join(counters)}: {e!r}"[:300]
        return None, _PMC_BROKEN
    finally:
        shutil.rmtree(d, ignore_errors=True)


CHILD_WARMUP = 2


def child_args_for(args, workload=None, mode=None, variant=None, grid=None, steps=64, noise_mm=None, holes=None):
    a = ["--workload", workload or args.workload, "--grid", str(grid or args.grid), "--mode", mode or args.mode,
         "--variant", str(args.variant if variant is None else variant), "--steps", str(steps), "--warmup", str(CHILD_WARMUP),
         "--noise-mm", str(args.noise_mm if noise_mm is None else noise_mm), "--holes", str(args.holes if holes is None else holes)]
    if args.tum_dir:
        a += ["--tum-dir", args.tum_dir]
    if args.depth_cache:
        a += ["--depth-cache", args.depth_cache]
    if args.labels and (workload or args.workload) == args.workload:
        a += ["--labels", str(args.labels)]
    if args.emulate_world > 1 and grid is None:
        a += ["--emulate-world", str(args.emulate_world), "--emulate-rank", str(args.emulate_rank)]
    if args.voxel_mm > 0 and grid is None:
        a += ["--voxel-mm", str(args.voxel_mm)]
    return a


def _child_shape(child_args, fpl):
    """(dispatches of the kernel in the timed part of a child run, frames they integrate)."""
    steps = int(child_args[child_args.index("--steps") + 1])
    return (steps + fpl - 1) // fpl, steps


def measure_traffic(kernel_substr, child_args, fpl=1, wide_reads=True):
    """HBM bytes per launch of the dominant kernel from the PMC counters, collected as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE and WRITE_SIZE in separate passes (they do not fit one), KiB units, FETCH_SIZE doubled on gfx950 (it tallies
    the 128-byte requests of wide coalesced reads at 64 bytes), WRITE_SIZE as read.  fpl > 1: the child integrates `--steps`
    frames in passes of up to fpl, and the figure is the child's bytes per fpl frames (the launches of a trajectory differ
    widely: the mean over the whole sequence is what the timed mean launch is to be compared with).  Returns (bytes or None, note)."""
    keep, steps = _child_shape(child_args, fpl)
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        r, why = _pmc_pass([counter], kernel_substr, child_args, keep)
        if r is None:
            return None, why
        vals[counter] = r[counter]
        PMC_KERNEL_NAMES[kernel_substr] = r["kernel_name"]
    scale = 1.0 if fpl == 1 else vals["FETCH_SIZE"][1] * fpl / float(steps)      # dispatches -> launches of fpl frames
    fetch, write = 2.0 * vals["FETCH_SIZE"][0] * 1024.0 * scale, vals["WRITE_SIZE"][0] * 1024.0 * scale
    note = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes over a short child run of "
            f"the same workload), mean over {vals['FETCH_SIZE'][1]} / {vals['WRITE_SIZE'][1]} dispatches of the kernel"
            + (f" ({steps} consecutive frames, scaled to {fpl} frames)" if fpl > 1 else "") + "; KiB units, "
            f"FETCH_SIZE x 2 (gfx950 counts wide coalesced reads at half their bytes), WRITE_SIZE as read: "
            f"read {fetch / 1e6:.1f} MB + written {write / 1e6:.1f} MB per launch")
    if not wide_reads:
        note += ("; this kernel reads the volume in 32-byte row pieces (buffer loads, merged into whole lines in L2) and gathers "
                 "4-byte depth samples -- the x 2 is calibrated on the all-free-space launch, whose bytes are known (DESIGN.md section 5)")
    return fetch + write, note


PMC_KERNEL_NAMES = {}          # kernel substring -> the kernel's full name as the profiled child runs saw it

N_SIMDS = 1024                 # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)


VALU_PEAK_CYCLES_PER_INST = 2.0   # a SIMD issues at most one wave64 VALU instruction per 2 cycles (32 lanes per cycle: MI355X_MICROARCH.md,
                                  # "v_fma_f32 (wave64) 2 cyc"; 1024 SIMDs x 32 lanes x 2 flop x 2.4 GHz = the 157 TFLOP/s fp32 vector peak)


def measure_valu(kernel_substr, child_args, fpl=32):
    """What an issue-bound launch is measured against, from counters and one hardware constant (no modelled instruction price):
    valu_issue_frac = SQ_INSTS_VALU (wave instructions) x 2 cycles -- the SIMDs' peak issue rate -- / (GRBM_GUI_ACTIVE / 8 XCDs
    x 1024 SIMDs): the share of the chip's VALU issue slots the launch used, <= 1 by construction.  (No counter of this chip gives
    "cycles the VALU was busy": SQ_ACTIVE_INST_VALU sums, per wavefront, the quad-cycles it spends in VALU instructions, and the
    instructions of a SIMD's wavefronts overlap in its pipeline -- that quotient read 0.92 on S-surf and 1.01 on the trajectory.  What
    the instruction mix of these kernels costs per instruction -- compares, selects, conversions, packed operations and reciprocals
    issue slower than a plain multiply-add -- is measured by tools/microbench/valu_rate.hip: 3.5 cycles on average)
    Beside it, how the resident wavefronts spent their time: executing, waiting for memory / barriers, stalled at issue."""
    r, why = _pmc_pass(["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
                        "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"], kernel_substr, child_args, _child_shape(child_args, fpl)[0])
    if r is None:
        return None, why
    PMC_KERNEL_NAMES[kernel_substr] = r["kernel_name"]
    cyc = r["GRBM_GUI_ACTIVE"][0] / 8.0
    if cyc <= 0:
        return None, "GRBM_GUI_ACTIVE read 0"
    steps = _child_shape(child_args, fpl)[1]
    per = r["SQ_INSTS_VALU"][1] * fpl / float(steps) if fpl > 1 else 1.0      # dispatches -> launches of fpl frames
    wave_cyc = max(r["SQ_WAVE_CYCLES"][0], 1.0)
    return {"valu_issue_frac": round(r["SQ_INSTS_VALU"][0] * VALU_PEAK_CYCLES_PER_INST / (cyc * N_SIMDS), 4),
            "valu_peak_cycles_per_inst": VALU_PEAK_CYCLES_PER_INST,
            "valu_insts_per_launch": int(r["SQ_INSTS_VALU"][0] * per), "salu_insts_per_launch": int(r["SQ_INSTS_SALU"][0] * per),
            "salu_per_valu": round(r["SQ_INSTS_SALU"][0] / max(r["SQ_INSTS_VALU"][0], 1.0), 3),
            "waves_per_launch": int(r["SQ_WAVES"][0] * per), "kernel_cycles": int(cyc * per),
            "mean_waves_per_simd": round(r["SQ_WAVE_CYCLES"][0] * 4.0 / (cyc * N_SIMDS), 2),
            "wave_time": {"executing": round(r["SQ_ACTIVE_INST_ANY"][0] / wave_cyc, 4), "waiting_memory_or_barrier": round(r["SQ_WAIT_ANY"][0] / wave_cyc, 4),
                          "stalled_at_issue": round(r["SQ_WAIT_INST_ANY"][0] / wave_cyc, 4)},
            "source": "bench.py measure_valu: rocprofv3 --pmc counters of a short child run; the 2-cycle peak from MI355X_MICROARCH.md",
            "note": "one rocprofv3 --pmc pass over a short child run, means over "
                    f"{r['SQ_INSTS_VALU'][1]} dispatches ({steps} consecutive frames; per-launch figures scaled to {fpl} frames): "
                    "valu_issue_frac = SQ_INSTS_VALU x 2 cycles (peak issue rate of a SIMD) / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), <= 1 by "
                    "construction; wave_time = SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES (disjoint shares of the "
                    "resident wavefronts' time)"}, None


def main():
    args = parse()
    import torch
    from semantic_slam_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs: python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # TSDF_BENCH_BACKEND=gloo: rehearsal of the N-rank path on a box with fewer GPUs than ranks (ranks share
    # devices, buffers travel through gloo on the host); the driver's runs use RCCL.
    backend = os.environ.get("TSDF_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or args.dist_world1
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "RANK" not in os.environ:      # --dist-world1 without a launcher
            import socket
            so = socket.socket()
            so.bind(("127.0.0.1", 0))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local_rank), MASTER_PORT=str(so.getsockname()[1]))
            so.close()
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    comm_dev = "cpu" if backend == "gloo" else "cuda"

    dims, vs, part_world, scaling = grid_for(args, world)
    cache_dir = args.depth_cache
    own_cache = None
    if not cache_dir and not args.no_traffic and not args.pmc_child and world == 1:
        own_cache = cache_dir = tempfile.mkdtemp(prefix="bench_frames_", dir=os.environ.get("TMPDIR") or "/tmp")
    W = Workload(args.workload, dims, vs, args.noise_mm, args.holes, args.tum_dir, cache_dir)
    D = args.grid

    # z-slab of this rank (ref layout is z-major, so a slab is one contiguous range)
    Dz = dims[2]
    zb, ze = rank * Dz // world, (rank + 1) * Dz // world
    if args.emulate_world > 1 and world == 1:
        assert 0 <= args.emulate_rank < args.emulate_world
        zb, ze = args.emulate_rank * Dz // args.emulate_world, (args.emulate_rank + 1) * Dz // args.emulate_world
    n_global = dims[0] * dims[1] * dims[2]
    cfg = capi.make_config(dims, vs, W.origin, trunc=W.trunc, base2world=W.base2world, z_begin=zb, z_end=ze, device=local_rank)
    vol = capi.Volume(cfg)
    variant = args.variant if args.variant >= 0 else (3 if args.mode == "frame" else 0)
    vol.set_kernel_variant(variant)
    fpl = vol.frames_per_launch
    d_dev = [torch.from_numpy(d).cuda() for d in W.depths]
    H, Wd = W.depths[0].shape

    # ---- per-voxel label fusion in the same passes (--labels K; BASELINE.json configs[4]) --------------------------------------
    lab_dev = sc_dev = None
    if args.labels > 0:
        assert args.workload == "ssurf" and fpl > 1, "--labels rides on the fused S-surf sequence"
        vol.labels_enable(0.5)
        rng = np.random.default_rng(9)
        lab_dev, sc_dev = [], []
        for k in range(len(d_dev)):
            # K instance rectangles that drift across the image with the orbit (the scene's sphere keeps the middle of it)
            masks = np.zeros((args.labels, H, Wd), np.uint8)
            for m in range(args.labels):
                y0 = int((0.08 + 0.11 * m) * H + 6 * math.sin(0.2 * k + m)) % max(H - 120, 1)
                x0 = int((0.05 + 0.13 * m) * Wd + 9 * math.cos(0.15 * k + m)) % max(Wd - 160, 1)
                masks[m, y0:y0 + H // 3 + 17 * m, x0:x0 + Wd // 3 + 23 * m] = 255
            labels_k = ((np.arange(args.labels) * 7 + (k // 16)) % 80 + 1).astype(np.uint16)       # classes change along the sequence
            scores_k = rng.uniform(0.8, 1.0, args.labels).astype(np.float32)
            m_dev = torch.from_numpy(masks).cuda()
            l_d = torch.empty((H, Wd), dtype=torch.uint16, device="cuda")
            s_d = torch.empty((H, Wd), dtype=torch.float32, device="cuda")
            vol.compose_labels(m_dev.data_ptr(), labels_k, scores_k, l_d.data_ptr(), s_d.data_ptr())
            vol.sync()
            lab_dev.append(l_d)
            sc_dev.append(s_d)
            del m_dev
        lab_stream = torch.cuda.Stream()
        vol.set_stream(lab_stream.cuda_stream)       # so that torch events on this stream bracket the library's launches

    def run_block_labels(v, start, n):
        poses, idx = W.block(start, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(lab_stream)
        v.integrate_frames_labels_device([d_dev[i].data_ptr() for i in idx], [lab_dev[i].data_ptr() for i in idx],
                                         [sc_dev[i].data_ptr() for i in idx], poses)
        e1.record(lab_stream)
        e1.synchronize()
        return e0.elapsed_time(e1)

    def run_block(v, start, n, Wl=None, ddev=None):
        """n consecutive steps queued back to back on the handle's stream; device milliseconds (HIP events there)."""
        if args.labels > 0 and v is vol and Wl is None:
            return run_block_labels(v, start, n)
        Wl, ddev = Wl or W, ddev or d_dev
        poses, idx = Wl.block(start, n)
        if len(ddev) == 1:
            return v.integrate_sequence_timed(ddev[0].data_ptr(), poses)
        return v.integrate_frames_timed([ddev[i].data_ptr() for i in idx], poses)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    args.depth_cache = cache_dir      # what the child runs are told
    if args.pmc_child:   # profiled child of measure_traffic: the launches only
        run_block(vol, W.n_pose - args.warmup % W.n_pose, args.warmup)     # warm-up on the sequence's last poses ...
        run_block(vol, 0, args.steps)                                      # ... then the sequence from its start, as the timed legs run it
        vol.close()
        return

    on_dev = "cpu" if backend == "gloo" else "cuda"

    def timed_region(v, K, Wl=None, ddev=None, min_ms=MIN_TIMED_MS):
        """The contract's timed region on handle v: warm-up (at least --warmup steps, two full launches, 8 steps), a calibration
        block, then the K-step sequence `repeats` times between barrier + synchronize fences.  Returns (wall seconds and kernel
        milliseconds, both max over ranks; steps timed; repeats; frames integrated into v so far; warm-up steps)."""
        Wl, ddev = Wl or W, ddev or d_dev
        fpl_ = v.frames_per_launch
        n_warm = max(args.warmup, 2 * fpl_, 8)
        own = (None, None) if (Wl is W and ddev is d_dev) else (Wl, ddev)
        run_block(v, 0, n_warm, *own)
        cal_ms = run_block(v, n_warm, K, *own)         # calibration block (also warm): how long K steps take
        done = n_warm + K
        repeats = max(1, int(math.ceil(1.2 * min_ms / max(cal_ms, 1e-3))))   # 20 % margin: the calibration block may run slow
        if dist is not None:   # every rank times the same number of blocks
            rt = torch.tensor([repeats], dtype=torch.int64, device=on_dev)
            dist.all_reduce(rt, op=dist.ReduceOp.MAX)
            repeats = int(rt[0])
        # one call: the repeats' launches are queued back to back on the handle's stream (a call per repeat left the device
        # idle for ~0.4 ms between repeats while the host drained, returned and came back: 5 % at --steps 20); with a depth frame
        # per pose the call's arguments (K x repeats device pointers and poses) are marshalled ahead of the timed region
        timed_call = None
        if len(ddev) > 1 and not (args.labels > 0 and v is vol):
            poses_t, idx_t = Wl.block(done, K * repeats)
            timed_call = v.frames_timed_call([ddev[i].data_ptr() for i in idx_t], poses_t)
        fence()
        t0 = time.perf_counter()
        k_ms = timed_call() if timed_call is not None else run_block(v, done, K * repeats, None if Wl is W else Wl, None if ddev is d_dev else ddev)
        fence()
        wall_ = time.perf_counter() - t0
        tt = torch.tensor([wall_, k_ms], dtype=torch.float64, device=on_dev)
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt[0]), float(tt[1]), K * repeats, repeats, done + K * repeats, n_warm

    def check_full_coverage(v, Wl, done, what):
        """The promise of the S-band / S-full workloads, asserted on the device's result: every voxel of the slab updated by
        every frame (and, S-band, every TSDF value inside the truncation band).  Returns (updates per frame, slab voxels)."""
        t_h, w_h = v.download()
        upd = float(w_h.astype(np.float64).sum())       # each update adds exactly 1 to one weight
        if Wl.full_coverage:
            assert w_h.min() == w_h.max() == float(done), f"{what}: full coverage means every voxel every frame"
            if Wl.name == "sfull":
                assert np.all(t_h == 1.0)
            else:
                assert 0.0 < t_h.min() and t_h.max() < 1.0, f"{what}: S-band keeps every TSDF value inside the truncation band"
        return upd / done, v.n_voxels

    K = args.steps
    wall, kernel_ms_total, timed_steps, repeats, frames_done, n_warm = timed_region(vol, K)

    # ---- what was integrated: count updates on this rank's slab ------------------------------------------------
    n_upd_per_frame, n_slab = check_full_coverage(vol, W, frames_done, "headline")
    # several frames per pass over a scene with free / unseen space: what a launch touches (its voxels updated by at least one
    # of its frames) is counted, not bounded -- every launch of one pass over the step sequence on a fresh volume, weights > 0
    # counted on the device; per fpl frames, like the measured traffic it is compared with
    touched_per_launch, labelled_per_launch = None, 0.0
    if fpl > 1 and not W.full_coverage and world == 1:
        nz_s = ze - zb
        t_dev = torch.empty(n_slab, dtype=torch.float32, device="cuda")
        w_dev = torch.empty(n_slab, dtype=torch.float32, device="cuda")
        seq = W.n_pose if W.n_pose >= 2 * fpl else 3 * fpl        # the frames the traffic children run
        total = labelled = 0
        lab_host = np.empty(n_slab, np.uint16) if args.labels > 0 else None
        for start in range(0, seq, fpl):
            vol.reset()
            if args.labels > 0:
                vol.labels_enable(0.5)           # clears the label state
            run_block(vol, start, min(fpl, seq - start))
            vol.copy_slices_to_device(0, nz_s, t_dev.data_ptr(), w_dev.data_ptr())
            total += int(torch.count_nonzero(w_dev))
            if args.labels > 0:
                capi.check(vol.lib.tsdf_download_labels(vol._h, lab_host.ctypes.data, None, None), "tsdf_download_labels")
                labelled += int(np.count_nonzero(lab_host))
        touched_per_launch = total * fpl / float(seq)
        labelled_per_launch = labelled * fpl / float(seq)
        del t_dev, w_dev, lab_host

    # ---- N > 1: north_star's "512^3 grid at 1, 2, 4 and 8 GPUs" beside the headline, and the N = 1 figure both scale from ---
    strong_512 = n1_here = None
    frame_bytes_ = 4.0 * H * Wd + 100.0
    if multi and (args.strong_leg or not args.no_extras) and args.workload == "sband" and args.variant < 0 and args.emulate_world <= 1:
        sd, svs = (512, 512, 512), 0.005
        Ws = Workload("sband", sd, svs)
        s_dev = [torch.from_numpy(Ws.depths[0]).cuda()]
        szb, sze = rank * sd[2] // world, (rank + 1) * sd[2] // world
        with capi.Volume(capi.make_config(sd, svs, Ws.origin, trunc=Ws.trunc, z_begin=szb, z_end=sze, device=local_rank)) as sv_:
            sv_.set_kernel_variant(3)
            s_wall, s_kms, s_steps, s_rep, s_done, _ = timed_region(sv_, K, Ws, s_dev, min_ms=150.0)
            check_full_coverage(sv_, Ws, s_done, "strong_512")
            slab_bytes = 8 * sv_.n_voxels
        n_s = sd[0] * sd[1] * sd[2]
        s_ms = s_wall / s_steps * 1e3
        strong_512 = {
            "workload": f"sband 512x512x512 @ 5 mm cut into {world} z-slab(s) of {sze - szb} slices, one launch per step per rank "
                        "(BASELINE.json north_star: \"a 512^3 grid at 1, 2, 4 and 8 GPUs\"; strong scaling)",
            "grid": list(sd), "voxel_size_m": svs, "n_gpus": world, "scaling": "strong",
            "ms_per_step": round(s_ms, 5), "value": round(n_s / s_ms / 1e3, 1), "unit": "Mvoxels/s", "steps_timed": s_steps,
            "hbm_GBps_all_ranks": round((16.0 * n_s + world * frame_bytes_) / (s_ms * 1e-3) / 1e9, 1),
            "hbm_frac_of_n_gpus": round((16.0 * n_s + world * frame_bytes_) / (s_ms * 1e-3) / 1e9 / (world * HBM_PEAK_GBS), 4),
            "slab_resident_bytes": int(slab_bytes), "cache_resident": bool(slab_bytes < 256 * 2 ** 20),
            "note": "same fences and max-over-ranks timing as the headline; cache_resident: a rank's slab (TSDF + weight) is smaller "
                    "than the 256 MB Infinity Cache, so its 16 B per voxel need not all reach HBM -- hbm_frac_of_n_gpus is then the "
                    "algorithmic rate against N x 8 TB/s, not an HBM measurement"}
        del s_dev
        if rank == 0:
            # the N = 1 configuration (configs[1]: the whole 512^3 grid on one GPU) on THIS node's first GPU, while the other
            # ranks wait at the next fence: the figure the N-GPU values scale from, measured in the same job
            with capi.Volume(capi.make_config(sd, svs, Ws.origin, trunc=Ws.trunc, device=local_rank)) as nv:
                nv.set_kernel_variant(3)
                n_dev = [torch.from_numpy(Ws.depths[0]).cuda()]
                run_block(nv, 0, 40, Ws, n_dev)
                tot, st = 0.0, 0
                t1 = time.perf_counter()
                while tot < 150.0:
                    tot += run_block(nv, 40 + st, 64, Ws, n_dev)
                    st += 64
                torch.cuda.synchronize()
                n1_wall = (time.perf_counter() - t1) / st * 1e3
                n1_here = {"workload": "sband 512x512x512 @ 5 mm on one GPU (rank 0 alone; BASELINE.json configs[1], the N = 1 line's workload)",
                           "ms_per_step": round(n1_wall, 5), "kernel_ms": round(tot / st, 5), "value": round(n_s / n1_wall / 1e3, 1),
                           "unit": "Mvoxels/s", "hbm_frac": round((16.0 * n_s + frame_bytes_) / (tot / st * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "steps_timed": st}
        fence()

    # ---- after the timed region, N > 1: the one-voxel halo (RCCL) and the extraction it feeds --------------------
    extraction = None
    extraction_hung = False
    if multi and not args.no_extract:
        def halo_and_extract():
            try:
                torch.cuda.set_device(local_rank)      # the current device is per thread
                if args.inject_fault == "raise":
                    raise RuntimeError("injected fault (--inject-fault raise)")
                if args.inject_fault == "hang":
                    threading.Event().wait()
                from semantic_slam_amd.sharded import ShardedVolume
                sv = ShardedVolume(dims, lambda a, b: vol, dist=dist, comm_device=comm_dev)
                assert (sv.z_begin, sv.z_end) == (zb, ze)
                fence()
                t1 = time.perf_counter()
                halo = sv.halo_exchange()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                n_x = len(vol.extract_crossings(halo))
                t3 = time.perf_counter()
                et = torch.tensor([t2 - t1, t3 - t2, float(n_x)], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
                dist.all_reduce(et[:2], op=dist.ReduceOp.MAX)
                dist.all_reduce(et[2:], op=dist.ReduceOp.SUM)
                return {"halo_exchange_ms": round(float(et[0]) * 1e3, 3), "crossings_ms": round(float(et[1]) * 1e3, 3),
                        "vertices": int(et[2]), "halo_bytes_per_boundary": 8 * dims[0] * dims[1],
                        "backend": "rccl (buffers in HBM, device-to-device slice copy, device-resident halo)" if comm_dev == "cuda"
                                   else "gloo (host buffers)",
                        "note": "after the timed region: every rank sends its first slice to the rank below (grouped isend/irecv), "
                                "then extracts its slab's zero crossings with the received slice as +z neighbour; max over ranks"}
            except Exception as e:   # noqa: BLE001 -- the headline has been measured: report the failure, keep the line
                return {"error": repr(e)[:400]}
        # On a thread with a deadline: the headline has been measured, and a wire that never answers (this path has never
        # run over RCCL -- the build box has one GPU) must not take the line with it.
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.update(r=halo_and_extract()), daemon=True)
        th.start()
        th.join(args.extract_deadline)
        if th.is_alive():
            extraction_hung = True
            extraction = {"error": f"halo exchange + extraction did not finish within {args.extract_deadline:g} s; the line is printed without it"}
        else:
            extraction = box.get("r")

    # A dead wire must be visible in the driver's record: the line is still printed (the timed region is done), but the
    # process ends with a non-zero code -- 3 = the halo exchange / extraction never came back (a stuck collective),
    # 4 = it failed with an error -- on every rank.
    extraction_failed = extraction is not None and "error" in extraction
    if rank != 0:
        if extraction_hung:
            os._exit(3)          # a collective is stuck on the other thread: nothing to tear down politely
        vol.close()
        if dist is not None:
            dist.destroy_process_group()
        if extraction_failed:
            sys.exit(4)
        return

    # ---- roofline of the dominant kernel on this rank -------------------------------------------------------------
    frame_bytes = 4.0 * H * Wd + 100.0
    if fpl == 1:
        # one launch per step: SURVEY.md section 8(d) to the letter -- 16 B per voxel updated (TSDF + weight, read and
        # written) + one pass over the depth frame + the parameters
        launches = timed_steps
        units_per_launch = n_upd_per_frame
        alg_bytes = 16.0 * units_per_launch + frame_bytes
        unit_name = "voxel updated by the launch's frame (SURVEY.md section 8d: 4 B TSDF + 4 B weight, read and written)"
        bytes_model = "16 B x voxels updated + 4*H*W (one pass over the depth frame) + 100 (intrinsics, pose)"
    else:
        # up to fpl frames per pass with the voxels held in registers: a launch has to read and write each voxel it
        # touches once, whatever the number of its frames that update it, and to read each of its depth frames once
        full, rem = divmod(K, fpl)
        launches = (full + (1 if rem else 0)) * repeats
        frames_per = K / (full + (1 if rem else 0))
        units_per_launch = touched_per_launch if touched_per_launch is not None else min(float(n_slab), frames_per * n_upd_per_frame)
        alg_bytes = 16.0 * units_per_launch + frames_per * frame_bytes
        if args.labels > 0:
            # label state of a voxel that received evidence in the launch: label u16 + Fp f32 + Bp f32, read and written (repeated
            # evidence within a launch is a read-modify-write per frame in the kernel; L2 serves the repeats), and the frame's
            # label + score images beside its depth
            alg_bytes += 20.0 * labelled_per_launch + frames_per * 6.0 * H * Wd
        unit_name = ("voxel updated by at least one frame of the launch (4 B TSDF + 4 B weight, read once and written once per "
                     "launch)" + ("" if touched_per_launch is None and W.full_coverage else
                                  "; counted per launch on a fresh volume over one pass of the pose sequence, mean per 32 frames" if touched_per_launch is not None else
                                  "; bounded by frames x updates per frame (N > 1: not counted)"))
        bytes_model = "16 B x voxels touched by the launch + frames per launch x (4*H*W + 100)" + (
            "" if args.labels <= 0 else " + 20 B x voxels that received label evidence in the launch (u16 label + f32 Fp + f32 Bp, read and "
            "written) + frames per launch x 6*H*W (label and score images)")
    kernel_ms = kernel_ms_total / launches
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    if fpl == 1:
        kname = ("void tsdfk::integrate_tile<2, false>(tsdfk::IntegrateParams)" if cfg.dim_x % 256 == 0 else
                 "tsdfk::integrate_multi_single<...>") if variant in (0, 3) else f"variant {variant}"
        ksub = "integrate_tile<" if cfg.dim_x % 256 == 0 else "integrate_multi_single<"
    else:
        # what the library launches for a known sequence: over the brick work list while the previous launch's claims pay
        # (decided per launch; the S-band and variant 7 never do), else per voxel
        listed = args.workload in ("ssurf", "traj", "sfull") and variant != 7
        kname = (f"tsdfk::integrate_brick_list<false, {'true' if args.labels > 0 else 'false'}, false> over the work list of tsdfk::classify_brick_list (decided per launch)"
                 if listed else "tsdfk::integrate_multi_inline<1, true, false, false, false> (every voxel projected)")
        ksub = "integrate_brick_list<" if listed else "integrate_multi_inline<"

    traffic, traffic_note = None, "not measured (--no-traffic)"
    main_valu = None
    if not args.no_traffic and world == 1 and (args.emulate_world <= 1 or args.labels > 0):
        # the child launches what the timed region launched: single frames, or passes of the same number of frames
        ca_main = child_args_for(args, variant=variant, steps=6 if fpl == 1 else W.n_pose if W.n_pose >= 2 * fpl else 3 * fpl)
        traffic, traffic_note = measure_traffic(ksub, ca_main, fpl=fpl, wide_reads="brick_list" not in ksub)
        if fpl > 1:      # these launches are bound by instruction issue: the measured VALU share beside the bytes
            vu, vu_note = measure_valu(ksub, ca_main, fpl)
            main_valu = vu if vu is not None else {"error": vu_note}

    # the kernel's name as the profiler printed it in this run's child passes (what profiles/*_kernel_stats.csv lists); without
    # such a pass (N > 1, --no-traffic) the name the source gives it
    kname_source = "static (no profiled pass in this run)"
    if ksub in PMC_KERNEL_NAMES:
        kname, kname_source = PMC_KERNEL_NAMES[ksub], "rocprofv3 Kernel_Name of this run's PMC passes"

    mode_desc = ("one kernel launch per step (tsdf_integrate_device per frame: the reference's TSDF::Integrate call shape)" if fpl == 1
                 else f"tsdf_integrate_frames{'_labels' if args.labels > 0 else ''}_device, up to {fpl} frames per pass over the volume"
                      + (f", per-voxel label fusion from {args.labels} instance masks per frame in the same passes" if args.labels > 0 else ""))
    line = {
        # BASELINE.json's metric, naming the grid that was integrated (N = 1: its 512³; N > 1: configs[3]'s 1024³ by default)
        "metric": f"Mvoxels/sec integrated, {grid_label(dims)} grid @ 640×480 depth; achieved HBM GB/s %peak",
        "value": round((n_global if args.emulate_world <= 1 else n_slab) * timed_steps / wall / 1e6, 1),
        "unit": "Mvoxels/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": round(wall / timed_steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,            # the reference publishes no number for this metric
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload} {dims[0]}x{dims[1]}x{dims[2]} @ {vs * 1000:g} mm, 640x480 depth resident in HBM: "
                               f"{W.desc}; {mode_desc}",
                   "grid": list(dims), "voxel_size_m": vs, "image": [H, Wd], "trunc_margin_m": float(cfg.trunc_margin),
                   "partition": f"{part_world} z-slab(s) of {ze - zb} slices ({n_slab} voxels), one per GPU"
                                + ("" if part_world == 1 else
                                   f"; strong scaling: the {dims[0]}x{dims[1]}x{dims[2]} @ {vs * 1000:g} mm grid"
                                   + (" of BASELINE.json configs[3]" if dims == (1024, 1024, 1024) and abs(vs - 0.002) < 1e-9 else "")
                                   + " is the same at every N > 1"
                                   + (" (at N = 8 a slab holds the 134 M voxels of the N = 1 line's 512^3 grid)" if dims == (1024, 1024, 1024) else "")
                                   if scaling == "strong" else
                                   f"; weak scaling: every rank's slab holds the {D}^3 voxels of the N = 1 grid, the global grid grows with N")
                                + (f"; single-GPU rehearsal of rank {args.emulate_rank} of {part_world} (value = this slab's rate)"
                                   if args.emulate_world > 1 and world == 1 else ""),
                   "mode": args.mode, "kernel_variant": variant, "frames_per_launch": fpl},
        "timing": {"timed_steps": timed_steps, "repeats": repeats, "timed_region_s": round(wall, 4),
                   "kernel_s": round(kernel_ms_total * 1e-3, 4), "warmup_steps_run": n_warm + K,
                   "note": f"the {K}-step sequence is repeated until >= {MIN_TIMED_MS / 1e3:g} s of device time; ms_per_step = "
                           "wall / timed_steps between barrier + synchronize fences, max over ranks"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None if traffic is None else int(traffic),
                     "traffic_note": traffic_note,
                     "traffic_over_algorithmic": None if traffic is None else round(traffic / alg_bytes, 4),
                     "kernel": kname, "kernel_name_source": kname_source, "launches": launches, "kernel_ms": round(kernel_ms, 5),
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "algorithmic_bytes_per_unit": 16, "units_per_launch": int(units_per_launch),
                     "unit_name": unit_name, "bytes_model": bytes_model,
                     "voxel_updates_per_frame": int(n_upd_per_frame),
                     "note": "per-rank slab launch; kernel_ms = HIP-event time of the timed region (events on the handle's "
                             "stream) / launches; achieved = algorithmic_bytes_per_launch / kernel_ms"},
    }
    if main_valu is not None:
        line["roofline"]["valu"] = main_valu
        if traffic is not None:
            line["roofline"]["hbm_measured_GBps"] = round(traffic / (kernel_ms * 1e-3) / 1e9, 1)
            line["roofline"]["hbm_measured_frac"] = round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if args.labels > 0:
        line["labels"] = {"instance_masks_per_frame": args.labels, "voxels_labelled_per_launch": int(labelled_per_launch),
                          "label_state_bytes_per_voxel": 10,
                          "note": "Integrate + per-voxel label fusion in the same passes (tsdf_integrate_frames_labels_device; rule: "
                                  "ref src/ObjectPoint.cpp:190-219 per voxel, DESIGN.md N3); label + score images composed on the device "
                                  "from synthetic instance masks in MaskRCNN's output format"}
    if multi:
        # what a SCALE record must show: how many ranks the communicator saw, through which backend, what the halo costs, and the
        # job's rate against N GPUs' HBM (SURVEY.md section 8d "scaling table": Mvox/s and % of (G x 8 TB/s))
        job_bytes = 16.0 * n_upd_per_frame * world + world * frame_bytes if fpl == 1 else None
        line["multi_gpu"] = {"world": world, "backend": "rccl (torch.distributed 'nccl')" if backend != "gloo" else "gloo",
                             "comm_device": comm_dev, "halo_bytes_per_boundary": 8 * dims[0] * dims[1],
                             "data_path_collectives": 0, "halo_and_extraction_ran": extraction is not None and not extraction_failed,
                             "global_grid": list(dims), "voxel_size_m": vs, "voxels_per_rank": int(n_slab),
                             "hbm_GBps_all_ranks": None if job_bytes is None else round(job_bytes / (wall / timed_steps) / 1e9, 1),
                             "hbm_frac_of_n_gpus": None if job_bytes is None else round(job_bytes / (wall / timed_steps) / 1e9 / (world * HBM_PEAK_GBS), 4),
                             "hbm_note": "algorithmic bytes of all ranks (16 B x voxels updated per frame, this rank's count x N, + a depth frame "
                                         "per rank) / ms_per_step (wall, max over ranks) against N x 8 TB/s"}
        if strong_512 is not None:
            line["strong_512"] = strong_512
        if n1_here is not None:
            line["n1_same_job"] = n1_here
    if extraction is not None:
        line["extraction"] = extraction
        line["extraction_hung"] = bool(extraction_hung)
    extras = world == 1 and args.emulate_world <= 1 and not args.no_extras

    def timed_leg(v, start_warm, n_warm_, n_block, min_ms=150.0):
        """warm-up, then n_block-step blocks until min_ms of device time; returns (ms per step, steps timed)."""
        run_block(v, start_warm, n_warm_)
        tot, steps, pos = 0.0, 0, start_warm + n_warm_
        while tot < min_ms and steps < 100000:
            tot += run_block(v, pos, n_block)
            pos += n_block
            steps += n_block
        return tot / steps, steps

    if extras and args.variant < 0:
        # The same workload through the other mode: what holding the voxels in registers over 32 frames buys / what one
        # launch per frame costs.  Bit-identical results either way (tests/test_gpu_multiframe.py).
        other = 0 if fpl == 1 else 3
        vol.set_kernel_variant(other)
        vol.reset()
        ofpl = vol.frames_per_launch
        ms_o, n_o = timed_leg(vol, 0, max(2 * ofpl, 8), 2 * ofpl if ofpl > 1 else 32)
        rec = {"mode": "fused" if ofpl > 1 else "frame", "frames_per_launch": ofpl, "ms_per_step": round(ms_o, 5),
               "value": round(n_global / ms_o / 1e3, 1), "unit": "Mvoxels/s", "steps_timed": n_o}
        if ofpl > 1 and W.full_coverage:
            b = 16.0 * n_slab + ofpl * frame_bytes
            rec["roofline"] = {
                "bound": "valu_issue", "hbm_bytes_per_launch": int(b), "kernel_ms": round(ms_o * ofpl, 5),
                "hbm_achieved_GBps": round(b / (ms_o * ofpl * 1e-3) / 1e9, 1),
                "hbm_frac": round(b / (ms_o * ofpl * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": f"each voxel is read and written once per {ofpl} frames, so HBM is far from binding; the launch is bound by "
                        "VALU instruction issue (the exact projection, the reference's two divisions per update, eight compares per "
                        "voxel-frame: SQ counters and the per-instruction issue costs in profiles/ and DESIGN.md section 4)"}
        if ofpl > 1 and not args.no_traffic:
            # what the launch really moves and how busy its VALUs are: three short profiled child runs of the same leg
            ca = child_args_for(args, mode="fused", variant=0, steps=4 * ofpl)
            ksub_o = "integrate_brick_list<" if args.workload != "sband" else "integrate_multi_inline<"
            tr, tr_note = measure_traffic(ksub_o, ca, fpl=ofpl, wide_reads="brick_list" not in ksub_o)
            vu, vu_note = measure_valu(ksub_o, ca, ofpl)
            r = rec.setdefault("roofline", {"bound": "valu_issue", "kernel_ms": round(ms_o * ofpl, 5)})
            r["traffic"] = None if tr is None else int(tr)
            r["traffic_note"] = tr_note
            if tr is not None:
                r["hbm_measured_GBps"] = round(tr / (ms_o * ofpl * 1e-3) / 1e9, 1)
                r["hbm_measured_frac"] = round(tr / (ms_o * ofpl * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            r["valu"] = vu if vu is not None else {"error": vu_note}
        line["fused_sequence" if ofpl > 1 else "per_frame_launches"] = rec
        vol.set_kernel_variant(variant)
        vol.reset()
    if extras and args.workload == "sband" and args.variant < 0:
        # S-full (SURVEY.md section 8d as written: every voxel updated with dist = 1).  The kernels prove that the TSDF value
        # cannot change there (free space) and move only the weights, and the fused path's patch classification proves it per
        # workgroup without projecting a voxel: rates of a different byte model, reported for continuity with round 1.
        Wf = Workload("sfull", dims, vs)
        with capi.Volume(capi.make_config(dims, vs, Wf.origin, device=local_rank)) as fv:
            f_dev = torch.from_numpy(Wf.depths[0]).cuda()
            def fblock(v, start, n):
                return v.integrate_sequence_timed(f_dev.data_ptr(), Wf.block(start, n)[0])
            res = {}
            for tag, var, nb in (("per_frame_launches", 3, 32), ("fused_classified", 0, 256), ("fused_per_voxel", 7, 128)):
                fv.set_kernel_variant(var)
                fv.reset()
                fblock(fv, 0, 64)
                tot, steps = 0.0, 0
                while tot < 100.0:
                    tot += fblock(fv, 64 + steps, nb)
                    steps += nb
                res[tag] = {"ms_per_step": round(tot / steps, 5), "value": round(n_global / (tot / steps) / 1e3, 1), "unit": "Mvoxels/s"}
            res["per_frame_launches"]["hbm_GBps_at_8B_per_voxel"] = round(8.0 * n_global / res["per_frame_launches"]["ms_per_step"] / 1e6, 1)
            res["note"] = ("free space: the TSDF value provably stays 1, so one launch per frame moves 8 B per voxel (weights), the fused "
                           "path moves them once per 32 frames, and with the patch classification (variant 0) no voxel is projected at all")
            line["sfull"] = res
            del f_dev
    if extras and args.workload in ("sband", "sfull"):
        # The realistic workload of SURVEY.md section 8(d): sphere + wall, orbit of 64 poses, depth re-rendered per pose
        # (64 frames resident in HBM), through the sequence path.
        def realistic(noise_mm, holes, budget_ms, profile):
            Ws = Workload("ssurf", dims, vs, noise_mm, holes, cache_dir=cache_dir)
            with capi.Volume(capi.make_config(dims, vs, Ws.origin, device=local_rank)) as sv_:
                s_dev = [torch.from_numpy(d).cuda() for d in Ws.depths]
                def sblock(start, n):
                    poses, idx = Ws.block(start, n)
                    return sv_.integrate_frames_timed([s_dev[i].data_ptr() for i in idx], poses)
                sblock(0, 192)          # tables and pre-pass buffers allocated, the per-launch decision settled, clocks up
                tot, steps = 0.0, 0
                while tot < budget_ms:  # 256 steps (8 launches) per call, as the headline loop queues them
                    tot += sblock(192 + steps, 256)
                    steps += 256
                _, w_r = sv_.download()
                upd_r = float(w_r.astype(np.float64).sum()) / (192 + steps)
                del w_r, s_dev
            ms_r = tot / steps
            rec = {"workload": f"ssurf {dims[0]}x{dims[1]}x{dims[2]} @ {vs * 1000:g} mm: {Ws.desc}; fused sequence path",
                   "ms_per_step": round(ms_r, 5), "value": round(n_global / ms_r / 1e3, 1), "unit": "Mvoxels/s",
                   "updated_fraction": round(upd_r / n_global, 4), "frames": steps}
            if profile and not args.no_traffic:
                rec["roofline"] = leg_roofline("ssurf", dims[0], ms_r * 32, noise_mm, holes)
            return rec

        def leg_roofline(workload, grid, launch_ms, noise_mm=0.0, holes=0.0):
            """Measured bytes and VALU share of a classified fused launch (integrate_brick_list): three short profiled child runs."""
            # the children run whole passes over the pose sequence (the trajectory's launches differ widely)
            ca = child_args_for(args, workload=workload, mode="fused", variant=0, grid=grid, steps=194 if workload == "traj" else 128,
                                noise_mm=noise_mm, holes=holes)
            tr, tr_note = measure_traffic("integrate_brick_list<", ca, fpl=32, wide_reads=False)
            vu, vu_note = measure_valu("integrate_brick_list<", ca, 32)
            r = {"bound": "valu_issue", "launch_ms": round(launch_ms, 5), "frames_per_launch": 32,
                 "traffic": None if tr is None else int(tr), "traffic_note": tr_note,
                 "valu": vu if vu is not None else {"error": vu_note},
                 "note": "launch_ms = HIP-event time per 32-frame launch (tile tables, brick work list and the Integrate kernel); traffic and "
                         "the VALU share are the Integrate kernel's (integrate_brick_list, > 90 % of the launch)"}
            if tr is not None:
                r["hbm_measured_GBps"] = round(tr / (launch_ms * 1e-3) / 1e9, 1)
                r["hbm_measured_frac"] = round(tr / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            return r

        line["realistic_workload"] = realistic(0.0, 0.0, 150.0, True)
        # the same scene through a sensor's imperfections: Gaussian noise (sigma 2 mm) and 5 % of every frame
        # dropped in 8 x 8 pixel blocks -- free-space bricks behind a dropout cannot be claimed as a whole any more
        line["realistic_workload_noisy"] = realistic(2.0, 0.05, 60.0, False)
        # BASELINE configs[2]: 1024^3 @ 2 mm on the 194 keyframe poses of the reference's saved fr3_office run, fused; 8.6 GB of
        # volume + 238 MB of frames beside the headline's 1 GB -- only where the card has the room
        free_b, _ = torch.cuda.mem_get_info()
        if args.grid == 512 and free_b > 14 * 2 ** 30:
            try:
                d2, v2 = (1024, 1024, 1024), 0.002
                Wt = Workload("traj", d2, v2, 0.0, 0.0, args.tum_dir, cache_dir)
                with capi.Volume(capi.make_config(d2, v2, Wt.origin, base2world=Wt.base2world, device=local_rank)) as tv:
                    t_dev = [torch.from_numpy(d).cuda() for d in Wt.depths]
                    def tblock(start, n):
                        poses, idx = Wt.block(start, n)
                        return tv.integrate_frames_timed([t_dev[i].data_ptr() for i in idx], poses)
                    tblock(0, Wt.n_pose)                      # one trip along the trajectory: buffers, decisions, clocks
                    tot, steps = 0.0, 0
                    while tot < 150.0:
                        tot += tblock(steps, Wt.n_pose)
                        steps += Wt.n_pose
                    del t_dev
                ms_t = tot / steps
                n2 = d2[0] * d2[1] * d2[2]
                rec = {"workload": f"traj 1024x1024x1024 @ 2 mm: {Wt.desc}; fused sequence path (BASELINE.json configs[2])",
                       "ms_per_step": round(ms_t, 5), "value": round(n2 / ms_t / 1e3, 1), "unit": "Mvoxels/s", "frames": steps}
                if not args.no_traffic:
                    rec["roofline"] = leg_roofline("traj", 1024, ms_t * 32)
                line["configs2_traj1024"] = rec
            except Exception as e:   # noqa: BLE001 -- a companion leg must not take the line with it
                line["configs2_traj1024"] = {"error": repr(e)[:300]}
    if extras and len(W.depths) == 1:
        # PCIe-inclusive rate of the reference-style call (tsdf_integrate: host depth pointer -> pinned staging -> H2D on a
        # copy stream).  By default the library collects such frames and applies them 32 at a time as one fused sequence
        # (deferred integration); "immediate" is the same with one kernel per call.  Beside the headline, never as it.
        rec = {}
        n_host = 256
        for tag, defer in (("deferred", 32), ("immediate", 0)):
            vol.set_kernel_variant(0)     # the library's default policy (the headline's variant 3 forbids fused launches)
            vol.set_deferral(defer)
            for k in range(32):
                vol.integrate(W.depths[0], W.poses[k % W.n_pose])
            vol.sync()
            t1 = time.perf_counter()
            for k in range(n_host):
                vol.integrate(W.depths[0], W.poses[k % W.n_pose])
            vol.sync()
            dt = time.perf_counter() - t1
            rec[tag] = {"ms_per_step": round(dt / n_host * 1e3, 5), "value": round(n_global * n_host / dt / 1e6, 1), "unit": "Mvoxels/s"}
        vol.set_deferral(32)
        vol.set_kernel_variant(variant)
        rec["note"] = ("tsdf_integrate with a host depth pointer, 1.2 MB H2D per frame, Python ctypes call overhead included; "
                       "deferred (the default): frames collected in HBM, one fused launch per 32 calls, flushed by any call that "
                       "observes the volume; immediate: one kernel per call")
        line["host_depth_path"] = rec
    if not args.no_cpu_baseline and world == 1 and args.emulate_world <= 1:
        base, ref = cpu_baseline(args, W, cfg.cam_K, float(cfg.trunc_margin))
        line["cpu_baseline"] = base
        if ref is not None:
            line["cpu_reference"] = ref
    print(json.dumps(line))
    sys.stdout.flush()
    if own_cache:
        shutil.rmtree(own_cache, ignore_errors=True)
    if extraction_hung:
        os._exit(3)              # see above: the line is out, a collective is stuck on the other thread -> rc 3
    vol.close()
    if dist is not None:
        dist.destroy_process_group()
    if extraction_failed:
        sys.exit(4)              # the line is out; the halo exchange / extraction raised -> rc 4


if __name__ == "__main__":
    main()
