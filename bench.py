#!/usr/bin/env python3
"""bench.py -- Mvoxels/s of the HIP TSDF Integrate path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 512] [--workload sfull|ssurf]

One "step" = one Integrate of one 640x480 depth frame into the whole grid.  At N = 1 the
workload is BASELINE.json configs[1]: 512^3 @ 5 mm, synthetic depth + pose stream (S-full,
SURVEY.md section 8d: every voxel updated every frame).  For N > 1 every rank owns one z-slab and
there is no data-path collective (voxels are independent); by default the scaling is WEAK: each rank's
slab holds as many voxels as the whole N = 1 grid and the global grid grows inside the same physical
box (1024^3 @ 2.5 mm at N = 8); --scaling strong cuts the 512^3 grid itself.  Launch with
torch.distributed.run as the driver does.  The depth frame is resident in HBM before the timed
region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=640)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--grid", type=int, default=512, help="grid edge in voxels (512 or 1024)")
    ap.add_argument("--workload", default="sfull", choices=["sfull", "ssurf"])
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (see DESIGN.md)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank integrates a slab as large as the whole N = 1 grid (the global grid "
                         "grows with N: 512^3 -> 512x512x1024 -> 512x1024x1024 -> 1024^3 at N = 1, 2, 4, 8, same physical "
                         "extent, finer voxels); strong = the N = 1 grid cut into N slabs")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single-GPU rehearsal of one rank of an N-GPU job: integrate only rank 0's z-slab of N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the contract's timed region (for profiler runs: no streaming-variant, host-depth or "
                         "S-surf companion legs, which launch more kernels after it)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(args, dims, vs, origin, cfg, depth, poses, n_upd_expected_per_slice):
    """Time the CPU oracle (and the reference's own kernel body when oracle/_ref was built) on a
    bounded z-slab sample of the same workload, all host threads.  Checker code: it is
    measured here as the baseline, never used to produce the GPU result."""
    from oracle.oracle import Oracle, Ref
    orc = Oracle()
    D = dims[0]
    nz = min(D, 64)                      # sample: the middle nz slices of the grid
    zb = (D - nz) // 2
    threads = orc.max_threads()
    t, w = orc.init_grid(dims, zb, zb + nz)
    orc.integrate(cfg.cam_K, poses[0], depth, dims, origin, vs, cfg.trunc_margin, t, w, z_begin=zb,
                  z_end=zb + nz, threads=threads)  # warm-up: page in the slab
    frames, t0, n_upd = 0, time.perf_counter(), 0
    while True:
        n_upd += orc.integrate(cfg.cam_K, poses[(frames + 1) % len(poses)], depth, dims, origin, vs,
                               cfg.trunc_margin, t, w, z_begin=zb, z_end=zb + nz, threads=threads)
        frames += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or frames >= 200:
            break
    out = {"value": round(nz * D * D * frames / el / 1e6, 1), "unit": "Mvoxels/s", "cores": threads,
           "kind": "port",
           "sample": f"oracle/tsdf_oracle.c (OpenMP over rows), {frames} frames of the same workload into "
                     f"z-slab [{zb},{zb + nz}) of the {D}^3 grid, {el:.1f} s"}
    if n_upd_expected_per_slice is not None:
        assert n_upd == frames * nz * n_upd_expected_per_slice, "S-full must update every voxel (N_upd == N)"
    ref = None
    if Ref.available() and D <= 1024:
        # the reference body has no slab form: give it a grid that IS the slab (origin shifted in z
        # on the host; timing only, values are not compared here)
        r = Ref()
        rd = (D, D, nz)
        ro = np.array([origin[0], origin[1], origin[2] + zb * vs], np.float32)
        t2, w2 = orc.init_grid(rd)
        r.integrate(cfg.cam_K, poses[0], depth, rd, ro, vs, cfg.trunc_margin, t2, w2, threads=threads)
        f2, t0 = 0, time.perf_counter()
        while True:
            r.integrate(cfg.cam_K, poses[(f2 + 1) % len(poses)], depth, rd, ro, vs, cfg.trunc_margin, t2, w2,
                        threads=threads)
            f2 += 1
            el2 = time.perf_counter() - t0
            if el2 > args.cpu_seconds / 2 or f2 >= 200:
                break
        ref = {"value": round(nz * D * D * f2 / el2 / 1e6, 1), "unit": "Mvoxels/s", "cores": threads,
               "kind": "reference",
               "sample": f"GpuIntegrate body of the reference (src/tsdf.cu:15-60) built for the host by "
                         f"oracle/Makefile, {f2} frames into a {D}x{D}x{nz} grid, {el2:.1f} s"}
    return out, ref


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
    # devices, the two scalars of the timing reduction travel through gloo); the driver's runs use RCCL.
    backend = os.environ.get("TSDF_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    D = args.grid
    vs = {512: 0.005, 1024: 0.002}.get(D, 2.56 / D)
    dims = [D, D, D]
    part_world = args.emulate_world if (args.emulate_world > 1 and world == 1) else world
    if args.scaling == "weak" and part_world > 1:
        # per-rank work fixed: grow the grid with N inside the same physical box (z, then y, then x doubled in
        # turn, voxels halved whenever z is doubled); a rank count that is not a power of two just gets N x the slices
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
    dims = tuple(dims)
    if args.workload == "sfull":
        origin = synth.sfull_volume(dims, vs)
        depth = synth.sfull_depth()
        n_pose = 64
        poses = np.stack([synth.sfull_pose(k) for k in range(n_pose)])
    else:
        origin = synth.surf_volume(max(dims), vs, 1.0)
        scene = synth.SurfScene(dims, vs, origin)
        n_pose = 64
        poses = np.stack([scene.pose(k, n_pose) for k in range(n_pose)])
        depth = scene.depth(poses[0], quantize=True)  # one resident frame, orbiting camera

    # z-slab of this rank (ref layout is z-major, so a slab is one contiguous range)
    Dz = dims[2]
    zb, ze = rank * Dz // world, (rank + 1) * Dz // world
    if args.emulate_world > 1 and world == 1:
        zb, ze = 0, Dz // args.emulate_world
    n_global = dims[0] * dims[1] * dims[2]
    cfg = capi.make_config(dims, vs, origin, z_begin=zb, z_end=ze, device=local_rank)
    vol = capi.Volume(cfg)
    vol.set_kernel_variant(args.variant)
    d_dev = torch.from_numpy(depth).cuda()

    def pose_block(start, n):
        return np.stack([poses[(start + i) % n_pose] for i in range(n)])

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        vol.integrate_sequence_timed(d_dev.data_ptr(), pose_block(0, args.warmup))
    fence()
    t0 = time.perf_counter()
    # exactly K steps, queued back to back on the handle's stream, HIP events around them
    kernel_ms_total = vol.integrate_sequence_timed(d_dev.data_ptr(), pose_block(args.warmup, args.steps))
    fence()
    wall = time.perf_counter() - t0

    tt = torch.tensor([wall, kernel_ms_total], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    wall, kernel_ms_total = float(tt[0]), float(tt[1])

    # --- what was integrated: count updates on this rank's slab ------------------------------
    t_host, w_host = vol.download()
    n_frames = args.warmup + args.steps
    n_slab = vol.n_voxels
    upd_total = float(w_host.astype(np.float64).sum())  # each update adds exactly 1 to one weight
    if args.workload == "sfull":
        assert w_host.min() == w_host.max() == float(n_frames), "S-full: every voxel every frame"
        assert np.all(t_host == 1.0)
    n_upd_per_frame = upd_total / n_frames
    del t_host, w_host

    # --- roofline of the dominant kernel on this rank ---------------------------------------------
    # The default path applies FPL (tsdf_frames_per_launch: 32) consecutive frames per pass over the volume:
    # one launch = FPL steps.  Bytes one launch must move (DESIGN.md "Bytes model"): 4 B weight read +
    # 4 B weight write per voxel touched by any of its frames; the TSDF value is read only where the
    # free-space summary does not already say "this 256-voxel segment is all ones" and written only
    # where it changes; the summary words; FPL passes over the depth frame; the parameters.
    # SURVEY.md section 8(d) priced every updated voxel of every frame at 16 B; that figure is reported
    # beside it, as is the plain streaming variant in which those 16 B really move.
    H, W = depth.shape
    v = args.variant
    fused = v in (0, 4, 5, 7, 8)
    fpl = vol.frames_per_launch if fused else 1
    full, rem = divmod(args.steps, fpl)          # K steps = `full` launches of fpl frames + one of `rem`
    launches = full + (1 if rem else 0)
    has_summary = v in (0, 3, 4, 5, 7, 8) or 32 <= v < 64 or 80 <= v < 96 or v >= 112
    has_elide = has_summary or v in (18, 19, 22, 23, 26, 27) or v >= 64
    flag_bytes = 4.0 * n_slab / 256.0 if has_summary else 0.0

    def launch_bytes(n):
        """(algorithmic bytes, SURVEY's 16-B figure, voxels touched, TSDF values read, written) of one launch of n frames."""
        touched = min(float(n_slab), n * n_upd_per_frame)   # exact for sfull; upper bound otherwise
        if args.workload == "sfull":
            # every voxel updated every frame, every TSDF value stays exactly 1 (asserted above)
            t_read = 0.0 if has_summary else touched
            t_written = 0.0 if has_elide else touched
        else:
            t_read = t_written = touched
        b = 8.0 * touched + 4.0 * t_read + 4.0 * t_written + flag_bytes + n * (4.0 * H * W + 100.0)
        return b, n * (16.0 * n_upd_per_frame + 4.0 * H * W + 100.0), touched, t_read, t_written

    count_note = "exact" if args.workload == "sfull" else \
        "upper bound (TSDF reads/writes elided on the device are not counted there)"
    bytes_per_launch, bytes_survey, n_touched, n_t_read, n_t_written = launch_bytes(fpl if full else rem)
    total_bytes = full * launch_bytes(fpl)[0] + (launch_bytes(rem)[0] if rem else 0.0)
    total_survey = full * launch_bytes(fpl)[1] + (launch_bytes(rem)[1] if rem else 0.0)
    kernel_ms = kernel_ms_total / launches
    achieved = total_bytes / (kernel_ms_total * 1e-3) / 1e9
    achieved_survey = total_survey / (kernel_ms_total * 1e-3) / 1e9

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.isfile(tpath):
        try:
            rec = json.load(open(tpath))
            key = f"{args.workload}_{D}_slab{ze - zb}" + ("" if args.variant == 0 else f"_v{args.variant}")
            if dims != (D, D, D):
                key += "_" + "x".join(str(d) for d in dims)
            if key in rec:
                traffic = rec[key]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None

    line = {
        "metric": f"Mvoxels/sec integrated, {D}\u00b3 grid @ 640\u00d7480 depth; achieved HBM GB/s %peak",   # BASELINE.json
        "value": round((n_global if args.emulate_world <= 1 else n_slab) * args.steps / wall / 1e6, 1),
        "unit": "Mvoxels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload} {dims[0]}x{dims[1]}x{dims[2]} @ {vs * 1000:g} mm, 640x480 depth resident in HBM, "
                               f"{'every voxel updated every frame' if args.workload == 'sfull' else 'sphere+wall orbit'}",
                   "grid": list(dims), "voxel_size_m": vs, "image": [H, W],
                   "partition": f"{world} z-slab(s) of {ze - zb} slices ({n_slab} voxels), one per GPU"
                                + (f"; {args.scaling} scaling from the {D}^3 grid of N = 1" if world > 1 else ""),
                   "kernel_variant": args.variant},
        # roofline.achieved follows the contract to the letter: SURVEY.md section 8(d)'s per-unit figure (16 B per
        # voxel updated: TSDF + weight, read + write, plus one pass over the depth frame) x the units one launch
        # processes / the launch duration.  The kernel provably moves far fewer bytes (bit-identical results), so
        # the figure exceeds the HBM peak; "physical" prices the same launches by the bytes that really move.
        "roofline": {"bound": "hbm", "achieved": round(achieved_survey, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved_survey / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel": ("tsdfk::integrate_multi_inline<1,true,false,false,false,C>, C = patch classification, decided "
                                "per launch from the previous launch's claims (DESIGN.md section 4)") if fused else
                               ("tsdfk::integrate_tile<2,true,true,false,true,false,true>" if v == 3 else f"variant {v}"),
                     "frames_per_launch": fpl, "launches": launches,
                     "kernel_ms": round(kernel_ms, 5),
                     "algorithmic_bytes_per_launch": int(bytes_survey),
                     "algorithmic_bytes_per_unit": 16,
                     "units_per_launch": int(n_upd_per_frame * (fpl if full else rem)),
                     "unit_name": "voxel updated by one frame (SURVEY.md section 8d: 4 B TSDF + 4 B weight, read and written)",
                     "frac_above_one": "the launch applies frames_per_launch frames to voxels held in registers, the free-space "
                                       "summary elides TSDF traffic whose result is provably unchanged, and workgroups whose "
                                       "whole patch a depth tile table proves free space (this input: all of them) or "
                                       "untouched are updated without projecting a voxel -- bit-identical results, so the "
                                       "16 B per update of the model do not move (traffic = PMC bytes per launch); "
                                       "per_voxel_kernel is the same input with that classification off",
                     "physical": {"bytes_per_launch": int(bytes_per_launch), "achieved": round(achieved, 1),
                                  "frac": round(achieved / HBM_PEAK_GBS, 4), "unit": "GB/s",
                                  "voxels_touched_per_launch": int(n_touched),
                                  "tsdf_values_read_per_launch": int(n_t_read),
                                  "tsdf_values_written_per_launch": int(n_t_written),
                                  "tsdf_counts": count_note,
                                  "bytes_model": "per launch: 8 B per voxel touched (weight r+w) + 4 B per TSDF value read + "
                                                 "4 B per TSDF value written + 4 B per 256-voxel summary word + "
                                                 "frames_per_launch * (4*H*W + 100)"},
                     "voxel_updates_per_frame": int(n_upd_per_frame),
                     "binding_resource": "instruction issue (VALU + scalar unit), profiles/r01_sfull512_sq_counters.json; "
                                         "see streaming_variant for the access pattern's HBM rate when all 16 B move",
                     "note": "per-rank slab launch; kernel_ms = HIP-event time of the timed region / launches"},
    }
    if world == 1 and args.variant == 0 and args.emulate_world <= 1 and not args.no_extras:
        # The same workload through the plain streaming variant (no elision, no summary: all 16 B per
        # updated voxel really move).  This is the kernel to read as "how close to the HBM roofline
        # does the access pattern get"; the default kernel above is faster because it moves fewer bytes.
        vol.set_kernel_variant(17)
        vol.reset()
        vol.integrate_sequence_timed(d_dev.data_ptr(), pose_block(0, 10))
        n_s = min(args.steps, 200)
        ms_s = vol.integrate_sequence_timed(d_dev.data_ptr(), pose_block(10, n_s)) / n_s
        _, w_s = vol.download()
        upd_s = float(w_s.astype(np.float64).sum()) / (10 + n_s)
        del w_s
        b_s = 16.0 * upd_s + 4.0 * H * W + 100.0
        line["roofline"]["streaming_variant"] = {
            "kernel": "tsdfk::integrate_tile<R=1,NT> (variant 17): 16 B per updated voxel, nothing elided",
            "kernel_ms": round(ms_s, 5), "bytes_per_launch": int(b_s),
            "achieved": round(b_s / (ms_s * 1e-3) / 1e9, 1), "frac": round(b_s / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "value": round(n_global / ms_s / 1e3, 1), "unit": "GB/s (value: Mvoxels/s)"}
        vol.set_kernel_variant(0)
        vol.reset()
    if world == 1 and args.emulate_world <= 1 and not args.no_extras:
        # PCIe-inclusive rate of the reference-style call (tsdf_integrate: host depth pointer, staged
        # through the pinned ring, 1.2 MB H2D per frame).  Reported beside the headline, never as it.
        n_host = min(args.steps, 200)
        vol.sync()
        t1 = time.perf_counter()
        for k in range(n_host):
            vol.integrate(depth, poses[k % n_pose])
        vol.sync()
        dt = time.perf_counter() - t1
        line["host_depth_path"] = {"ms_per_step": round(dt / n_host * 1e3, 5),
                                   "value": round(n_global * n_host / dt / 1e6, 1), "unit": "Mvoxels/s",
                                   "note": "tsdf_integrate with a host depth pointer: memcpy to pinned staging + "
                                           "H2D copy + kernel per frame, Python ctypes call overhead included"}
    if world == 1 and args.emulate_world <= 1 and args.variant == 0 and not args.no_extras:
        # The same workload with the patch classification switched off (variant 7): every voxel of every frame is
        # projected and tested.  On S-full the default's advantage is a property of the input (the whole volume is
        # free space in front of a constant depth); this is the rate of the per-voxel kernel itself.
        vol.set_kernel_variant(7)
        vol.reset()
        n_w, n_t = vol.frames_per_launch, 10 * vol.frames_per_launch
        vol.integrate_sequence_timed(d_dev.data_ptr(), pose_block(0, n_w))
        ms_p = vol.integrate_sequence_timed(d_dev.data_ptr(), pose_block(n_w, n_t)) / n_t
        line["per_voxel_kernel"] = {"kernel_variant": 7, "ms_per_step": round(ms_p, 5), "value": round(n_global / ms_p / 1e3, 1),
                                    "unit": "Mvoxels/s", "frames": n_t,
                                    "note": "patch classification off: every voxel of every frame projected, sampled and tested"}
        vol.set_kernel_variant(0)
        vol.reset()
    if world == 1 and args.emulate_world <= 1 and args.variant == 0 and args.workload == "sfull" and not args.no_extras:
        # The same grid and kernel on the realistic workload of SURVEY.md section 8(d) (S-surf: a sphere in front of
        # a wall seen from an orbit, uint16-quantised depth): only part of the volume is updated per frame and the
        # TSDF values near the surfaces really change, so nothing about it is "all ones".
        s_origin = synth.surf_volume(max(dims), vs, 1.0)
        scene = synth.SurfScene(dims, vs, s_origin)
        s_poses = np.stack([scene.pose(k, 64) for k in range(64)])
        s_depth = torch.from_numpy(scene.depth(s_poses[0], quantize=True)).cuda()
        with capi.Volume(capi.make_config(dims, vs, s_origin, device=local_rank)) as sv:
            n_w, n_t = 2 * sv.frames_per_launch, 10 * sv.frames_per_launch
            sv.integrate_sequence_timed(s_depth.data_ptr(), np.stack([s_poses[i % 64] for i in range(n_w)]))
            ms_r = sv.integrate_sequence_timed(s_depth.data_ptr(), np.stack([s_poses[(n_w + i) % 64] for i in range(n_t)])) / n_t
            _, w_r = sv.download()
        upd_r = float(w_r.astype(np.float64).sum()) / (n_w + n_t)
        del w_r
        line["realistic_workload"] = {
            "workload": f"ssurf {dims[0]}x{dims[1]}x{dims[2]} @ {vs * 1000:g} mm (sphere + wall, orbit of 64 poses, depth quantised at 1/5000 m)",
            "ms_per_step": round(ms_r, 5), "value": round(n_global / ms_r / 1e3, 1), "unit": "Mvoxels/s",
            "updated_fraction": round(upd_r / n_global, 4), "frames": n_t,
            "algorithmic_GBps": round((16.0 * upd_r + 4.0 * H * W + 100.0) / (ms_r * 1e-3) / 1e9, 1)}
    if not args.no_cpu_baseline and world == 1 and args.emulate_world <= 1:
        per_slice = D * D if args.workload == "sfull" else None
        base, ref = cpu_baseline(args, dims, vs, origin, cfg, depth, poses, per_slice)
        line["cpu_baseline"] = base
        if ref is not None:
            line["cpu_reference"] = ref
    print(json.dumps(line))
    sys.stdout.flush()
    vol.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
