#!/bin/bash
# Everything under profiles/<round>_* comes from ONE call of this script on the GPU box:
#   bash tools/make_profiles.sh r04        (then copy gpurun_out/<round>/* into profiles/)
set -o pipefail
R=${1:-r04}
out=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
stats() {   # stats <tag> <bench args...>: rocprofv3 --kernel-trace --stats of the bench command (no companion legs) + its line
  tag=$1; shift
  rm -rf $out/_kt_$tag
  rocprofv3 --kernel-trace --stats -d $out/_kt_$tag -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-extras --no-traffic --no-cpu-baseline > $out/${R}_bench_${tag}_profiled_run.json 2> $out/_kt_$tag.err || return 1
  cp $(find $out/_kt_$tag -name "*kernel_stats.csv" | head -1) $out/${R}_${tag}_kernel_stats.csv
  rm -rf $out/_kt_$tag $out/_kt_$tag.err
}
CFG4="--workload ssurf --grid 2048 --voxel-mm 2 --emulate-world 8 --labels 6"     # BASELINE.json configs[4] as one rank sees it
stats sband512 && stats ssurf512 --workload ssurf && stats traj1024 --workload traj && stats ssurf200 --workload ssurf --grid 200 \
  && stats cfg4_slab2048_labels_rank1 $CFG4 --emulate-rank 1 || exit 1
cd $GRAFT_REPO_ROOT
# the driver's command, as the driver runs it (in-run PMC traffic, companion legs, CPU baseline)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${R}_bench_default.json 2> $out/_bench.err || exit 1
python3 bench.py --workload ssurf > $out/${R}_bench_ssurf512.json 2>> $out/_bench.err || exit 1
python3 bench.py --workload traj > $out/${R}_bench_traj1024.json 2>> $out/_bench.err || exit 1
# BASELINE.json configs[4]: one rank's 2048 x 2048 x 256 slab of the 2048^3 @ 2 mm grid with per-voxel label fusion (ranks 1 and 2:
# the slab with the sphere's visible cap, and the one behind it), measured bytes and issue-slot share in the line
for r in 1 2; do
  python3 bench.py $CFG4 --emulate-rank $r --no-cpu-baseline > $out/${R}_bench_cfg4_slab2048_labels_rank$r.json 2>> $out/_bench.err || exit 1
done
# the N > 1 line on the one GPU there is: one rank's slab of configs[3] (1024^3 @ 2 mm cut eight ways), and the whole N-rank code
# path over RCCL with one rank (process group, fences, strong_512 / n1_same_job legs, halo step, extraction)
python3 bench.py --emulate-world 8 --emulate-rank 3 --steps 20 --warmup 5 --no-cpu-baseline > $out/${R}_bench_configs3_rank3_of_8.json 2>> $out/_bench.err || exit 1
# north_star's 512^3 strong scaling slab by slab, and the driver's N = 2 command at full size over gloo (both ranks on this GPU)
( echo "# bench.py --emulate-world N --emulate-rank r --grid 512: ms per step, Mvox/s of the slab, roofline.frac"
  for wr in "8 0" "8 5" "4 0" "2 1"; do set -- $wr
    echo "N = $1, rank $2: $(python3 bench.py --emulate-world $1 --emulate-rank $2 --grid 512 --steps 20 --warmup 5 --no-extras --no-traffic --no-cpu-baseline 2>> $out/_bench.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])")"
  done ) > $out/${R}_strong512_rank_slabs.txt
TSDF_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 2>> $out/_bench.err | grep "^{" > $out/${R}_bench_2rank_gloo_rehearsal.json
python3 bench.py --dist-world1 --steps 20 --warmup 5 --no-extras --strong-leg --no-traffic --no-cpu-baseline 2>> $out/_bench.err | grep "^{" > $out/${R}_bench_dist_world1_rccl.json || exit 1     # (RCCL prints a version banner on stdout first)
# SQ counters of the Integrate kernel of the classified fused launches and of the per-voxel fused kernel in the band
for t in "ssurf512 --workload ssurf" "traj1024 --workload traj" "sband512_fused --workload sband --mode fused"; do
  set -- $t; tag=$1; shift
  bash tools/sq_counters.sh ${R}_$tag "$@" > /dev/null || exit 1
  cp gpurun_out/sq_${R}_$tag.json $out/${R}_${tag}_sq_counters.json
done
# calibration of FETCH_SIZE / WRITE_SIZE on launches whose bytes are known (all free space: weights in, weights out)
( bash tools/pmc_bytes.sh ${R}_cal_bricks integrate_brick_list --workload sfull --mode fused --variant 8
  bash tools/pmc_bytes.sh ${R}_cal_rows integrate_multi_inline --workload sfull --mode fused --variant 7 ) > $out/${R}_pmc_calibration_sfull512.txt 2>&1
python3 tools/batch_time.py --n 16 > $out/${R}_batch_time_16x200.txt 2>&1
( python3 tools/host_path_time.py
  g++ -O2 -o /tmp/host_copy tools/microbench/host_copy.cpp && /tmp/host_copy ) > $out/${R}_host_path_time.txt 2>&1
# where the brick work list's extra traffic comes from: the sub-lists dealt to the XCDs by image-row wedges (measurement build)
if [ -f semantic_slam_amd/libtsdf_hip_exp.so ]; then
  for m in 0 1 2 3 0; do
    TSDF_HIP_LIB=$PWD/semantic_slam_amd/libtsdf_hip_exp.so TSDF_WEDGE_MODE=$m python3 bench.py --workload ssurf --no-extras --no-cpu-baseline 2>> $out/_bench.err | python3 -c "
import sys, json; d = json.loads(sys.stdin.read().strip()); r = d['roofline']; v = r.get('valu') or {}
print('S-surf 512^3, list_bucket mode $m: %.5f ms per frame, measured traffic %.1f MB per launch (%.2f x algorithmic), %.2f wavefronts per SIMD, issue-slot share %.3f' % (d['ms_per_step'], (r['traffic'] or 0) / 1e6, r.get('traffic_over_algorithmic') or 0, v.get('mean_waves_per_simd') or 0, v.get('valu_issue_frac') or 0))"
  done > $out/${R}_wedge_modes.txt 2>&1
  # fine (4-pixel) tiles beside the 8-pixel tables, off / on, same box (the measurement build reads TSDF_FINE_TILES)
  ( for m in 0 1 0 1; do for w in "--workload ssurf" "--workload traj" "--workload ssurf --noise-mm 2 --holes 0.05"; do
      echo "fine tiles $m | $w: $(TSDF_HIP_LIB=$PWD/semantic_slam_amd/libtsdf_hip_exp.so TSDF_FINE_TILES=$m python3 bench.py $w --no-extras --no-traffic --no-cpu-baseline 2>> $out/_bench.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip()); print(d['ms_per_step'], 'ms per frame,', d['value'], 'Mvox/s')")"
    done; done
    for m in 0 1; do echo "fine tiles $m:"; TSDF_HIP_LIB=$PWD/semantic_slam_amd/libtsdf_hip_exp.so TSDF_FINE_TILES=$m python3 tools/claim_rate.py --workload ssurf --shapes 2,4,8 2>/dev/null | tail -2; done ) > $out/${R}_fine_tiles.txt 2>&1
  # sequence calls with the pre-pass on the handle's stream / beside the previous launch (TSDF_PIPELINE forces it off / on for every slab size)
  ( for rep in 1 2; do for w in "--workload ssurf --grid 200" "--workload ssurf --grid 320" "--workload ssurf" "--workload traj"; do for m in 0 1; do
      echo "pipelined $m | $w: $(TSDF_HIP_LIB=$PWD/semantic_slam_amd/libtsdf_hip_exp.so TSDF_PIPELINE=$m python3 bench.py $w --no-extras --no-traffic --no-cpu-baseline 2>> $out/_bench.err | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip()); print(d['ms_per_step'], 'ms per frame,', d['value'], 'Mvox/s')")"
    done; done; done ) > $out/${R}_pipeline.txt 2>&1
  # the caller's frame into the pinned ring: memcpy against streaming stores, same box, the reference's call shape
  ( echo "memcpy:"; TSDF_HIP_LIB=$PWD/semantic_slam_amd/libtsdf_hip_exp.so TSDF_PLAIN_MEMCPY=1 python3 tools/host_path_time.py 2>/dev/null | head -1
    echo "streaming stores (what ships):"; TSDF_HIP_LIB=$PWD/semantic_slam_amd/libtsdf_hip_exp.so python3 tools/host_path_time.py 2>/dev/null | head -1
    echo "memcpy:"; TSDF_HIP_LIB=$PWD/semantic_slam_amd/libtsdf_hip_exp.so TSDF_PLAIN_MEMCPY=1 python3 tools/host_path_time.py 2>/dev/null | head -1
    echo "streaming stores (what ships):"; TSDF_HIP_LIB=$PWD/semantic_slam_amd/libtsdf_hip_exp.so python3 tools/host_path_time.py 2>/dev/null | head -1 ) >> $out/${R}_host_path_time.txt 2>&1
fi
python3 tools/claim_rate.py --workload ssurf --shapes 2,4,8 > $out/${R}_claim_rate.txt 2>&1
python3 tools/claim_rate.py --workload traj --grid 1024 --shapes 2,4,8 >> $out/${R}_claim_rate.txt 2>&1
rm -f $out/_bench.err
ls -la $out
