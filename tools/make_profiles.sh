#!/bin/bash
# Everything under profiles/<round>_* comes from ONE call of this script on the GPU box:
#   bash tools/make_profiles.sh r03        (then copy gpurun_out/<round>/* into profiles/)
set -o pipefail
R=${1:-r03}
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
stats sband512 && stats ssurf512 --workload ssurf && stats traj1024 --workload traj && stats ssurf200 --workload ssurf --grid 200 || exit 1
cd $GRAFT_REPO_ROOT
# the driver's command, as the driver runs it (in-run PMC traffic, companion legs, CPU baseline)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${R}_bench_default.json 2> $out/_bench.err || exit 1
python3 bench.py --workload ssurf > $out/${R}_bench_ssurf512.json 2>> $out/_bench.err || exit 1
python3 bench.py --workload traj > $out/${R}_bench_traj1024.json 2>> $out/_bench.err || exit 1
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
python3 tools/host_path_time.py > $out/${R}_host_path_time.txt 2>&1
python3 tools/claim_rate.py --workload ssurf --shapes 2,4,8 > $out/${R}_claim_rate.txt 2>&1
python3 tools/claim_rate.py --workload traj --grid 1024 --shapes 2,4,8 >> $out/${R}_claim_rate.txt 2>&1
rm -f $out/_bench.err
ls -la $out
