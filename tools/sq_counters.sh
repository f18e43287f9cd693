#!/bin/bash
# SQ counters of the fused kernel on a workload: two rocprofv3 --pmc passes (8 SQ slots each), then a summary.
#   bash tools/sq_counters.sh <tag> <bench args...>     e.g.  bash tools/sq_counters.sh sband_fused --workload sband --mode fused --variant 7
set -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $out/sq_${tag}_a $out/sq_${tag}_b
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $out/sq_${tag}_a -o a --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 128 --warmup 32 --no-extras --no-traffic --no-cpu-baseline > $out/sq_${tag}_a.json 2> $out/sq_${tag}_a.err || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH GRBM_GUI_ACTIVE -d $out/sq_${tag}_b -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 128 --warmup 32 --no-extras --no-traffic --no-cpu-baseline > $out/sq_${tag}_b.json 2> $out/sq_${tag}_b.err || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/sq_summary.py $tag
