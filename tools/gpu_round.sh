#!/bin/bash
# One gpurun call: full GPU suite, the profiler passes of the default bench command, a 2-rank gloo rehearsal.
# Usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag>
set -o pipefail
tag=${1:-r02}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu -rs > $out/${tag}_gputests.log 2>&1 || { tail -30 $out/${tag}_gputests.log; exit 1; }
tail -3 $out/${tag}_gputests.log
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { tail -20 $out/${tag}_bench.err; exit 1; }
rm -rf $out/${tag}_prof_stats
( cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/${tag}_prof_stats -o st --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-traffic --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/${tag}_bench_prof.json 2> $GRAFT_REPO_ROOT/$out/${tag}_bench_prof.err ) || { tail -20 $out/${tag}_bench_prof.err; exit 1; }
TSDF_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $out/${tag}_bench_gloo2.json 2> $out/${tag}_bench_gloo2.err || { tail -20 $out/${tag}_bench_gloo2.err; exit 1; }
python bench.py --workload ssurf --no-cpu-baseline > $out/${tag}_bench_ssurf.json 2> $out/${tag}_bench_ssurf.err || { tail -20 $out/${tag}_bench_ssurf.err; exit 1; }
python bench.py --workload traj --no-cpu-baseline > $out/${tag}_bench_traj.json 2> $out/${tag}_bench_traj.err || { tail -20 $out/${tag}_bench_traj.err; exit 1; }
python tools/batch_time.py > $out/${tag}_batch_time.log 2>&1 || { tail -20 $out/${tag}_batch_time.log; exit 1; }
echo done
