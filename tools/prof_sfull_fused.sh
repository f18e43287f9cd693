export TMPDIR=/tmp
cd /tmp
for v in 8 11; do
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02r_sfull_v$v -o st --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload sfull --mode fused --variant $v --no-extras --no-traffic --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02r_sfull_v$v.json 2>/dev/null
echo variant $v; cut -c1-150 $GRAFT_REPO_ROOT/gpurun_out/r02r_sfull_v$v/st_kernel_stats.csv | head -5
done
