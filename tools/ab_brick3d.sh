one() { python bench.py "$@" --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for w in "ssurf" "traj" "ssurf --grid 200 --voxel-mm 4" "ssurf --grid 384" "sfull --mode fused"; do
  echo "$w: $(one --workload $w)"
done
