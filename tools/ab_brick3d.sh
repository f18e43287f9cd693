one() { python bench.py "$@" --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for extra in "" "--noise-mm 2" "--holes 0.05" "--noise-mm 2 --holes 0.05" "--noise-mm 2 --holes 0.15"; do
  echo "ssurf 512 $extra: $(one --workload ssurf $extra)   traj 1024 $extra: $(one --workload traj $extra)   ssurf 200@4mm $extra: $(one --workload ssurf --grid 200 --voxel-mm 4 $extra)"
done
