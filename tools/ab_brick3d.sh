one() { python bench.py "$@" --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for w in "ssurf" "traj" "ssurf --grid 200 --voxel-mm 4" "ssurf --grid 384" "ssurf --grid 128"; do
  echo "$w: $(one --workload $w)"
done
python tools/batch_time.py --n 16 --frames 320 2>&1 | grep -v amdgpu | head -3
python tools/batch_time.py --n 1 --frames 320 2>&1 | grep -v amdgpu | head -3
python tools/launch_cost.py 2>&1 | grep -v amdgpu
