# fused-path numbers at the library's own brick choice
one() { python bench.py "$@" --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
echo "ssurf 512: $(one --workload ssurf)"
echo "traj 1024: $(one --workload traj)"
echo "ssurf 384: $(one --workload ssurf --grid 384)"
echo "ssurf 200: $(one --workload ssurf --grid 200)"
echo "ssurf 200 @ 4 mm: $(one --workload ssurf --grid 200 --voxel-mm 4)"
echo "sfull 512 fused: $(one --workload sfull --mode fused)"
echo "sband 512 fused: $(one --workload sband --mode fused)"
python tools/batch_time.py --n 16 2>&1 | grep -v amdgpu
python tools/batch_time.py --n 2 --edge 400 2>&1 | grep -v amdgpu
