# brick shapes (TSDF_BRICK3D="q,r,s") with the rotated slice order, classification by policy
one() { python bench.py "$@" --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for shape in 2,4,8 2,8,4 2,2,16 1,8,8 1,4,16 4,2,8 4,1,16 2,1,32; do
  echo "shape $shape: ssurf512 $(TSDF_BRICK3D=$shape one --workload ssurf --grid 512)  traj1024 $(TSDF_BRICK3D=$shape one --workload traj --grid 1024)  ssurf200@4mm $(TSDF_BRICK3D=$shape one --workload ssurf --grid 200 --voxel-mm 4)"
done
for shape in 5,12,1 2,4,8 5,3,4 2,2,16 1,8,8; do
  echo "16 x 200^3 instance masks, shape $shape:"; TSDF_BRICK3D=$shape python tools/batch_time.py --n 16 2>&1 | grep "batched launch, classification by policy (bricks)\|per-volume launches, classification by policy"
done
