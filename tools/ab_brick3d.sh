#!/bin/bash
# Brick shapes of the classified fused launches: TSDF_BRICK3D="q,r,s" = q quads x r rows x s slices per
# wavefront, against the library's own choice; slice order rotated (variant 0) and not (variant 10); with and without the
# super-brick pre-pass (variants 8 / 12).   bash tools/ab_brick3d.sh   (on the GPU box, from the repo root)
one() { python bench.py "$@" --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for w in "ssurf" "traj" "ssurf --grid 200 --voxel-mm 4" "ssurf --grid 384"; do
  echo "== $w: library's choice $(one --workload $w)"
  for shape in 16,4,1 8,8,1 4,8,2 4,4,4 2,8,4 2,4,8 1,8,8; do
    echo "   shape $shape: rotated $(TSDF_BRICK3D=$shape one --workload $w --variant 0)   plain $(TSDF_BRICK3D=$shape one --workload $w --variant 10)"
  done
  echo "   super-brick pre-pass: with $(one --workload $w --variant 8)   without $(one --workload $w --variant 12)"
done
