for args in "--n 8 --edge 128" "--n 64 --edge 100" "--n 64 --edge 64" "--n 16 --edge 64" "--n 4 --edge 100"; do
  python tools/batch_time.py $args --frames 320 2>&1 | grep -v amdgpu | grep -v "1024-voxel\|forced on"
done
