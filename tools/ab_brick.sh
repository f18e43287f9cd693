# bricks per wavefront (default when classifying) vs rows / patches per workgroup (variant 11) on the fused sequence path
python -m pytest tests/test_gpu_multiframe.py tests/test_gpu_classification_adversarial.py tests/test_gpu_ref_kernel.py -x -q -m gpu 2>&1 | tail -3
for lib in ""; do
  for w in ssurf traj "sfull --mode fused" "sband --mode fused"; do for v in 0 11; do
    TSDF_HIP_LIB=${lib:+$PWD/$lib} python bench.py --workload $w --variant $v --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-default} $w variant $v', d['ms_per_step'], d['value'])"
  done; done
done
