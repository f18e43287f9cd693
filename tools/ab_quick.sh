python -m pytest tests/test_gpu_multiframe.py tests/test_gpu_classification_adversarial.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
for w in ssurf traj "sfull --mode fused" "sband --mode fused"; do
  python bench.py --workload $w --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['value'])"
done
