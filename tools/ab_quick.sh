python -m pytest tests/test_gpu_multiframe.py tests/test_gpu_classification_adversarial.py tests/test_gpu_fuzz.py tests/test_gpu_ref_kernel.py tests/test_gpu_labels.py -x -q -m gpu 2>&1 | tail -2
for g in 200 384 512; do for v in 0 11; do
  python bench.py --workload ssurf --grid $g --variant $v --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ssurf grid $g variant $v', d['ms_per_step'], d['value'])"
done; done
python bench.py --workload traj --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('traj', d['ms_per_step'], d['value'])"
