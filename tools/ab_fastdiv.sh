python -m pytest tests/test_gpu_fastdiv.py tests/test_gpu_multiframe.py tests/test_gpu_ref_kernel.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sband or sfull or variant" 2>&1 | tail -3
for w in "sband --mode fused" "ssurf" "traj"; do python bench.py --workload $w --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['value'])"; done
