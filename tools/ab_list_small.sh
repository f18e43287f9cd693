for g in 200 384; do for v in 0 13 0 13; do echo "ssurf $g variant $v: $(python bench.py --workload ssurf --grid $g --variant $v --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip()); print(d['ms_per_step'], d['value'])")"; done; done
for v in 0 13 0 13; do echo "ssurf noisy 512 variant $v: $(python bench.py --workload ssurf --noise-mm 2 --holes 0.05 --variant $v --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip()); print(d['ms_per_step'], d['value'])")"; done
for v in 0 13; do echo "sfull 512 fused variant $v: $(python bench.py --workload sfull --mode fused --variant $v --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip()); print(d['ms_per_step'], d['value'])")"; done
python tools/batch_time.py --n 16 2>&1 | head -4

bash tools/ab_list.sh
