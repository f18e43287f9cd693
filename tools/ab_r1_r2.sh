python tools/probe_sband.py --variants 87,83,87,83
python tools/probe_sband.py --workload sfull --variants 87,83,87,83
for v in 87 83 87 83; do python bench.py --workload ssurf --mode frame --variant $v --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ssurf frame variant $v', d['ms_per_step'], d['value'])"; done
