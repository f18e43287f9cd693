# same-box A/B of two builds of the library on the fused S-surf path:  bash tools/ab_lib.sh <other .so> [grids...]
lib=$1; shift
for g in "$@"; do for l in "" "$lib" "" "$lib"; do echo "ssurf $g ${l:-default}: $(TSDF_HIP_LIB=${l:+$PWD/$l} python bench.py --workload ssurf --grid $g --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip()); print(d['ms_per_step'], d['value'])")"; done; done
