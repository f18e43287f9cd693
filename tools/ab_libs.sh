# same-box comparison of several builds of the library:  bash tools/ab_libs.sh "<lib1.so lib2.so ...>" "<bench args>" ...
# ("default" = the in-tree libtsdf_hip.so; every build is run twice, interleaved)
libs=$1; shift
for a in "$@"; do for rep in 1 2; do for l in $libs; do
  if [ "$l" = default ]; then lp=""; else lp="$PWD/$l"; fi
  echo "$a | $l: $(TSDF_HIP_LIB=$lp python bench.py $a --no-extras --no-traffic --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip()); print(d['ms_per_step'], d['value'])")"
done; done; done
