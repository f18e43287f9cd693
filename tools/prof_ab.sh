# rocprofv3 kernel stats of one bench workload under several kernel variants:  bash tools/prof_ab.sh <workload> <grid> <variant>...
cd /tmp && export TMPDIR=/tmp
w=$1; g=$2; shift; shift
for v in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_${w}${g}_v$v
  rm -rf $out
  rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --grid $g --variant $v --no-extras --no-traffic --no-cpu-baseline > $out.json 2> $out.err
  python3 - <<PY
import csv, json
d = json.load(open("$out.json"))
print("== $w $g variant $v: ms_per_step", d["ms_per_step"])
for r in csv.DictReader(open("$out/p_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.1:
        print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.1f} ms  {float(r['Percentage']):5.1f}%")
PY
done
