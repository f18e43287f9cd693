# HBM-side bytes per dispatch of the dominant kernel of one bench invocation, raw counters (KiB):
#   bash tools/pmc_bytes.sh <tag> <kernel substring> <bench args...>
tag=$1; ksub=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_${tag}_$c
  rocprofv3 --pmc $c -d $out/pmc_${tag}_$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 64 --warmup 32 --no-extras --no-traffic --no-cpu-baseline > $out/pmc_${tag}_$c.json 2> $out/pmc_${tag}_$c.err || exit 1
done
python3 - <<PY
import csv, glob, statistics
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$out/pmc_${tag}_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "$ksub" in r["Kernel_Name"]]
    v = v[len(v) // 4:]
    print("$tag", c, "dispatches", len(v), "median KiB", statistics.median(v), "= MB", statistics.median(v) * 1024 / 1e6)
PY
