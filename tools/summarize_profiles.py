#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs a gpurun call left under gpurun_out/ into the small summaries that
are committed under profiles/ (and profiles/pmc_traffic.json, which bench.py reads for
roofline.traffic).

    python tools/summarize_profiles.py <tag> <stats_dir> <fetch_dir> <write_dir> <traffic_key>

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half the bytes of a
wide coalesced streaming read (MI355X_MICROARCH.md, HBM section), so it is doubled; WRITE_SIZE
is exact for 16-byte-per-lane streaming stores.  The two counters need separate passes.
"""
import csv
import glob
import json
import os
import sys


def one(pattern):
    """Newest match (gpurun merges new files next to those of earlier calls)."""
    g = glob.glob(pattern, recursive=True)
    assert g, pattern
    return max(g, key=os.path.getmtime)


def counter_avg(d, counter, kernel_substr):
    rows = list(csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))))
    v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)


def main():
    tag, stats_dir, fetch_dir, write_dir, key = sys.argv[1:6]
    # the default Integrate kernel; bench.py also launches the streaming variant (different template args)
    kernel = sys.argv[6] if len(sys.argv) > 6 else "integrate_multi_inline<1, true, false, false, false, "   # both instantiations (classification on / off)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = open(one(os.path.join(stats_dir, "**", "*_kernel_stats.csv"))).read()
    open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w").write(stats)
    fetch_kib, nf = counter_avg(fetch_dir, "FETCH_SIZE", kernel)
    write_kib, nw = counter_avg(write_dir, "WRITE_SIZE", kernel)
    rec = {"kernel": kernel, "dispatches_fetch_pass": nf, "dispatches_write_pass": nw,
           "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
           "fetch_bytes_corrected_x2": 2 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
           "hbm_bytes_per_launch": 2 * fetch_kib * 1024 + write_kib * 1024,
           "correction": "FETCH_SIZE x2 on gfx950 for wide coalesced reads; WRITE_SIZE exact (MI355X_MICROARCH.md HBM)"}
    json.dump(rec, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
    tpath = os.path.join(out, "pmc_traffic.json")
    allrec = json.load(open(tpath)) if os.path.isfile(tpath) else {}
    allrec[key] = {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "source": f"profiles/{tag}_pmc.json"}
    json.dump(allrec, open(tpath, "w"), indent=1)
    print(json.dumps(rec, indent=1))
    print(stats)


if __name__ == "__main__":
    main()
