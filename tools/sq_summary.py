#!/usr/bin/env python3
"""Per-dispatch averages of the SQ counters tools/sq_counters.sh collected for the dominant fused kernel, with the derived
per-wavefront / per-voxel-frame figures DESIGN.md section 4 argues with.  Writes gpurun_out/sq_<tag>.json."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc, kernel = {}, None
for part in "ab":
    f = glob.glob(os.path.join(root, "gpurun_out", f"sq_{tag}_{part}", "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "integrate_multi_inline<" in r["Kernel_Name"] or "integrate_brick_list<" in r["Kernel_Name"]]
    # the dominant instantiation (most dispatches), full-size launches only (the largest grid)
    names = {}
    for r in rows:
        names[r["Kernel_Name"]] = names.get(r["Kernel_Name"], 0) + 1
    kernel = max(names, key=names.get)
    rows = [r for r in rows if r["Kernel_Name"] == kernel]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in by.items():
        v = v[len(v) // 4:]                    # drop the warm-up launches
        acc[k] = sum(v) / len(v)
line = json.loads(open(os.path.join(root, "gpurun_out", f"sq_{tag}_a.json")).read().strip().splitlines()[-1])
fpl = line["config"]["frames_per_launch"]
vox = line["config"]["grid"][0] * line["config"]["grid"][1] * line["config"]["grid"][2]
waves = acc["SQ_WAVES"]
d = {"valu_insts_per_wave": acc["SQ_INSTS_VALU"] / waves, "salu_insts_per_wave": acc["SQ_INSTS_SALU"] / waves,
     "valu_insts_per_voxel_frame": acc["SQ_INSTS_VALU"] * 64 / (vox * fpl) / 1.0,
     "frames_per_launch": fpl, "bench_ms_per_step": line["ms_per_step"], "bench_value": line["value"]}
cyc = acc["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
d["kernel_cycles_per_xcd"] = cyc
d["mean_waves_per_simd"] = acc["SQ_WAVE_CYCLES"] * 4 / (cyc * 1024)          # quad-cycles -> cycles, 1024 SIMDs
d["cycles_per_wave_frame_per_simd"] = cyc / (waves / 1024 * fpl)
d["valu_insts_per_wave_frame"] = d["valu_insts_per_wave"] / fpl
d["cycles_per_valu_inst_if_valu_alone"] = cyc / (acc["SQ_INSTS_VALU"] / 1024)
d["scalar_unit_busy_if_1p19_cycles_each"] = acc["SQ_INSTS_SALU"] / 256 * 1.19 / cyc
acc["_derived"] = d
acc["_note"] = (f"per-dispatch averages of {kernel} ({line['config']['workload'][:60]}...), two rocprofv3 --pmc passes; SQ_WAVE_CYCLES / "
                "SQ_WAIT_* count quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; valu_insts_per_voxel_frame counts a wave "
                "instruction as 64 lane-instructions over the voxels x frames of the launch (classified wavefront-frames included)")
json.dump(acc, open(os.path.join(root, "gpurun_out", f"sq_{tag}.json"), "w"), indent=1)
print(json.dumps(acc, indent=1))
