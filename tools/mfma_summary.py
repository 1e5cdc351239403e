#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass over SQ_VALU_MFMA_BUSY_CYCLES (+ GRBM_GUI_ACTIVE) into per-kernel
MFMA-pipe utilisation: busy cycles summed over the chip / (kernel cycles x 1024 SIMDs).  Joined with the
kernel trace of the same run for the durations.
usage: mfma_summary.py <counter_collection.csv> <kernel_trace.csv> <out.json>"""
import collections
import csv
import json
import sys

SIMDS = 256 * 4
per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
names = {}
for r in csv.DictReader(open(sys.argv[1])):
    d = r["Dispatch_Id"]
    per_dispatch[d][r["Counter_Name"]] += float(r["Counter_Value"])
    names[d] = r["Kernel_Name"]
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d, c in per_dispatch.items():
    k = names[d]
    if "moeinf::ffn" not in k:
        continue
    k = k.split("(")[0].replace("void ", "")
    for n, v in c.items():
        agg[k][n].append(v)
    if d in dur:
        agg[k]["duration_ns"].append(dur[d])
out = {}
for k, c in agg.items():
    e = {n: sum(v) / len(v) for n, v in c.items()}
    e["launches"] = len(next(iter(c.values())))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs
        e["mfma_busy_frac_by_gui_active"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "duration_ns" in e:
        e["mfma_busy_frac_by_duration_at_2p4GHz"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["duration_ns"] * 2.4 * SIMDS)
    out[k] = e
json.dump({"note": "per launch averages of a counter pass; busy fraction = SQ_VALU_MFMA_BUSY_CYCLES (summed over 1024 SIMDs) / (kernel cycles x 1024)",
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
