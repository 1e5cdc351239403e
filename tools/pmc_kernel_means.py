#!/usr/bin/env python3
"""per-kernel means of every counter in a rocprofv3 --pmc counter_collection.csv (kernels whose name contains argv[2])"""
import collections, csv, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
per = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
for r in csv.DictReader(open(sys.argv[1])):
    per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
for d, c in per.items():
    k = names[d].split("(")[0].replace("void ", "")
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    for n, v in c.items(): agg[k][n].append(v)
out = {k: {n: sum(v) / len(v) for n, v in c.items()} | {"launches": len(next(iter(c.values())))} for k, c in agg.items()}
print(json.dumps(out, indent=1))
