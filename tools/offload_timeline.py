#!/usr/bin/env python3
"""Where the host link idles in the offload regime: from a rocprofv3 --kernel-trace CSV of a miss-heavy decode run, the gaps
between consecutive pull_retile_kernel launches (the tier mover's copies) and what ran in them.
usage: offload_timeline.py <kernel_trace.csv>"""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("moeinf::", "")))
rows.sort()
pulls = [r for r in rows if "pull_retile" in r[2]]
pulls = pulls[len(pulls) // 2:]  # the timed legs, not the warm-up
busy = sum(e - s for s, e, _ in pulls)
span = pulls[-1][1] - pulls[0][0]
gaps = [(pulls[i + 1][0] - pulls[i][1], pulls[i][1], pulls[i + 1][0]) for i in range(len(pulls) - 1)]
print(f"{len(pulls)} pulls, busy {busy / 1e6:.1f} ms of {span / 1e6:.1f} ms = {busy / span:.3f}; mean pull {busy / len(pulls) / 1e3:.1f} us")
big = [g for g in gaps if g[0] > 20000]
small = [g for g in gaps if g[0] <= 20000]
print(f"gaps <= 20 us: {len(small)} (mean {sum(g[0] for g in small) / max(1, len(small)) / 1e3:.1f} us, total {sum(g[0] for g in small) / 1e6:.2f} ms); gaps > 20 us: {len(big)} (mean {sum(g[0] for g in big) / max(1, len(big)) / 1e3:.1f} us, total {sum(g[0] for g in big) / 1e6:.2f} ms)")
inside = collections.defaultdict(lambda: [0, 0])
for g, a, b in big[:400]:
    for s, e, n in rows:
        if e > a and s < b and "pull_retile" not in n:
            inside[n[:60]][0] += 1
            inside[n[:60]][1] += min(e, b) - max(s, a)
for n, (c, t) in sorted(inside.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"   in the big gaps: {n:60s} {c:5d} x, {t / max(1, c) / 1e3:7.1f} us each")
hist = collections.Counter(min(20, g[0] // 25000) for g in big)
print("big-gap histogram (25-us bins):", dict(sorted(hist.items())))
