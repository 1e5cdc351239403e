#!/bin/bash
# The offload regime on its own (expert cache = half of the expert bytes): every sub-leg of bench.py's offload leg, per setting of
# the tier mover given as "KEY=VAL[,KEY=VAL]" (e.g. MOEINF_H2D_PULL=0), each run twice (A/B/A/B).
# Usage: bash tools/offload_ab.sh <outdir> <workload> [setting ...]      setting "base" = the defaults
set -u
OUT=${1:?outdir}; WL=${2:-deepseek-v2-lite}; shift; shift; mkdir -p "$OUT"
for rnd in 0 1; do
for setting in ${@:-base}; do
  tag=${setting//[=,]/_}
  envs=(); [ "$setting" != base ] && IFS=',' read -r -a envs <<< "$setting"
  env "${envs[@]}" timeout 900 python bench.py --workload $WL --no-cpu-baseline --no-other-configs --no-traffic --no-dropin \
    --miss-heavy-frac 0.5 --prompt 0 --windows 1 --steps 6 --warmup 2 ${OFFLOAD_AB_FLAGS:-} > "$OUT/offload_${tag}_$rnd.json" 2> "$OUT/offload_${tag}_$rnd.err"
  cp bench_details.json "$OUT/offload_${tag}_${rnd}_details.json"
  python - "$OUT/offload_${tag}_${rnd}_details.json" "$setting" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); m = d["miss_heavy"]
print("%-28s" % sys.argv[2], "ms/token", m["ms_per_token"], "h2d GB/s", m["h2d_GBps"], "over pcie bound", m["ms_per_token_over_pcie_bound"], "hit", m["hit_rate"], "| warm-up stream GB/s", (d.get("prefetch_stream") or {}).get("GBps"), "| decode ms/token", d["ms_per_step"])
if "-v" in sys.argv:
    for s in m["sub_legs"]:
        print("   %-8s %-11s %-20s %-11s attn %5.1f ms/token %8.3f hit %.3f overlap %.3f issued %4d useful %4d GB/s %s" % (
            s["routing"], s["policy"], s.get("speculation_kind"), s.get("activations", "")[:11], s["attention_standin_us_per_layer"], s["ms_per_token"], s["hit_rate"], s["overlap"] or 0,
            s["prefetch_issued"], s["prefetch_useful"], s["h2d_GBps"]))
PY
done
done
