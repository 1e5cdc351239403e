#!/bin/bash
# The DeepSeek-V2-Lite offload regime on its own (expert cache = half of the expert bytes): every sub-leg of bench.py's
# offload leg — on-demand, routing x policy, attention stand-in, EAM predictor, residual stream + next-layer gate lookahead —
# with the tier mover's whole-blob copies on (default) and off.  Usage: bash tools/offload_ab.sh <outdir> [workload] [blob MiB ...]
set -u
OUT=${1:?outdir}; WL=${2:-deepseek-v2-lite}; shift; shift; mkdir -p "$OUT"
for blob in ${@:-64}; do
  MOEINF_H2D_WHOLE_BLOB_MB=$blob timeout 900 python bench.py --workload $WL --no-cpu-baseline --no-other-configs --no-traffic \
    --miss-heavy-frac 0.5 --prompt 0 --windows 1 --steps 6 --warmup 2 > "$OUT/offload_blob$blob.json" 2> "$OUT/offload_blob$blob.err"
  cp bench_details.json "$OUT/offload_blob${blob}_details.json"
  python - "$OUT/offload_blob${blob}_details.json" "blob<=${blob}MiB" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); m = d["miss_heavy"]
print(sys.argv[2], "ms/token", m["ms_per_token"], "h2d GB/s", m["h2d_GBps"], "over pcie bound", m["ms_per_token_over_pcie_bound"], "hit", m["hit_rate"], "overlap", m["overlap"])
for s in m["sub_legs"]:
    print("   %-8s %-11s %-20s %-11s attn %5.1f ms/token %8.3f hit %.3f overlap %.3f issued %4d useful %4d wasted %s GB/s %s" % (
        s["routing"], s["policy"], s.get("speculation_kind"), s.get("activations", "")[:11], s["attention_standin_us_per_layer"], s["ms_per_token"], s["hit_rate"], s["overlap"] or 0,
        s["prefetch_issued"], s["prefetch_useful"], s.get("prefetch_wasted"), s["h2d_GBps"]))
PY
done
