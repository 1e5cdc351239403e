#!/bin/bash
# round 5, closing evidence: the default bench line (as the driver runs it) + a two-rank line on one GPU over the peer-store transport
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5z}; mkdir -p "$OUT"
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"; echo "bench exit $?"
tail -3 "$OUT/bench_default.time"; grep -E "live traffic|failed|FAILED" "$OUT/bench_default.err" | tail -6
MOEINF_BENCH_SHARE_GPU0=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-other-configs --layers 8 --cpu-sample-layers 4 --cpu-sample-steps 2 > "$OUT/bench_two_ranks_one_gpu.json" 2> "$OUT/bench_two_ranks_one_gpu.err"; echo "two-rank bench exit $?"
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "windows", d["windows_ms"], "parity", d["parity"]["ok"], "frac", d["roofline"]["frac"], "traffic x", d["roofline"].get("traffic_over_algorithmic"))
print("prefill", d["prefill"]["ms_all_layers"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], (d["cpu_baseline"].get("reference_compiled") or {}).get("value"))
m=d["miss_heavy"]; print("miss", m["ms_per_token"], m["hit_rate"], m["ms_per_token_over_pcie_bound"])
for s in m["sub_legs"]: print("   ", s["routing"], s["policy"], "hit", s["hit_rate"], "ms", s["ms_per_token"])
for o in d["other_configs"]:
    print(" other:", o.get("workload","")[:64], o.get("ms_per_step"), o.get("error"), "mean_rel", (o.get("parity") or {}).get("mean_rel_err"), "max_rel", (o.get("parity") or {}).get("max_rel_err"), "ok", (o.get("parity") or {}).get("ok"), "frac", o.get("frac_of_hbm_peak_whole_step"))
    if o.get("offload_regime"):
        for s in o["offload_regime"]["sub_legs"]: print("      ", s["routing"], s["policy"], s["speculation"][:16], "attn", s["attention_standin_us_per_layer"], "hit", s["hit_rate"], "ms", s["ms_per_token"], "ovl", s["overlap"], "pf", s["prefetch_issued"], s["prefetch_useful"], "exposed", s["exposed_wait_ms"], "busy", s["h2d_link_busy_ms"])
t=json.loads(open("$OUT/bench_two_ranks_one_gpu.json").read().strip().splitlines()[-1])
print("two ranks:", t["n_gpus"], t["ms_per_step"], t["parity"]["ok"], t["ep_transport"]["chosen"], t["ep_phases_us_per_layer"])
PY
