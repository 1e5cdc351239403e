#!/bin/bash
# round 5: the widened default bench line (fp16 legs, offload regime matrix, reference-compiled baseline, live traffic) + new tests
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5i}; mkdir -p "$OUT"
( time timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"; echo "bench exit $?"
tail -3 "$OUT/bench_default.time"; grep -E "live traffic|failed|FAILED|Error|error" "$OUT/bench_default.err" | tail -10
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"], "parity", d["parity"]["ok"], "traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_over_algorithmic"), d["roofline"].get("traffic_live_attempt"))
    print("cpu", {k: d["cpu_baseline"][k] for k in ("value","cores","kind")}, "ref", d["cpu_baseline"].get("reference_compiled"))
    m=d["miss_heavy"]; print("miss", m["ms_per_token"], m["hit_rate"], m["overlap"])
    for s in m["sub_legs"]: print("   ", s["routing"], s["policy"], s["speculation"][:12], "hit", s["hit_rate"], "ms", s["ms_per_token"], "ovl", s["overlap"])
    for o in d["other_configs"]:
        print(" other:", o.get("workload","")[:60], o.get("ms_per_step"), o.get("error"), "mean_rel", (o.get("parity") or {}).get("mean_rel_err"), "max_rel", (o.get("parity") or {}).get("max_rel_err"), "ok", (o.get("parity") or {}).get("ok"))
        if o.get("offload_regime"):
            for s in o["offload_regime"]["sub_legs"]: print("      ", s["routing"], s["policy"], s["speculation"][:12], "attn", s["attention_standin_us_per_layer"], "hit", s["hit_rate"], "ms", s["ms_per_token"], "moe-only", s["moe_ms_per_token_without_the_standin"], "ovl", s["overlap"], "GBps", s["h2d_GBps"], "exposed", s["exposed_wait_ms"])
except Exception as ex:
    print("parse failed", ex)
PY
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "fp16 or one_launch" > "$OUT/pytest_new.log" 2>&1; echo "pytest exit $?"
tail -15 "$OUT/pytest_new.log"
