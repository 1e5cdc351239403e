#!/bin/bash
# round 4, call C: full GPU test suite (new: peer-store tests, fp32-exact arms, serving oracle arm), smoke, default bench line,
# then the speculation study with MEASURED attention times (profiles/r04_attention_block_time_stock_pytorch.jsonl)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4c; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rf -s > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|exact\| = |vs oracle chain|reference-caller replay" "$OUT/pytest_gpu.log" | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit $?"; tail -3 "$OUT/bench_default.err"
P="lfu,lfu+engine_predictor_la1,lfu+engine_predictor_la2+governor,lfu+engine_predictor_la8+governor"
timeout 500 python tools/prefetch_study.py --workload mixtral_8x7b --layers 8 --attn-us 225 --policies $P --note "attention block of Mixtral-8x7B at batch 1, 2k context: 225 us (stock PyTorch-ROCm)" > "$OUT/ps_mixtral_b1_ctx2k.jsonl" 2> "$OUT/ps1.err"
timeout 500 python tools/prefetch_study.py --workload mixtral_8x7b --layers 8 --attn-us 763 --batch 8 --policies $P --note "batch 8, 8k context: 763 us" > "$OUT/ps_mixtral_b8_ctx8k.jsonl" 2> "$OUT/ps2.err"
timeout 400 python tools/prefetch_study.py --workload deepseek_v2_lite --layers 26 --cache-frac 0.25 --attn-us 270 --policies $P --note "attention block of a DeepSeek-V2-Lite-shaped layer at batch 1, 2k context: 270 us" > "$OUT/ps_deepseek_b1_ctx2k.jsonl" 2> "$OUT/ps3.err"
timeout 400 python tools/prefetch_study.py --workload deepseek_v2_lite --layers 26 --cache-frac 0.25 --attn-us 1499 --batch 8 --policies $P --note "batch 8, 8k context: 1499 us" > "$OUT/ps_deepseek_b8_ctx8k.jsonl" 2> "$OUT/ps4.err"
for f in "$OUT"/ps_*.jsonl; do echo "== $f"; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d["policy"], d["ms_per_token"], "hit", d["hit_rate"], "pf", d["prefetch_issued"], d["prefetch_useful"], "h2dGiB", d["h2d_GiB"], "exposed", d["exposed_wait_ms"], "compute-only", d.get("compute_only_ms_per_token"))
PY
done
tail -2 "$OUT"/ps*.err | tail -12
