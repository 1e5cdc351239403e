#!/bin/bash
# round 4, call E: fixed fp16 / exchange tests; forced-EP lines after the two-level arrival + stage records; DeepSeek decode A/B
# (MOEINF_SH1_NW=8, MOEINF_SR_ORDER=1); Switch-base-8 with twelve tiles per batch in stage 2
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4e; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ep_peer.py tests/test_gpu_ep_processes.py tests/test_gpu_interface.py -m gpu -q -rf -s -k "fp16 or switch or peer or processes or expert_parallel or ep_" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|fp16 t=|switch " "$OUT/pytest_gpu.log" | tail -20
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0"
for wl in mixtral-8x7b deepseek-v2-lite; do
  timeout 300 python bench.py --workload $wl --force-ep --ep-transport peer-store --no-other-configs --miss-heavy-frac 0 --prompt 0 --cpu-sample-layers 2 --cpu-sample-steps 2 > "$OUT/bench_ep1_${wl}_peer-store.json" 2> "$OUT/bench_ep1_${wl}.err"
  python - "$OUT/bench_ep1_${wl}_peer-store.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("forced EP", d["ms_per_step"], d["ep_transport"]["chosen"], d["parity"]["ok"], d.get("ep_phases_us_per_layer"))
except Exception as ex: print("no line", ex)
PY
done
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --workload deepseek-v2-lite $LEAN > "$OUT/ds_$tag.json" 2> "$OUT/ds_$tag.err"; python - "$OUT/ds_$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["ms_per_step"], d["windows_ms"])
except Exception as ex: print(sys.argv[2], "no line", ex)
PY
}
run base A=1
run sh1nw8 MOEINF_SH1_NW=8
run srorder MOEINF_SR_ORDER=1
run both MOEINF_SH1_NW=8 MOEINF_SR_ORDER=1
run base2 A=1
for u in 12 4; do
MOEINF_DEC1_SWITCH_U=$u timeout 200 python bench.py --workload switch-base-8 $LEAN > "$OUT/switch_u$u.json" 2> "$OUT/switch_u$u.err"
python - "$OUT/switch_u$u.json" $u <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("switch U", sys.argv[2], d["ms_per_step"], d["windows_ms"])
except Exception as ex: print("no line", ex)
PY
done
