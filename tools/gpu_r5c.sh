#!/bin/bash
# round 5: the one-launch decode layer (layer_fused.hip) — parity of every DeepSeek-shaped case, then A/B against the three launches
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5c}; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_chained.py tests/test_gpu_interface.py tests/test_gpu_tiers.py -q -x -k "deepseek or DeepSeek or shared or batch1 or selfrout" > "$OUT/pytest_deepseek.log" 2>&1; echo "pytest exit $?"
tail -5 "$OUT/pytest_deepseek.log"
SWEEP_ENVS="MOEINF_LAYER1=0;MOEINF_LAYER1=1;MOEINF_LAYER1=0;MOEINF_LAYER1=1" timeout 300 python tools/ffn_sweep.py deepseek_v2_lite:1:26 > "$OUT/ffn_sweep_layer1.txt" 2>&1
cat "$OUT/ffn_sweep_layer1.txt"
for v in 0 1; do
  MOEINF_LAYER1=$v timeout 200 python bench.py --workload deepseek-v2-lite --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 3 > "$OUT/bench_ds_layer1_$v.json" 2> "$OUT/bench_ds_layer1_$v.err"; echo "bench LAYER1=$v exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_ds_layer1_$v.json").read().strip().splitlines()[-1])
print("LAYER1=$v", d["ms_per_step"], d["windows_ms"])
PY
done
