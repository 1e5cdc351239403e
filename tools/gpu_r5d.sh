#!/bin/bash
# round 5: the one-launch decode layer with stage-2 weights requested ahead of h — parity, then the variants against the three launches
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5d}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_chained.py -q -x -k "deepseek or DeepSeek or shared or batch1 or selfrout" > "$OUT/pytest_deepseek.log" 2>&1; echo "pytest exit $?"
tail -4 "$OUT/pytest_deepseek.log"
SWEEP_ENVS="MOEINF_LAYER1=0;MOEINF_LAYER1=1;MOEINF_LAYER1_U1=4;MOEINF_LAYER1_U1=4,MOEINF_LAYER1_PRE=4;MOEINF_LAYER1_PRE=4;MOEINF_LAYER1=0;MOEINF_LAYER1=1" timeout 400 python tools/ffn_sweep.py deepseek_v2_lite:1:26 > "$OUT/ffn_sweep_layer1.txt" 2>&1
cat "$OUT/ffn_sweep_layer1.txt"
for v in "MOEINF_LAYER1=0" "MOEINF_LAYER1=1" "MOEINF_LAYER1_U1=4"; do
  env $v timeout 200 python bench.py --workload deepseek-v2-lite --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 3 > "$OUT/bench_ds_$v.json" 2> "$OUT/bench_ds_$v.err"; echo "bench $v exit $?"
  python - <<PY
import json
d=json.loads(open("$OUT/bench_ds_$v.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["windows_ms"])
PY
done
