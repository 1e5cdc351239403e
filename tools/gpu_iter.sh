#!/bin/bash
# kernel iteration on the GPU box: the decode-path parity tests, then lean bench lines (no CPU baseline / extra legs)
# usage: tools/gpu_iter.sh <tag> [pytest -k expression] [workloads...]
set -u
OUT=gpurun_out/${1:-iter}; mkdir -p "$OUT"
K=${2:-"batch1 or decode or fused or golden or 8x7b or v2_lite"}
WLS=${3:-"mixtral-8x7b deepseek-v2-lite"}
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 600 python -m pytest tests -m gpu -q -x -k "$K" > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -15
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 3 --prompt 0"
for wl in $WLS; do
  timeout 300 python bench.py --workload $wl $LEAN > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err" || tail -5 "$OUT/bench_$wl.err"
  python - "$OUT/bench_$wl.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(d["config"]["workload"][:24], "ms/step", d["ms_per_step"], d["windows_ms"], {n:(v["avg_launch_us"] if isinstance(v,dict) else v) for n,v in k.items() if isinstance(v,dict)})
PY
done
