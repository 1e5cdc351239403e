#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3i}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_chained.py tests/test_gpu_interface.py -m gpu -q -rf -k "decode or fused or chained or golden or predictor or shapes or batch1" > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -12
LEAN="--workload deepseek-v2-lite --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 3"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py $LEAN > "$OUT/ds_$tag.json" 2> "$OUT/ds_$tag.err"; python - "$OUT/ds_$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels"]
    print(f"{sys.argv[2]:28s} ms/token {d['ms_per_step']:.4f} windows {d['windows_ms']} ffn1 {k['ffn_stage1']['avg_launch_us']} ffn2 {k['ffn_stage2']['avg_launch_us']} route {k['route(gate+topk+index)']['avg_launch_us']}")
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
run wide A=1
run narrow MOEINF_WIDE_OUT=0
run wide2 A=1
run narrow2 MOEINF_WIDE_OUT=0
