#!/bin/bash
# DeepSeek-V2-Lite decode: stage-2 half-tile kernel vs the arrival-counter form, batch-depth sweeps; predictor test; min_share sweep
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3d}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_chained.py tests/test_gpu_interface.py -m gpu -q -rf -k "deepseek or chained or predictor or batch1 or decode" > "$OUT/pytest.log" 2>&1
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
run half_u8 A=1
run old_arrival MOEINF_DEC1_HALF=0
run half_u4 MOEINF_DEC1_HALF_U=4
run sr_u8 MOEINF_SR_U=8
run sh1_u8 MOEINF_SH1_U=8
run sr_u8_sh1_u8 MOEINF_SR_U=8 MOEINF_SH1_U=8
run half_u8_again A=1
POL="lfu+engine_predictor_la2+governor"
for ms in 0.25 0.4; do
  timeout 300 python tools/prefetch_study.py --workload deepseek_v2_lite --layers 26 --cache-frac 0.25 --policies "$POL" --min-share $ms > "$OUT/ps_ds_ms$ms.jsonl" 2> "$OUT/ps_ds_ms$ms.err"; cut -c1-400 "$OUT/ps_ds_ms$ms.jsonl"
  timeout 300 python tools/prefetch_study.py --workload mixtral_8x7b --layers 8 --policies "$POL" --min-share $ms > "$OUT/ps_mx_ms$ms.jsonl" 2> "$OUT/ps_mx_ms$ms.err"; cut -c1-400 "$OUT/ps_mx_ms$ms.jsonl"
done
