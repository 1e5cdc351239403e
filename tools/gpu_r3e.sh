#!/bin/bash
# compute-bound grouped GEMM (ffn_gemm_big): parity tests, prefill sweep vs the round-2 kernels; DeepSeek stage-2 batch depth
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3e}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -rf -k "compute_bound or prefill or many_tokens or long_prefill or gemm" > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -14
SWEEP_ENVS="A=1;MOEINF_GEMM_BIG=0;MOEINF_GEMM_BIG_ROWS=128" timeout 900 python tools/ffn_sweep.py mixtral_8x7b:512:2 mixtral_8x7b:2048:2 mixtral_8x7b:4096:2 deepseek_v2_lite:4096:4 nllb_moe_54b:8192:1 2>&1 | tee "$OUT/ffn_sweep_prefill.txt" | tail -20
LEAN="--workload deepseek-v2-lite --no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --prompt 0 --windows 3"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py $LEAN > "$OUT/ds_$tag.json" 2> "$OUT/ds_$tag.err"; python - "$OUT/ds_$tag.json" "$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["kernels"]
    print(f"{sys.argv[2]:28s} ms/token {d['ms_per_step']:.4f} windows {d['windows_ms']} ffn1 {k['ffn_stage1']['avg_launch_us']} ffn2 {k['ffn_stage2']['avg_launch_us']} route {k['route(gate+topk+index)']['avg_launch_us']}")
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
run defaults A=1
run dec1_u8 MOEINF_DEC1_U=8
run dec1_u12 MOEINF_DEC1_U=12
run half_u12 MOEINF_DEC1_HALF=1 MOEINF_DEC1_HALF_U=12
