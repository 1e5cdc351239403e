#!/bin/bash
# big-GEMM iteration: the parity tests that reach ffn_gemm_big, then a sweep.  usage: gpu_r3m.sh <tag> [sweep specs]; SWEEP_ENVS as for ffn_sweep.py
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3m}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -rf -k "t2048 or t4096 or 256x256 or t512 or short_last_pass" > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -14
SWEEP_ENVS="${SWEEP_ENVS:-A=1}" timeout 900 python tools/ffn_sweep.py ${2:-mixtral_8x7b:4096:2 deepseek_v2_lite:4096:4 mixtral_8x7b:2048:2} 2>&1 | tee "$OUT/ffn_sweep.txt" | tail -20
