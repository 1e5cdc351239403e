#!/bin/bash
# prefill GEMM iteration: parity tests that reach the GEMM kernels + sweep
set -u
OUT=gpurun_out/${1:-gemm}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "prefill or many or mid_sized or long or t512 or golden or variable" > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -20
timeout 600 python tools/ffn_sweep.py ${2:-mixtral_8x7b:512:2 mixtral_8x7b:2048:2 deepseek_v2_lite:512:4} 2>&1 | tail -20
