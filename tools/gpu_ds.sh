#!/bin/bash
# DeepSeek decode iteration: parity tests that touch the path + per-layer timing
set -u
OUT=gpurun_out/${1:-ds}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_interface.py -m gpu -q -x -k "deepseek or fused or shared or blocks" > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -20
timeout 120 python tools/ds_experiments.py base 2>&1 | tail -3
MOEINF_FFN_PAIR=0 timeout 120 python tools/ds_experiments.py base 2>&1 | tail -3
