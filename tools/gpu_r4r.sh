#!/bin/bash
# ring2 as shipped: parity of every tile form at full size, then the sweep over token counts against MOEINF_GEMM_RING2=0
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4r}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -s -k "mixtral or many_experts" 2>&1 | grep -E "passed|failed|Error|assert|mixtral t=|FAILED" | tee "$OUT/parity.txt"
SWEEP_ENVS=";MOEINF_GEMM_RING2=0" timeout 400 python tools/ffn_sweep.py mixtral_8x7b:320:2 mixtral_8x7b:384:2 mixtral_8x7b:512:2 mixtral_8x7b:640:2 mixtral_8x7b:768:2 2>&1 | tee "$OUT/sweep.txt"
