#!/bin/bash
# ring2 with a split tail (half workgroups in a half-empty last round): parity at full size (bf16 + fp16), then A/B against MOEINF_RING2_TAIL=0
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4w}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 240 python -m pytest tests/test_gpu_fullsize.py -q -k "mixtral or many_experts" 2>&1 | tail -3 | tee "$OUT/parity.txt"
SWEEP_ENVS=";MOEINF_RING2_TAIL=0" timeout 200 python tools/ffn_sweep.py mixtral_8x7b:512:2 mixtral_8x7b:384:2 mixtral_8x7b:704:2 2>&1 | tee "$OUT/sweep.txt"
