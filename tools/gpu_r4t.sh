#!/bin/bash
# ring2 with its shipped thresholds: every full-size parity test + the GEMM-sized parity tests, NLLB / Mixtral sweeps vs RING2=0
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4t}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_interface.py -q 2>&1 | tail -6 | tee "$OUT/parity.txt"
SWEEP_ENVS=";MOEINF_GEMM_RING2=0" timeout 300 python tools/ffn_sweep.py nllb_moe_54b:2048:1 nllb_moe_54b:4096:1 mixtral_8x7b:1024:2 2>&1 | tee "$OUT/sweep.txt"
SWEEP_ENVS=";MOEINF_RING2_MIN_ROWS_PLAIN=16" timeout 200 python tools/ffn_sweep.py mixtral_8x7b:48:2 mixtral_8x7b:96:2 2>&1 | tee -a "$OUT/sweep.txt"
