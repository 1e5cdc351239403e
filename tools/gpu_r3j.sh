#!/bin/bash
# (round 3 study; MOEINF_RING_K4 and the ffn_gemm_ring kernel it selected were removed in round 4 — kept as the record of what
#  profiles/r03_ffn_sweep_prefill_ring_k4.txt ran)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3j}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rf -k "mixtral" > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -14
SWEEP_ENVS="${SWEEP_ENVS:-A=1;MOEINF_RING_K4=3;MOEINF_RING_K4=0}" timeout 900 python tools/ffn_sweep.py ${2:-mixtral_8x7b:512:2 mixtral_8x7b:384:2 mixtral_8x7b:640:2} 2>&1 | tee "$OUT/ffn_sweep_prefill.txt" | tail -20
