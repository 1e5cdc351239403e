#!/bin/bash
# round 5, first GPU minutes: the EARLY form of ffn_gemm_ring2 — parity (MOEINF_TEST_EXPERIMENTAL=1) and the 512-token sweep
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5a}; mkdir -p "$OUT"
MOEINF_TEST_EXPERIMENTAL=1 timeout 400 python -m pytest tests/test_gpu_fullsize.py -q -x -k experimental > "$OUT/early_parity.log" 2>&1; echo "early parity exit $?"
tail -5 "$OUT/early_parity.log"
SWEEP_ENVS="MOEINF_RING2_EARLY=0;MOEINF_RING2_EARLY=1;MOEINF_RING2_EARLY=0;MOEINF_RING2_EARLY=1" timeout 300 python tools/ffn_sweep.py mixtral_8x7b:512:4 mixtral_8x7b:384:4 mixtral_8x7b:768:4 > "$OUT/ffn_sweep_ring2_early.txt" 2>&1
cat "$OUT/ffn_sweep_ring2_early.txt"
