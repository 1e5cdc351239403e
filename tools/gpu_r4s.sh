#!/bin/bash
# (as run: ring2's upper bound was 256 by default then and the big kernel had a second threshold knob, since removed)
# where should ring2 start and stop?  low end against the hybrid kernel, high end against the big-tile kernel
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4s}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
SWEEP_ENVS=";MOEINF_RING2_MIN_ROWS_GATED=64,MOEINF_RING2_MIN_ROWS_PLAIN=64;MOEINF_RING2_MIN_ROWS_GATED=32,MOEINF_RING2_MIN_ROWS_PLAIN=32" timeout 400 python tools/ffn_sweep.py mixtral_8x7b:96:2 mixtral_8x7b:160:2 mixtral_8x7b:224:2 mixtral_8x7b:288:2 mixtral_8x7b:336:2 2>&1 | tee "$OUT/sweep_low.txt"
SWEEP_ENVS=";MOEINF_RING2_MAX_ROWS=340,MOEINF_GEMM_BIG_ROWS=340" timeout 300 python tools/ffn_sweep.py mixtral_8x7b:704:2 mixtral_8x7b:768:2 mixtral_8x7b:832:2 mixtral_8x7b:896:2 2>&1 | tee "$OUT/sweep_high.txt"
