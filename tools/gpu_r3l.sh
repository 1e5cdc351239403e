#!/bin/bash
# ablation timings of ffn_gemm_big (MOEINF_GEMM_BIG_ABL: 1 no MFMA, 2 no DMA, 3 no LDS reads, 4 MFMA + barrier only, 5 DMA + barrier only).
# The ablation template variants lived in the tree only while this was measured (profiles/r03_ffn_gemm_big_ablation.txt);
# at HEAD the knob does nothing — kept as the record of how the numbers were taken.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3l}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
SWEEP_ENVS="${SWEEP_ENVS:-A=1;MOEINF_GEMM_BIG_ABL=1;MOEINF_GEMM_BIG_ABL=2;MOEINF_GEMM_BIG_ABL=3;MOEINF_GEMM_BIG_ABL=4;MOEINF_GEMM_BIG_ABL=5}" timeout 900 python tools/ffn_sweep.py ${2:-mixtral_8x7b:4096:2} 2>&1 | tee "$OUT/ffn_sweep_abl.txt" | tail -20
