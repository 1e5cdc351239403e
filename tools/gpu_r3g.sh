#!/bin/bash
# PMC passes over the compute-bound GEMM (Mixtral, 4096 tokens, 1 layer): SQ busy/stall/LDS-conflict counters, then L2 hit rate
set -u
export TMPDIR=/tmp
R=$(pwd); OUT=gpurun_out/${1:-r3g}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$R/$OUT/pmc_sq" -o m -- python "$R/tools/prefill_once.py" mixtral_8x7b 4096 1 6 > "$R/$OUT/pmc_sq.log" 2>&1)
python tools/pmc_kernel_means.py $(ls $OUT/pmc_sq/*counter_collection.csv | head -1) ffn_gemm | tee "$OUT/pmc_sq_means.json"
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$R/$OUT/pmc_tcc" -o m -- python "$R/tools/prefill_once.py" mixtral_8x7b 4096 1 6 > "$R/$OUT/pmc_tcc.log" 2>&1)
python tools/pmc_kernel_means.py $(ls $OUT/pmc_tcc/*counter_collection.csv | head -1) ffn_gemm | tee "$OUT/pmc_tcc_means.json"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/$OUT/pmc_fetch" -o m -- python "$R/tools/prefill_once.py" mixtral_8x7b 4096 1 6 > "$R/$OUT/pmc_fetch.log" 2>&1)
python tools/pmc_kernel_means.py $(ls $OUT/pmc_fetch/*counter_collection.csv | head -1) ffn_gemm | tee "$OUT/pmc_fetch_means.json"
tail -3 "$OUT/pmc_sq.log"
rm -rf "$OUT"/pmc_*/ 2>/dev/null
