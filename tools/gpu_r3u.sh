#!/bin/bash
# MFMA-busy of the big GEMM at 4096 tokens (PMC pass, kernel-trace only)
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/${1:-r3u}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$R/$OUT/pmc_mfma4096" -o m -- \
    python "$R/tools/prefill_once.py" mixtral_8x7b 4096 1 6 > /dev/null 2> "$R/$OUT/pmc_mfma4096.err")
python tools/mfma_summary.py "$OUT/pmc_mfma4096/m_counter_collection.csv" "$OUT/pmc_mfma4096/m_kernel_trace.csv" "$OUT/pmc_mfma_prefill4096_mixtral8x7b.json" 2> "$OUT/mfma4096_summary.err"
python tools/pmc_kernel_means.py "$OUT/pmc_mfma4096/m_counter_collection.csv" ffn_gemm | tee "$OUT/pmc_means.json"
rm -rf "$OUT"/pmc_mfma4096
