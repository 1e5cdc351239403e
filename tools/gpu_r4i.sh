#!/bin/bash
# 512-token Mixtral prefill: where does the time go?  (a) kernel selection variants,
# (b) PMC: fabric traffic of both stages (FETCH_SIZE), (c) PMC: wave-cycle buckets, LDS conflicts, MFMA-busy.
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/${1:-r4i}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
SWEEP_ENVS=";MOEINF_GEMM_RING2=0;MOEINF_FFN_GEMM_RGB=8;" timeout 400 python tools/ffn_sweep.py mixtral_8x7b:512:2 2>&1 | tee "$OUT/sweep.txt"
pmc() {  # tag, counters...
  local tag=$1; shift
  (cd /tmp && timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$OUT/pmc_$tag" -o m -- \
      python "$R/tools/prefill_once.py" mixtral_8x7b 512 2 8 > /dev/null 2> "$R/$OUT/pmc_$tag.err")
  python tools/pmc_kernel_means.py "$OUT/pmc_$tag/m_counter_collection.csv" ffn_gemm > "$OUT/pmc_$tag.json" 2>> "$OUT/pmc_$tag.err"
  rm -rf "$OUT/pmc_$tag"
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pmc sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
for f in fetch write tcc sq1 sq2; do echo "== $f"; cat "$OUT/pmc_$f.json"; tail -2 "$OUT/pmc_$f.err"; done
