#!/bin/bash
# memory-path counters of the 512-token prefill kernels against the batch-1 decode kernels (which stream at 6.4 TB/s)
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/${1:-r4o}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
pmc() {  # tag, tokens, layers, iters, counters...
  local tag=$1 tok=$2 lay=$3 it=$4; shift 4
  (cd /tmp && timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$OUT/pmc_$tag" -o m -- \
      python "$R/tools/prefill_once.py" mixtral_8x7b $tok $lay $it > /dev/null 2> "$R/$OUT/pmc_$tag.err")
  python tools/pmc_kernel_means.py "$OUT/pmc_$tag/m_counter_collection.csv" ffn > "$OUT/pmc_$tag.json" 2>> "$OUT/pmc_$tag.err"
  rm -rf "$OUT/pmc_$tag"
}
for w in "p512 512 2 6" "d1 1 8 24"; do
  set -- $w
  pmc ${1}_ta $2 $3 $4 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE
  pmc ${1}_tcp1 $2 $3 $4 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
  pmc ${1}_tcp2 $2 $3 $4 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum
  pmc ${1}_tcc $2 $3 $4 TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_CYCLE_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
  pmc ${1}_tlb $2 $3 $4 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum
  pmc ${1}_td $2 $3 $4 TD_TD_BUSY_sum TD_TC_STALL_sum TD_SPI_STALL_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
done
for f in "$OUT"/pmc_*.json; do echo "== $f"; cat "$f"; done
tail -2 "$OUT"/pmc_*.err | grep -i "error\|invalid\|fail" | head
