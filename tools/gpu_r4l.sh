#!/bin/bash
# is the 512-token prefill (a) memory time + matrix time back to back, and (b) power-limited?
#  - (the diagnostic builds of ring2 — no MFMAs / weight stream in cache / no activation DMA — were removed after the study;
#    their numbers: profiles/r04_ffn_sweep_ring2_prefill.txt)
#  - power / clocks sampled with rocm-smi while each of: 512-token prefill, batch-1 decode (pure streaming), 4096-token prefill
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4l}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
SWEEP_ENVS=";MOEINF_GEMM_RING2=0" timeout 400 python tools/ffn_sweep.py mixtral_8x7b:512:2 2>&1 | tee "$OUT/sweep.txt"
sample() {  # tag, seconds to sample, command...
  local tag=$1 n=$2; shift 2
  "$@" > "$OUT/work_$tag.log" 2>&1 &
  local pid=$!
  sleep 6   # engine set-up (weights generated on the GPU)
  for i in $(seq 1 $n); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk|socclk" | tr -s ' ' | tr '\n' '|' >> "$OUT/smi_$tag.txt"; echo >> "$OUT/smi_$tag.txt"
    kill -0 $pid 2>/dev/null || break
  done
  wait $pid
}
sample prefill512 12 timeout 120 python tools/prefill_once.py mixtral_8x7b 512 2 12000
sample prefill4096 12 timeout 120 python tools/prefill_once.py mixtral_8x7b 4096 2 1500
sample decode1 12 timeout 120 python tools/prefill_once.py mixtral_8x7b 1 8 80000
for t in prefill512 prefill4096 decode1; do echo "== $t"; head -12 "$OUT/smi_$t.txt" | cut -c1-400; done
