#!/bin/bash
# kernel-level breakdown of one prefill-sized layer (rocprofv3 kernel trace)
set -u
export TMPDIR=/tmp
R=$(pwd)
OUT=gpurun_out/${1:-r3ae}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
for spec in "deepseek_v2_lite 4096" "mixtral_8x7b 4096" "nllb_moe_54b 2048"; do
  set -- $spec
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_$1" -o m -- python "$R/tools/prefill_once.py" $1 $2 2 12 > /dev/null 2> "$R/$OUT/kt_$1.err")
  echo "== $1 $2 tokens"; python tools/rocprof_summary.py "$OUT/kt_$1/m_kernel_stats.csv" "$OUT/kernel_stats_$1.csv" > /dev/null; grep -v "at::native\|copyBuffer\|retile\|fillBuffer" "$OUT/kernel_stats_$1.csv" | head -14
  rm -rf "$OUT/kt_$1"
done
