set -u; export TMPDIR=/tmp; R=$(pwd); OUT=gpurun_out/r3ai; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt" -o m -- python "$R/tools/prefill_once.py" switch_base_8 512 2 12 > /dev/null 2> "$R/$OUT/kt.err")
python tools/rocprof_summary.py "$OUT/kt/m_kernel_stats.csv" "$OUT/kernel_stats_switch.csv" > /dev/null; grep -v "at::native\|copyBuffer\|retile\|fillBuffer" "$OUT/kernel_stats_switch.csv" | head -14; rm -rf $OUT/kt
