#!/bin/bash
# re-run of tools/prefetch_study.py with the round-2 priority queue (R8): on-demand vs speculative prefetch, Mixtral L=8 and DeepSeek
set -u
OUT=gpurun_out/${1:-pstudy}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 500 python tools/prefetch_study.py --workload mixtral_8x7b --layers 8 > "$OUT/prefetch_study_mixtral_l8.jsonl" 2> "$OUT/ps_mixtral.err" || tail -5 "$OUT/ps_mixtral.err"
cut -c1-420 "$OUT/prefetch_study_mixtral_l8.jsonl"
timeout 300 python tools/prefetch_study.py --workload deepseek_v2_lite --layers 26 --cache-frac 0.25 > "$OUT/prefetch_study_deepseek.jsonl" 2> "$OUT/ps_deepseek.err" || tail -5 "$OUT/ps_deepseek.err"
cut -c1-420 "$OUT/prefetch_study_deepseek.jsonl"
