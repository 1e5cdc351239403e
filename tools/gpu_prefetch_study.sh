#!/bin/bash
# tools/prefetch_study.py on the GPU box: on-demand vs speculative prefetch (with and without the governor)
set -u
OUT=gpurun_out/${1:-pstudy}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 300 python -m pytest tests/test_gpu_tiers.py tests/test_gpu_parity.py -m gpu -q -x -k "tiers or prefetch or eviction" > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -8
timeout 600 python tools/prefetch_study.py --workload mixtral_8x7b --layers 8 > "$OUT/prefetch_study_mixtral_l8.jsonl" 2> "$OUT/ps_mixtral.err" || tail -5 "$OUT/ps_mixtral.err"
cut -c1-460 "$OUT/prefetch_study_mixtral_l8.jsonl"
timeout 400 python tools/prefetch_study.py --workload deepseek_v2_lite --layers 26 --cache-frac 0.25 > "$OUT/prefetch_study_deepseek.jsonl" 2> "$OUT/ps_deepseek.err" || tail -5 "$OUT/ps_deepseek.err"
cut -c1-460 "$OUT/prefetch_study_deepseek.jsonl"
