#!/bin/bash
# new tests (engine-side predictor, reference-caller trace) + the prefetch study with the engine-side predictor
set -u
OUT=gpurun_out/${1:-r3c}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 600 python -m pytest tests/test_gpu_interface.py tests/test_gpu_dropin.py -m gpu -q -rf > "$OUT/pytest.log" 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|^E  " "$OUT/pytest.log" | tail -12
POL="lfu,lfu+prefetch+governor,lfu+engine_predictor_la1,lfu+engine_predictor_la2+governor,lfu+engine_predictor_la8+governor"
timeout 700 python tools/prefetch_study.py --workload mixtral_8x7b --layers 8 --policies "$POL" > "$OUT/prefetch_study_mixtral_l8.jsonl" 2> "$OUT/ps_mixtral.err" || tail -5 "$OUT/ps_mixtral.err"
cut -c1-520 "$OUT/prefetch_study_mixtral_l8.jsonl"
timeout 500 python tools/prefetch_study.py --workload deepseek_v2_lite --layers 26 --cache-frac 0.25 --policies "$POL" > "$OUT/prefetch_study_deepseek.jsonl" 2> "$OUT/ps_deepseek.err" || tail -5 "$OUT/ps_deepseek.err"
cut -c1-520 "$OUT/prefetch_study_deepseek.jsonl"
