#!/bin/bash
# round-2 GPU call B (state check after re-entry): full pytest -m gpu (no -x, all failures listed) + the default bench line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02b
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1100 python -m pytest tests -m gpu -q -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_gpu.log" | tail -30
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench exit $?"; tail -8 "$OUT/bench_default.err"; head -c 6000 "$OUT/bench_default.json"
