#!/bin/bash
# full pytest -m gpu (all failures reported, no -x), log under gpurun_out/<tag>/
set -u
OUT=gpurun_out/${1:-pytest}
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1500 python -m pytest tests -m gpu -q -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_gpu.log" | tail -40
