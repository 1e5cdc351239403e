#!/bin/bash
# round 5, closing run at HEAD: full pytest -m gpu, smoke(), the default bench line (as the driver runs it), the two-rank line on one GPU
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5final}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1200 python -m pytest tests -m gpu -q -rf > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_gpu.log" | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
bash tools/gpu_r5z.sh "${1:-r5final}"
