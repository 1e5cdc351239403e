#!/bin/bash
# run selected GPU tests: tools/gpu_pytest_sel.sh <tag> <pytest args...>
set -u
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 1500 python -m pytest "$@" -q -rf > "$OUT/pytest_sel.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_sel.log"
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit|^E  " "$OUT/pytest_sel.log" | tail -60
