#!/bin/bash
# round 5: the opt-in one-launch layer's own test + the per-round profiles (rocprofv3 kernel stats, PMC traffic, MFMA busy)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5j}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -k "one_launch" > "$OUT/pytest_one_launch.log" 2>&1; echo "pytest exit $?"; tail -5 "$OUT/pytest_one_launch.log"
timeout 1500 bash tools/run_profiles.sh "$OUT/profiles" > "$OUT/run_profiles.log" 2>&1; echo "profiles exit $?"; tail -15 "$OUT/run_profiles.log"
