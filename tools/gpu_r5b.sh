#!/bin/bash
# round 5: the N>1 work — bench.py spawning its own ranks, agreement / liveness of the exchange, multi-device prefetch_op
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r5b}; mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_dropin.py tests/test_gpu_ep_peer.py tests/test_gpu_ep_processes.py -q -rf -x > "$OUT/pytest.log" 2>&1; echo "pytest exit $?"
tail -40 "$OUT/pytest.log"
