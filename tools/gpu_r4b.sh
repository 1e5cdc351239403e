#!/bin/bash
# round 4, call B: the peer-store exchange on the GPU — in-process multi-rank tests, real processes over hipIpc, the
# world-size-1 module test over all transports, then forced-EP bench lines (peer-store vs rccl) for Mixtral and DeepSeek
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4b; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_ep_peer.py tests/test_gpu_ep_processes.py "tests/test_gpu_interface.py::test_expert_parallel_module_through_rccl_world_size_1" tests/test_gpu_parity.py -k "ep or peer or expert_parallel" -q -rf -x > "$OUT/pytest_ep.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_ep.log"; tail -25 "$OUT/pytest_ep.log"
for wl in mixtral-8x7b deepseek-v2-lite; do
  for tr in peer-store rccl; do
    timeout 300 python bench.py --workload $wl --force-ep --ep-transport $tr --no-other-configs --miss-heavy-frac 0 --prompt 0 --cpu-sample-layers 2 --cpu-sample-steps 2 > "$OUT/bench_ep1_${wl}_$tr.json" 2> "$OUT/bench_ep1_${wl}_$tr.err"
    echo "bench $wl $tr exit $?"; python - "$OUT/bench_ep1_${wl}_$tr.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d["ms_per_step"], d.get("ep_phases_us_per_layer"), d["ep_transport"]["chosen"], d.get("parity",{}).get("ok"))
except Exception as ex: print("no line", ex)
PY
    tail -2 "$OUT/bench_ep1_${wl}_$tr.err"
  done
done
MOEINF_EP_PEER_POLL=0 timeout 300 python bench.py --workload deepseek-v2-lite --force-ep --ep-transport peer-store --no-other-configs --miss-heavy-frac 0 --prompt 0 --no-cpu-baseline > "$OUT/bench_ep1_deepseek_peer_waitkernels.json" 2> "$OUT/bench_ep1_deepseek_peer_waitkernels.err"
python - "$OUT/bench_ep1_deepseek_peer_waitkernels.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("wait-kernel mode", d["ms_per_step"], d.get("ep_phases_us_per_layer"))
except Exception as ex: print("no line", ex)
PY
