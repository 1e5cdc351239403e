#!/bin/bash
# round 4, call A: design probes — (1) which memory kinds hipIpc* can share between processes + hand-off coherence and
set -u
OUT=gpurun_out/r4a; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/ipc_probe tools/ipc_probe.hip 2> "$OUT/build.err" || { cat "$OUT/build.err"; }
timeout 300 /tmp/ipc_probe 2>&1 | tee "$OUT/ipc_probe.txt"
