#!/bin/bash
# One parametrised runner for the GPU box (replaces the per-lease tools/gpu_r*.sh scripts of rounds 3-5; the command
# lines that produced each file under profiles/ are listed in profiles/README.md).
#
#   gpurun --timeout S -- 'bash tools/gpu_run.sh <tag> <task> [<task> ...]'
#
# Everything is written under gpurun_out/<tag>/ (merged back by gpurun).  Tasks, run in the order given:
#   build                 __graft_entry__.build()
#   pytest[:<expr>]       python -m pytest tests -m gpu [-k <expr>]
#   smoke                 __graft_entry__.smoke()
#   bench                 the driver's command: python bench.py --gpus 1 --steps 20 --warmup 5
#   bench2                two ranks on this one GPU over the peer-store transport (8 layers)
#   kt:<workload>[:B]     rocprofv3 --kernel-trace --stats of a lean bench of <workload> (decode batch B)
#   pmc:<workload>:<C>[:B] rocprofv3 --pmc <C> --kernel-trace (C = one counter set, '+'-separated) over 8 layers
#   mfma:<workload>:<P>   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over a P-token prefill (8 layers): busy fraction + clock
#   py:<script>[:args]    python tools/<script> with ':'-separated args, output to <script>.log
# Every profiler / python invocation is wrapped in `timeout`; PMC passes are their own runs (kernel-trace only).
set -u
export TMPDIR=/tmp
R=$(pwd)
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
LEAN="--no-cpu-baseline --no-other-configs --miss-heavy-frac 0 --windows 1 --no-traffic"
for task in "$@"; do
  IFS=':' read -r kind a1 a2 a3 a4 <<< "$task"
  case "$kind" in
    build)
      python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1; echo "build exit $?" ;;
    pytest)
      if [ -n "${a1:-}" ]; then timeout 1500 python -m pytest tests -m gpu -q -rf -k "$a1" > "$OUT/pytest_gpu.log" 2>&1
      else timeout 1500 python -m pytest tests -m gpu -q -rf > "$OUT/pytest_gpu.log" 2>&1; fi
      echo "pytest exit $?" >> "$OUT/pytest_gpu.log"
      grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" "$OUT/pytest_gpu.log" | tail -20 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log" ;;
    bench)
      ( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"
      echo "bench exit $? line bytes $(wc -c < "$OUT/bench_default.json")"; tail -3 "$OUT/bench_default.time"
      cp bench_details.json "$OUT/bench_default_details.json" 2>/dev/null
      grep -E "live traffic|failed|FAILED" "$OUT/bench_default.err" | tail -6 ;;
    bench2)
      MOEINF_BENCH_SHARE_GPU0=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-other-configs \
        --layers 8 --cpu-sample-layers 4 --cpu-sample-steps 2 > "$OUT/bench_two_ranks_one_gpu.json" 2> "$OUT/bench_two_ranks_one_gpu.err"
      echo "two-rank bench exit $?"; cp bench_details.json "$OUT/bench_two_ranks_details.json" 2>/dev/null ;;
    kt)
      tag=${a1//-/}; tag=${tag//./}; b=${a2:-1}
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/kt_${tag}_b$b" -o m -- \
          python "$R/bench.py" --workload "$a1" --batch "$b" --steps 10 --warmup 2 $LEAN ${GPU_RUN_BENCH_FLAGS:-} > "$R/$OUT/kt_bench_${tag}_b$b.json" 2> "$R/$OUT/kt_${tag}_b$b.err")
      python tools/rocprof_summary.py "$OUT/kt_${tag}_b$b/m_kernel_stats.csv" "$OUT/kernel_stats_${tag}_b$b.csv" | head -12
      rm -f "$OUT"/kt_*/*kernel_trace.csv ;;
    pmc)
      tag=${a1//-/}; tag=${tag//./}; b=${a3:-1}; ctr=${a2//+/ }
      (cd /tmp && timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/$OUT/pmc_${a2}_${tag}_b$b" -o m -- \
          python "$R/bench.py" --workload "$a1" --batch "$b" --steps 3 --warmup 1 $LEAN --layers 8 --prompt ${GPU_RUN_PROMPT:-0} ${GPU_RUN_BENCH_FLAGS:-} > /dev/null 2> "$R/$OUT/pmc_${a2}_${tag}_b$b.err")
      python tools/pmc_kernel_means.py "$OUT/pmc_${a2}_${tag}_b$b/m_counter_collection.csv" > "$OUT/pmc_${a2}_${tag}_b$b.txt" 2>&1; head -20 "$OUT/pmc_${a2}_${tag}_b$b.txt"
      rm -f "$OUT"/pmc_*/m_kernel_trace.csv "$OUT"/pmc_*/m_counter_collection.csv ;;
    mfma)  # mfma:<workload>:<prompt tokens>: MFMA-pipe busy cycles + GUI-active cycles (= the clock the chip held) of the prefill GEMMs
      tag=${a1//-/}; tag=${tag//./}
      (cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_mfma_${tag}_p$a2" -o m -- \
          python "$R/bench.py" --workload "$a1" --steps 2 --warmup 1 $LEAN --layers 8 --prompt "$a2" > /dev/null 2> "$R/$OUT/pmc_mfma_${tag}_p$a2.err")
      python tools/mfma_summary.py "$OUT/pmc_mfma_${tag}_p$a2/m_counter_collection.csv" "$OUT/pmc_mfma_${tag}_p$a2/m_kernel_trace.csv" "$OUT/pmc_mfma_prefill${a2}_${tag}.json" > /dev/null 2> "$OUT/mfma_summary.err"
      python - "$OUT/pmc_mfma_prefill${a2}_${tag}.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["kernels"].items():
    if v["duration_ns"] > 50000:
        print(k[:64], "us %.1f" % (v["duration_ns"] / 1e3), "busy %.3f" % v["mfma_busy_frac_by_gui_active"], "GHz %.2f" % (v["GRBM_GUI_ACTIVE"] / v["duration_ns"] / 8), "launches", v["launches"])
PY
      rm -f "$OUT"/pmc_mfma_*/m_kernel_trace.csv "$OUT"/pmc_mfma_*/m_counter_collection.csv ;;
    py)
      args=(); [ -n "${a2:-}" ] && IFS=',' read -r -a args <<< "$a2"
      timeout ${GPU_RUN_PY_TIMEOUT:-900} python "tools/$a1" "${args[@]}" > "$OUT/${a1%.py}.log" 2>&1; echo "$a1 exit $?"; tail -${GPU_RUN_TAIL:-30} "$OUT/${a1%.py}.log" ;;
    *) echo "unknown task $task"; exit 2 ;;
  esac
done
ls "$OUT" | head -50
