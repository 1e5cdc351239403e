#!/bin/bash
# prefill-size FFN timings (tools/ffn_sweep.py): per-stage us and algorithmic GB/s at 512 / 2048 / 4096 tokens
set -u
OUT=gpurun_out/${1:-prefill}; mkdir -p "$OUT"
python -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 600 python tools/ffn_sweep.py ${2:-mixtral_8x7b:512:2 mixtral_8x7b:2048:2 mixtral_8x7b:4096:2 deepseek_v2_lite:4096:4} 2>&1 | tee "$OUT/ffn_sweep_prefill.txt" | tail -30
