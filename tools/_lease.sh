# round 6, lease g: the full GPU suite at HEAD, then the profiles that had none (NLLB batch 32, Switch, fp16), DeepSeek at HEAD, the prefill GEMM's MFMA busy + clock + L2
bash tools/gpu_run.sh r6g build pytest smoke
bash tools/gpu_run.sh r6g kt:nllb-moe-54b:32 pmc:nllb-moe-54b:FETCH_SIZE:32 pmc:nllb-moe-54b:WRITE_SIZE:32
bash tools/gpu_run.sh r6g kt:switch-base-8 pmc:switch-base-8:FETCH_SIZE pmc:switch-base-8:WRITE_SIZE
bash tools/gpu_run.sh r6g kt:deepseek-v2-lite pmc:deepseek-v2-lite:FETCH_SIZE pmc:deepseek-v2-lite:WRITE_SIZE
GPU_RUN_BENCH_FLAGS="--dtype fp16" bash tools/gpu_run.sh r6g_fp16 kt:mixtral-8x7b kt:deepseek-v2-lite kt:nllb-moe-54b:32
bash tools/gpu_run.sh r6g mfma:mixtral-8x7b:4096
GPU_RUN_PROMPT=4096 bash tools/gpu_run.sh r6g pmc:mixtral-8x7b:TCC_HIT_sum+TCC_MISS_sum
