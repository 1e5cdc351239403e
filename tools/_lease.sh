# round 6, lease j: the shipped rule for short last passes (A/B), new tests, then the default bench line with its new legs
mkdir -p gpurun_out/r6j
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -k "short_last_passes or grok or (mixtral_8x7b_layer and 2048)" 2>&1 | tail -4
for t in 4096 3840 3072 2048 1536; do
SWEEP_ENVS="MOEINF_GEMM_BIG_MOVE=0;MOEINF_GEMM_BIG_MOVE=1;MOEINF_GEMM_BIG_MOVE=0;MOEINF_GEMM_BIG_MOVE=1" timeout 600 python tools/ffn_sweep.py mixtral_8x7b:$t:2 2>&1 | tee -a gpurun_out/r6j/big_move_rule.txt
done
bash tools/gpu_run.sh r6j bench pytest:bench_ranks
