# round 6, lease i: ffn_gemm_big — short last passes moved to the end of the grid and split (A/B by token count + parity)
mkdir -p gpurun_out/r6i
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "mixtral_8x7b_layer or short_last_passes or skewed or (deepseek_v2_lite_layer and 4096) or nllb_moe_54b_layer_prefill" 2>&1 | tail -6
for t in 4096 3072 3840 4224 2048 1536 2560; do
SWEEP_ENVS="MOEINF_GEMM_BIG_SPLIT=0;MOEINF_GEMM_BIG_SPLIT=1;MOEINF_GEMM_BIG_SPLIT=4;MOEINF_GEMM_BIG_SPLIT=0;MOEINF_GEMM_BIG_SPLIT=1;MOEINF_GEMM_BIG_SPLIT=4" timeout 600 python tools/ffn_sweep.py mixtral_8x7b:$t:2 2>&1 | tee -a gpurun_out/r6i/big_split.txt
done
