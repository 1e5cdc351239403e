mkdir -p gpurun_out/r6t
timeout 900 python -m pytest tests/test_gpu_tiers.py tests/test_gpu_dropin.py tests/test_gpu_chained.py -q -x 2>&1 | tail -4
OFFLOAD_AB_FLAGS="--offload-attn-us 0 --offload-zipf-steps 1" bash tools/offload_ab.sh gpurun_out/r6t deepseek-v2-lite MOEINF_H2D_PULL=0 base | tee gpurun_out/r6t/pull_ab_deepseek.txt
OFFLOAD_AB_FLAGS="--layers 16" bash tools/offload_ab.sh gpurun_out/r6t mixtral-8x7b MOEINF_H2D_PULL=0 base | tee gpurun_out/r6t/pull_ab_mixtral.txt
