bash tools/gpu_run.sh r6f pytest:"dropin or interface or chained"
for w in deepseek-v2-lite mixtral-8x7b; do timeout 1200 python tools/dropin_time.py --workload $w --layers 8 > gpurun_out/r6f/dropin_$w.json 2> gpurun_out/r6f/dropin_$w.err; python -c "
import json,sys; d=json.load(open('gpurun_out/r6f/dropin_$w.json')); print({k:v for k,v in d.items() if k not in ('what','host_calls')}); [print('   ',k,v) for k,v in d['host_calls'].items()]"; done
SWEEP_ENVS="A=1;MOEINF_GEMM_BIG_MODE=3;A=2;MOEINF_GEMM_BIG_MODE=3" timeout 900 python tools/ffn_sweep.py mixtral_8x7b:4096:2 mixtral_8x7b:3840:2 mixtral_8x7b:2048:2 2>&1 | tee gpurun_out/r6f/big_mode3.txt
MOEINF_GEMM_BIG_MODE=3 timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "mixtral_8x7b_layer or deepseek_v2_lite_layer or skewed or prefill" 2>&1 | tail -5
timeout 900 python tools/decode_ab.py deepseek-v2-lite base MOEINF_SR_LDS_KB=30 MOEINF_SR_LDS_KB=45 MOEINF_SR_LDS_KB=70 MOEINF_SR_LDS_KB=45,MOEINF_SR_U=4 MOEINF_SR_LDS_KB=70,MOEINF_SR_U=4 MOEINF_SR_U=4 2>&1 | tee gpurun_out/r6f/deepseek_front1_occupancy.txt
