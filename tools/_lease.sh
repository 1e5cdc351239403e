timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_interface.py -q -k "v3 or grok or deepseek_golden or mixtral_golden or interface" 2>&1 | tail -8
