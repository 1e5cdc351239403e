bash tools/gpu_run.sh r6u build pytest smoke bench
