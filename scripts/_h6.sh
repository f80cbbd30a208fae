mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_att_chain_bf16_gpu.py tests/test_bf16_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -4
bash scripts/prof_step.sh r05b_cat --config catalogue100m --steps 8
