mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_att_chain_bf16_gpu.py tests/test_bf16_gpu.py tests/test_att_bwd_x3_gpu.py -x -q > gpurun_out/r5/h3_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r5/h3_tests.log
python scripts/bench_att_chain_h.py 2>&1 | tee gpurun_out/r5/h3_chain_h.txt
bash scripts/collect_pmc_catalogue.sh r05 > gpurun_out/r5/h3_pmc.log 2>&1; tail -30 gpurun_out/r5/h3_pmc.log
