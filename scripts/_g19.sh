mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_att_bwd_x3_gpu.py tests/test_step_gpu.py tests/test_kernels_gpu.py tests/test_bf16_gpu.py -q > gpurun_out/r5/x3tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r5/x3tests.log
timeout 300 python scripts/bench_att_bwd.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/bench_att_bwd7.txt; cat gpurun_out/r5/bench_att_bwd7.txt
bash scripts/collect_pmc_att_x3.sh r05
bash scripts/prof_step.sh r05l_fp32
grep "att_out_fwd\|dy1_stats" gpurun_out/r05l_fp32_timeline.txt
