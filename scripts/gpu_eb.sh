python -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py -q -x 2>&1 | tail -2
bash scripts/r3_ab.sh CLSR_NO_EARLY_SCATTER "" 1
bash scripts/prof_step.sh r3eb > /dev/null 2>&1
sed -n '/rnn_multi_bwd/,$p' gpurun_out/r3eb_timeline.txt | cut -c1-110
