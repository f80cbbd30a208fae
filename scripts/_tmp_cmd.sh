python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_siblings_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -2
bash scripts/prof_step.sh r3g > /dev/null 2>&1
awk '$1>1100 && $1<1480' gpurun_out/r3g_timeline.txt | cut -c1-110
