#!/bin/bash
# Round 6: the segmented sums as ONE launch -- kernel tests, step tests, the catalogue rooflines
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_det_grads_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "segment or det or sort or long_runs or stored_once or row_level or reproducible" > gpurun_out/r06c_tests.log 2>&1; echo "kernel tests rc $?"; tail -3 gpurun_out/r06c_tests.log
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_bf16_tables_gpu.py tests/test_siblings_gpu.py -x -q -m gpu > gpurun_out/r06c_step.log 2>&1; echo "step tests rc $?"; tail -2 gpurun_out/r06c_step.log
timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 20 > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err; echo "bench rc $?"; grep "roofline\|timed" gpurun_out/r06c_bench.err
