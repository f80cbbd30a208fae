#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_det_grads_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "segment or det or sort or long_runs or stored_once or row_level or reproducible" > gpurun_out/r06d_tests.log 2>&1; echo "kernel tests rc $?"; tail -3 gpurun_out/r06d_tests.log
timeout 900 python bench.py --no-cpu-baseline --no-extra --steps 20 > gpurun_out/r06d_bench.json 2> gpurun_out/r06d_bench.err; echo "bench rc $?"; grep "roofline gather_bwd\|timed" gpurun_out/r06d_bench.err
