#!/bin/bash
# Round 6, first call: what does the strict-fp32 step cost today, site by site; is it still green.
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 30 --warmup 5"
run() { tag=$1; shift; env "$@" $B 2> gpurun_out/r06a_$tag.err | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"])' $tag; }
run default X=1
run exact CLSR_EXACT_PRODUCTS=1
run rnn_fp32 CLSR_RNN_PRODUCTS=fp32
run attbwd_fp32 CLSR_ATT_BWD=fp32
run no_x3_enc CLSR_NO_X3_ENC=1
run no_enc_back_x3 CLSR_NO_ENC_BACK_X3=1
run no_l1fwd_x6 CLSR_NO_ATT_L1_FWD_X6=1
run no_hist_bwd_x3 CLSR_NO_ATT_HIST_BWD_X3=1
run no_hist_x3 CLSR_NO_ATT_HIST_X3=1
run fwd_x6 CLSR_ATT_FWD_X6=1
run exact_again CLSR_EXACT_PRODUCTS=1
CLSR_EXACT_PRODUCTS=1 timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > gpurun_out/r06a_exact_tests.log 2>&1; echo "exact tests rc $?"; tail -3 gpurun_out/r06a_exact_tests.log
