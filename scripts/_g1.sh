mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputests.log 2>&1; echo "pytest rc $?" 
tail -3 gpurun_out/r5/gputests.log
timeout 900 python bench.py > gpurun_out/r5/bench0.json 2> gpurun_out/r5/bench0.err; echo "bench rc $?"
bash scripts/prof_step.sh r05a_fp32
bash scripts/prof_step.sh r05a_bf16 --precision bf16
