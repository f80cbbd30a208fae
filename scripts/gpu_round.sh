# end-of-round evidence run: full GPU suite, fuzz, bench line, step timelines (fp32 + bf16)
python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r3_round_tests.log
python scripts/fuzz_step.py 12 31 > gpurun_out/r3_round_fuzz.log 2>&1
FUZZ_LONG=1 python scripts/fuzz_step.py 8 12 sli_rec,a2svd >> gpurun_out/r3_round_fuzz.log 2>&1
python bench.py > gpurun_out/r3_round_bench.json 2> gpurun_out/r3_round_bench.err
bash scripts/prof_step.sh r03_fp32 > /dev/null 2>&1
bash scripts/prof_step.sh r03_bf16 --precision bf16 > /dev/null 2>&1
tail -3 gpurun_out/r3_round_tests.log; grep -c "^ok" gpurun_out/r3_round_fuzz.log; grep "FAIL\|problems" gpurun_out/r3_round_fuzz.log | head; tail -c 1500 gpurun_out/r3_round_bench.json; tail -2 gpurun_out/r03_fp32_timeline.txt; tail -2 gpurun_out/r03_bf16_timeline.txt
