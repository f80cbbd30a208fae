# end-of-round evidence run: fuzz, bench line, step timelines (fp32 + bf16), kernel stats
python scripts/fuzz_step.py 12 41 > gpurun_out/r3_round_fuzz.log 2>&1
FUZZ_LONG=1 python scripts/fuzz_step.py 8 13 sli_rec,a2svd >> gpurun_out/r3_round_fuzz.log 2>&1
python bench.py > gpurun_out/r3_round_bench.json 2> gpurun_out/r3_round_bench.err
bash scripts/prof_step.sh r03_fp32 > /dev/null 2>&1
bash scripts/prof_step.sh r03_bf16 --precision bf16 > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
grep -c "^ok" gpurun_out/r3_round_fuzz.log; grep "FAIL\|problems" gpurun_out/r3_round_fuzz.log | head; tail -2 gpurun_out/r03_fp32_timeline.txt; tail -2 gpurun_out/r03_bf16_timeline.txt
grep '^{"metric' gpurun_out/r3_round_bench.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [(e["workload"][:32], e["ms_per_step"]) for e in d["extra_workloads"]], d.get("precision_modes", {}).get("bf16", {}).get("ms_per_step"))'
