mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5/h10_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r5/h10_tests.log
python scripts/fuzz_step.py 12 41 > gpurun_out/r5/h10_fuzz.log 2>&1; grep -c "^ok" gpurun_out/r5/h10_fuzz.log; grep "FAIL\|problems" gpurun_out/r5/h10_fuzz.log | head
python bench.py > gpurun_out/r5/h10_bench.json 2> gpurun_out/r5/h10_bench.err; tail -2 gpurun_out/r5/h10_bench.err
grep '^{"metric' gpurun_out/r5/h10_bench.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [(e["workload"][:32], e["ms_per_step"]) for e in d["extra_workloads"]], d.get("precision_modes", {}).get("bf16", {}).get("ms_per_step"))'
bash scripts/prof_step.sh r05c_kuaishou --config kuaishou
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
