mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5/h1_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r5/h1_tests.log
python bench.py > gpurun_out/r5/h1_bench.json 2> gpurun_out/r5/h1_bench.err; tail -2 gpurun_out/r5/h1_bench.err
grep '^{"metric' gpurun_out/r5/h1_bench.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [(e["workload"][:32], e["ms_per_step"]) for e in d["extra_workloads"]], d.get("precision_modes", {}).get("bf16", {}).get("ms_per_step"))'
