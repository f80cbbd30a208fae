#!/bin/bash
# full GPU suite + bench line + timelines (final-state evidence); usage: bash scripts/r06_full.sh <tag> [quick]
tag=${1:-r06x}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/${tag}_tests.log
if [ "$2" != "quick" ]; then
python scripts/fuzz_step.py 12 41 > gpurun_out/${tag}_fuzz.log 2>&1; grep -c "^ok" gpurun_out/${tag}_fuzz.log; grep "FAIL\|problems" gpurun_out/${tag}_fuzz.log | head -3
FUZZ_PRECISION=fp32x3 python scripts/fuzz_step.py 12 41 > gpurun_out/${tag}_fuzz_x3.log 2>&1; grep -c "^ok" gpurun_out/${tag}_fuzz_x3.log
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench.err
grep '^{"metric' gpurun_out/${tag}_bench.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["more"]["gather_bwd"]["frac"], [(e["workload"][:32], e["ms_per_step"]) for e in d["extra_workloads"]], {k: v.get("ms_per_step") for k, v in d.get("precision_modes", {}).items()})'
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash scripts/prof_step.sh ${tag}_fp32 | tail -1
bash scripts/prof_step.sh ${tag}_fp32x3 --precision fp32x3 | tail -1
bash scripts/prof_step.sh ${tag}_bf16 --precision bf16 | tail -1
bash scripts/prof_step.sh ${tag}_kuaishou --config kuaishou | tail -1
bash scripts/prof_step.sh ${tag}_cat --config catalogue100m --steps 8 | tail -1
fi
