mkdir -p gpurun_out/r5
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r5/gputests_final.log 2>&1; echo "all gpu tests rc $?"; tail -4 gpurun_out/r5/gputests_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py > gpurun_out/r5/bench2.json 2> gpurun_out/r5/bench2.err; echo "bench rc $?"
grep '^{"metric' gpurun_out/r5/bench2.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [(e["workload"][:40], e["ms_per_step"]) for e in d["extra_workloads"]], {k:v.get("ms_per_step") for k,v in d.get("precision_modes", {}).items()}, d.get("roofline_att_bwd",{}).get("frac"), d.get("roofline_mfma",{}).get("frac"), {k:(v.get("frac"),v.get("us_per_launch")) for k,v in d["roofline"].get("more",{}).items()})'
bash scripts/prof_step.sh r05_fp32
bash scripts/prof_step.sh r05_bf16 --precision bf16
bash scripts/prof_step.sh r05_cat --config catalogue100m
bash scripts/prof_step.sh r05_kuaishou --config kuaishou
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_stats && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 20 > /tmp/prof_stats.log 2>&1; cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r05_bench_kernel_stats.csv; head -5 gpurun_out/r05_bench_kernel_stats.csv
