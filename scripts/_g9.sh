mkdir -p gpurun_out/r5
timeout 1200 python bench.py > gpurun_out/r5/bench1.json 2> gpurun_out/r5/bench1.err; echo "bench rc $?"; tail -3 gpurun_out/r5/bench1.err
grep '^{"metric' gpurun_out/r5/bench1.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [(e["workload"][:40], e["ms_per_step"]) for e in d["extra_workloads"]], {k:v.get("ms_per_step") for k,v in d.get("precision_modes", {}).items()}, d.get("roofline_att_bwd",{}).get("frac"), d.get("roofline_mfma",{}).get("frac"), {k:(v.get("frac"),v.get("us_per_launch")) for k,v in d["roofline"].get("more",{}).items()})'
timeout 600 python scripts/fit_throughput.py 262144 2>&1 | tail -3
