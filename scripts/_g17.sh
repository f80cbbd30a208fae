mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q > gpurun_out/r5/fullsize.log 2>&1; echo "fullsize rc $?"; tail -3 gpurun_out/r5/fullsize.log
timeout 1200 python bench.py > gpurun_out/r5/bench2.json 2> gpurun_out/r5/bench2.err; echo "bench rc $?"; tail -2 gpurun_out/r5/bench2.err
grep '^{"metric' gpurun_out/r5/bench2.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [(e["workload"][:40], e["ms_per_step"]) for e in d["extra_workloads"]], {k:v.get("ms_per_step") for k,v in d.get("precision_modes", {}).items()}, d.get("roofline_att_bwd",{}).get("frac"), d.get("roofline_mfma",{}).get("frac"), {k:(v.get("frac"),v.get("us_per_launch")) for k,v in d["roofline"].get("more",{}).items()})'
