#!/bin/bash
# Standalone rocprofv3 kernel trace of the history gather on the 100M-item catalogue (the kernel behind bench.py's
# `roofline`): >= 20 dispatches, avg / min / max -> gpurun_out/<tag>_gather_hist_fwd_kernel_trace.md
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pg
rocprofv3 --kernel-trace --stats --kernel-include-regex gather_hist_fwd --output-format csv -d /tmp/pg -o g -- python $root/scripts/prof_kernels.py gather > /tmp/pg.log 2>&1
tail -3 /tmp/pg.log
f=$(find /tmp/pg -name "*kernel_trace.csv" | head -1)
python - "$f" > $root/gpurun_out/${tag}_gather_hist_fwd_kernel_trace.md <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gather_hist_fwd" in r["Kernel_Name"] and ("<13>" in r["Kernel_Name"] or "ILi13E" in r["Kernel_Name"])]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
b = 51600 * 4096
print("# gather_hist_fwd_kernel alone on the 100M-item catalogue (rocprofv3 --kernel-trace, scripts/prof_gather.sh)\n")
print("4096 histories x 50 steps, rows 384 B + 128 B, uniform ids over 100M items (38 GB table); algorithmic bytes per launch")
print("= 51 600 B x 4096 = %.2f MB (SURVEY 8d)\n" % (b / 1e6))
print("| dispatches | avg us | median us | min us | max us | GB/s at avg | fraction of 8 TB/s |")
print("|---|---|---|---|---|---|---|")
avg = sum(d) / len(d)
print("| %d | %.2f | %.2f | %.2f | %.2f | %.0f | %.3f |" % (len(d), avg, d[len(d) // 2], d[0], d[-1], b / avg / 1e3, b / avg / 1e3 / 8000))
PY
cat $root/gpurun_out/${tag}_gather_hist_fwd_kernel_trace.md
