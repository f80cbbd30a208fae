#!/bin/bash
# rocprofv3 kernel trace (>= 22 dispatches per kernel) + HBM traffic counters (separate --pmc passes: FETCH_SIZE needs 3
# TCC slots, WRITE_SIZE 2) of the HBM-bound embedding kernels on the 100M-item catalogue:
# -> gpurun_out/<tag>_embed_kernel_trace.md        usage on the GPU box: bash scripts/prof_embed.sh r04
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
re='ss_chunks|ss_borders|table_adam_rows|gather_hist_fwd_h'
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pe /tmp/pe_F /tmp/pe_W
timeout 500 rocprofv3 --kernel-trace --kernel-include-regex "$re" --output-format csv -d /tmp/pe -o e -- python $root/scripts/prof_kernels.py embed > /tmp/pe.log 2>&1
cp /tmp/pe.log $root/gpurun_out/${tag}_embed_isolated.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$re" --output-format csv -d /tmp/pe_${c:0:1} -o e -- python $root/scripts/prof_kernels.py embed > /tmp/pe_$c.log 2>&1
done
python - $(find /tmp/pe -name "*kernel_trace.csv" | head -1) $(find /tmp/pe_F -name "*counter_collection.csv" | head -1) $(find /tmp/pe_W -name "*counter_collection.csv" | head -1) > $root/gpurun_out/${tag}_embed_kernel_trace.md <<'PY'
import csv, re, sys
from collections import OrderedDict
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n
trace = list(csv.DictReader(open(sys.argv[1])))
dur = OrderedDict()
for r in trace:
    dur.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pmc = {}
for path, cname in ((sys.argv[2], "FETCH_SIZE"), (sys.argv[3], "WRITE_SIZE")):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == cname:
            pmc.setdefault((short(r["Kernel_Name"]), cname), []).append(float(r["Counter_Value"]))
print("# HBM-bound embedding kernels on the 100M-item catalogue, alone (rocprofv3 --kernel-trace; FETCH_SIZE / WRITE_SIZE from\n"
      "# separate --pmc passes, KB per dispatch; traffic = WRITE_SIZE + 2 x FETCH_SIZE, the gfx950 wide-load correction)\n")
print("| kernel | dispatches | avg us | median us | min us | max us | FETCH_SIZE KB | WRITE_SIZE KB | traffic MB |")
print("|---|---|---|---|---|---|---|---|---|")
for k, d in dur.items():
    d2 = sorted(d)
    f = pmc.get((k, "FETCH_SIZE"), [0.0]); w = pmc.get((k, "WRITE_SIZE"), [0.0])
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    print("| `%s` | %d | %.2f | %.2f | %.2f | %.2f | %.0f | %.0f | %.1f |" % (k[:70], len(d), sum(d) / len(d), d2[len(d2) // 2], d2[0], d2[-1], fa, wa, (wa + 2 * fa) / 1e3))
PY
cat $root/gpurun_out/${tag}_embed_kernel_trace.md; cat $root/gpurun_out/${tag}_embed_isolated.txt | tail -8
