#!/bin/bash
# Counters of the round-5 attention kernels in isolation (scripts/bench_att_bwd.py): HBM traffic (FETCH_SIZE / WRITE_SIZE, one
# counter per pass) and SQ activity (one set per pass); no trace domains besides --kernel-trace.
#   usage on the GPU box:  bash scripts/collect_pmc_att_x3.sh <tag>     -> gpurun_out/<tag>_att_x3_pmc.md
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
re="att_l1_bwd_x3|att_l0_bwd_x3|att_hist_fwd_x3|att_hist_bwd_x3|att_l1_fwd_kernel|proj_x3"
i=0
files=""
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/px_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$re" --output-format csv -d /tmp/px_$i -o p -- env ONLY_ST=1 python $root/scripts/bench_att_bwd.py > /tmp/px_$i.log 2>&1
  f=$(find /tmp/px_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && files="$files $f" || tail -5 /tmp/px_$i.log
done
python $root/scripts/pmc_table.py $files > $out/${tag}_att_x3_pmc.md
cat $out/${tag}_att_x3_pmc.md
