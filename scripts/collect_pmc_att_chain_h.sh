#!/bin/bash
# Counters of the speed-mode (bf16) chain kernels in isolation (scripts/bench_att_chain_h.py): HBM traffic (FETCH_SIZE /
# WRITE_SIZE, one counter per pass) and SQ activity; no trace domains besides --kernel-trace.
#   usage on the GPU box:  bash scripts/collect_pmc_att_chain_h.sh <tag>     -> gpurun_out/<tag>_att_chain_bf16_pmc.md
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
re="att_l1_bwd_x3|att_l0_bwd_x3|att_l0_fwd_kernel|att_l1_fwd_kernel"
i=0
files=""
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/ph_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$re" --output-format csv -d /tmp/ph_$i -o p -- python $root/scripts/bench_att_chain_h.py > /tmp/ph_$i.log 2>&1
  f=$(find /tmp/ph_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && files="$files $f" || tail -5 /tmp/ph_$i.log
done
python $root/scripts/pmc_table.py $files > $out/${tag}_att_chain_bf16_pmc.md
cat $out/${tag}_att_chain_bf16_pmc.md
