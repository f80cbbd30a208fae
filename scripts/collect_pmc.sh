#!/bin/bash
# SQ counters of ONE kernel family in isolation, one counter set per pass (no trace domains besides --kernel-trace).
#   usage on the GPU box:  bash scripts/collect_pmc.sh <tag> <kernel regex> <python script + args...>
tag=$1; re=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
files=""
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  rm -rf /tmp/pc_$i
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$re" --output-format csv -d /tmp/pc_$i -o p -- python $root/"$@" > /tmp/pc_$i.log 2>&1
  f=$(find /tmp/pc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && files="$files $f" || tail -5 /tmp/pc_$i.log
done
python $root/scripts/pmc_table.py $files > $out/${tag}_pmc.md
cat $out/${tag}_pmc.md
