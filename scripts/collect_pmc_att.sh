#!/bin/bash
# SQ counters of the exact-mode attention kernels in isolation (scripts/bench_att_fp32.py), one counter set per pass
# (no trace domains besides --kernel-trace).   usage on the GPU box:  bash scripts/collect_pmc_att.sh <tag>
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
re="att_l0_fwd|att_l0_bwd|att_l1_bwd|dw_multi|hgemm_l0g"
i=0
files=""
for set in "SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf /tmp/pa_$i
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$re" --output-format csv -d /tmp/pa_$i -o p -- python $root/scripts/bench_att_fp32.py > /tmp/pa_$i.log 2>&1
  f=$(find /tmp/pa_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && files="$files $f" || tail -5 /tmp/pa_$i.log
done
python $root/scripts/pmc_table.py $files > $out/${tag}_att_pmc.md
cat $out/${tag}_att_pmc.md
