#!/bin/bash
# SQ counters of the forward recurrence launch in isolation (scripts/bench_rnn.py <cfg>), one counter set per pass (no
# trace domains besides --kernel-trace).   usage on the GPU box:  bash scripts/collect_pmc_rnn.sh <tag> <t4|gru|all>
tag=${1:-r03}; cfg=${2:-t4}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
re="rnn1_kernel|rnn_multi"
i=0
files=""
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf /tmp/pr_$i
  rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$re" --output-format csv -d /tmp/pr_$i -o p -- python $root/scripts/bench_rnn.py $cfg > /tmp/pr_$i.log 2>&1
  f=$(find /tmp/pr_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && files="$files $f" || tail -5 /tmp/pr_$i.log
done
python $root/scripts/pmc_table.py $files > $out/${tag}_rnn_pmc_$cfg.md
cat $out/${tag}_rnn_pmc_$cfg.md
