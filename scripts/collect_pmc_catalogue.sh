#!/bin/bash
# Counters of the heavy kernels of the BASELINE configs[4] step (100M-item catalogue, 128-wide layers): HBM traffic
# (FETCH_SIZE / WRITE_SIZE, one counter per pass) and SQ activity (one set per pass); no trace domains besides
# --kernel-trace.  Counter collection serialises the dispatches: durations are the kernels ALONE.
#   usage on the GPU box:  bash scripts/collect_pmc_catalogue.sh <tag>     -> gpurun_out/<tag>_catalogue_pmc.md
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
re="dw_wide_kernel|pgemm_fast_kernel|pgemm3_kernel|pgemm_dw_kernel|rnn_multi_fwd|rnn_multi_bwd|dw_multi_kernel|pgemm_kloop|att_prod_bwd|dw_wide_reduce|proj_x3|att_l0_fwd_kernel"
i=0
files=""
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pc_$i
  timeout 900 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "$re" --output-format csv -d /tmp/pc_$i -o p -- python $root/bench.py --config catalogue100m --no-cpu-baseline --no-catalogue --no-extra --steps 4 --warmup 2 > /tmp/pc_$i.log 2>&1
  f=$(find /tmp/pc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && files="$files $f" || tail -5 /tmp/pc_$i.log
done
python $root/scripts/pmc_table.py $files > $out/${tag}_catalogue_pmc.md
cat $out/${tag}_catalogue_pmc.md
