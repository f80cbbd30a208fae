#!/bin/bash
# One GPU call that regenerates every rocprof artefact quoted in DESIGN.md / the bench line (outputs under gpurun_out/,
# copied into profiles/ by hand).   usage on the GPU box:  bash scripts/collect_profiles.sh <round tag, e.g. r02>
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
bash scripts/prof_step.sh ${tag}_fp32
bash scripts/prof_step.sh ${tag}_bf16 --precision bf16
cd /tmp && export TMPDIR=/tmp
# rocprofv3 --kernel-trace --stats of the bench command itself (what the bench line's kernel durations must agree with)
rm -rf /tmp/pb && rocprofv3 --kernel-trace --stats --output-format rocpd -d /tmp/pb -o bench -- python $root/bench.py --no-cpu-baseline > /tmp/pb.log 2>&1
grep "^{\"metric" /tmp/pb.log | tail -1 > $out/${tag}_bench_line_traced.json
f=$(find /tmp/pb -name "*.db" | head -1); [ -n "$f" ] && python $root/scripts/rocpd_stats.py $f > $out/${tag}_bench_kernel_stats.md
# HBM traffic of the gather kernel: separate --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && rocprofv3 --kernel-trace --pmc $c --kernel-include-regex gather_hist_fwd --output-format csv -d /tmp/pmc_$c -o g -- python $root/scripts/prof_kernels.py gather > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_gather_pmc_$c.csv
  tail -2 /tmp/pmc_$c.log
done
cd $root
python scripts/bench_hgemm.py > $out/${tag}_hgemm_isolated.txt 2>&1
python scripts/bench_dw.py > $out/${tag}_dw_isolated.txt 2>&1
python scripts/bench_att_fp32.py > $out/${tag}_att_fp32_isolated.txt 2>&1
python scripts/bench_rnn.py > $out/${tag}_rnn_isolated.txt 2>&1
python scripts/bench_small_gemm.py > $out/${tag}_small_gemm_isolated.txt 2>&1
ls -la $out | tail -20
