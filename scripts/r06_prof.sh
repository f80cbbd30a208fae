#!/bin/bash
# rocprof evidence of the round: gather kernel trace + HBM counters (rotated id sets), embedding kernels trace + counters,
# rocprofv3 --stats of the bench command
tag=${1:-r06}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
bash scripts/prof_gather.sh $tag | tail -3
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c && timeout 500 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex gather_hist_fwd --output-format csv -d /tmp/pmc_$c -o g -- python $root/scripts/prof_kernels.py gather > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/${tag}_gather_pmc_$c.csv
  tail -1 /tmp/pmc_$c.log
done
cd $root
EMBED_SITES=item bash scripts/prof_embed.sh ${tag}_item | tail -12
bash scripts/prof_embed.sh ${tag} | tail -4
cd /tmp
rm -rf /tmp/pb && timeout 900 rocprofv3 --kernel-trace --stats --output-format rocpd -d /tmp/pb -o bench -- python $root/bench.py --no-cpu-baseline > /tmp/pb.log 2>&1
grep "^{\"metric" /tmp/pb.log | tail -1 > $out/${tag}_bench_line_traced.json
f=$(find /tmp/pb -name "*.db" | head -1); [ -n "$f" ] && python $root/scripts/rocpd_stats.py $f > $out/${tag}_bench_kernel_stats.md
head -12 $out/${tag}_bench_kernel_stats.md
