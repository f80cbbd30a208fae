bash scripts/prof_step.sh r3ks --config kuaishou > /dev/null 2>&1
tail -2 gpurun_out/r3ks_timeline.txt
head -24 gpurun_out/r3ks_stats.md | cut -c1-130
