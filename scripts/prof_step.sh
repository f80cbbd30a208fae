#!/bin/bash
# rocprofv3 kernel trace of `python bench.py <args>` -> per-kernel table + one-step timeline under gpurun_out/
# usage (on the GPU box, from the repo root): bash scripts/prof_step.sh <tag> [bench.py args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$tag -o bench -- python $root/bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 20 "$@" > /tmp/prof_$tag.log 2>&1
grep -E "timed|Error|error" /tmp/prof_$tag.log | head -5
cd $root
f=$(find /tmp/prof_$tag -name "*.db" | head -1)
python scripts/rocpd_stats.py $f > gpurun_out/${tag}_stats.md
python scripts/step_timeline.py $f > gpurun_out/${tag}_timeline.txt
tail -3 gpurun_out/${tag}_timeline.txt
