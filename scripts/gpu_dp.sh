b() { env "$@" python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40 2>/dev/null | grep '^{"metric' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d.get("rccl_ranks"))'; }
bl() { env "$@" python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40 --local-bn 2>/dev/null | grep '^{"metric' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step", d.get("rccl_ranks"))'; }
for rep in 1 2; do
echo -n "plain            : "; b X=1
echo -n "DP world 1 syncBN: "; b CLSR_FORCE_DP=1
echo -n "DP world 1 local : "; bl CLSR_FORCE_DP=1
done
CLSR_FORCE_DP=1 bash scripts/prof_step.sh r3dp > /dev/null 2>&1
tail -n 3 gpurun_out/r3dp_timeline.txt
