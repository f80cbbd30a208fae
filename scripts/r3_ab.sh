#!/bin/bash
# A/B of an environment switch inside ONE gpurun call: bash scripts/r3_ab.sh VAR v1 v2 ... [-- bench args]
var=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for rep in 1 2; do
  for v in "${vals[@]}"; do
    line=$(env $var=$v python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40 "$@" 2>/dev/null | grep '^{"metric' | tail -1)
    echo "$var=$v  $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')"
  done
done
