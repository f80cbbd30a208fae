#!/bin/bash
# kernel resource usage of one HIP source: name, VGPRs, AGPRs, scratch bytes, occupancy, LDS   usage: scripts/kres.sh clsr_amd/csrc/rnn.hip [filter] [extra flags]
f=$1; pat=${2:-.}; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -Iclsr_amd/csrc "$@" -c $f -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur={}
for l in sys.stdin:
    m=re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", l)
    if not m: continue
    k,v=m.groups()
    if k=="Function Name":
        cur={"n":v}
    cur[k]=v
    if k.startswith("LDS"):
        print("%-70s V %4s A %4s scratch %5s occ %2s lds %6s" % (cur["n"][:70], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), v))
' | grep -E "$pat"
