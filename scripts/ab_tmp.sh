run() { # label, env..., -- args
  label=$1; shift
  a=$(env "$@" 2>&1 | grep -o "\"ms_per_step\": [0-9.]*" | head -1)
  echo "$label $a"
}
B="python bench.py --no-cpu-baseline --no-catalogue --no-extra --steps 40"
run "eager" $B
run "graph" $B --graph
for q in 1 2 3 4 5 6 8; do run "graph FORCE_GRAPH_QUEUES=$q" DEBUG_HIP_FORCE_GRAPH_QUEUES=$q $B --graph; done
run "graph PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $B --graph
run "graph PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $B --graph
for q in 2 6 8 16; do run "eager GPU_MAX_HW_QUEUES=$q" GPU_MAX_HW_QUEUES=$q $B; run "graph GPU_MAX_HW_QUEUES=$q" GPU_MAX_HW_QUEUES=$q $B --graph; done
run "eager DYNAMIC_QUEUES=0" DEBUG_HIP_DYNAMIC_QUEUES=0 $B
run "eager DYNAMIC_QUEUES=1" DEBUG_HIP_DYNAMIC_QUEUES=1 $B
run "eager STREAMOPS_CP_WAIT=0" GPU_STREAMOPS_CP_WAIT=0 $B
run "eager STREAMOPS_CP_WAIT=1" GPU_STREAMOPS_CP_WAIT=1 $B
run "eager HWQ8 DW_STREAMS=2" GPU_MAX_HW_QUEUES=8 CLSR_DW_STREAMS=2 $B
run "eager bf16" $B --precision bf16
run "eager bf16 HWQ8" GPU_MAX_HW_QUEUES=8 $B --precision bf16
run "graph bf16" $B --precision bf16 --graph
