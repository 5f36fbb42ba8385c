#!/bin/bash
# Sweep of HIP runtime environment settings over the flagship bench (short run twice, long run once).
run() { # label, env...
  label=$1; shift
  a=$(env "$@" timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step']*1000,2))")
  b=$(env "$@" timeout -k 10 120 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step']*1000,2))")
  c=$(env "$@" timeout -k 10 120 python bench.py --steps 2000 --warmup 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step']*1000,2))")
  echo "$label: 20/5 -> $a $b ; 2000 -> $c"
}
B="DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run default X=1
run pkt0 $B
run pkt0_batch1 $B DEBUG_HIP_GRAPH_BATCH_SIZE=1
run pkt0_batch32 $B DEBUG_HIP_GRAPH_BATCH_SIZE=32
run pkt0_batch1024 $B DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run pkt0_maxbatch8 $B DEBUG_CLR_MAX_BATCH_SIZE=8
run pkt0_maxbatch4096 $B DEBUG_CLR_MAX_BATCH_SIZE=4096
run pkt0_kernargcopy0 $B DEBUG_HIP_KERNARG_COPY_OPT=0
run pkt0_kernargcopy1 $B DEBUG_HIP_KERNARG_COPY_OPT=1
run pkt0_direct0 $B AMD_DIRECT_DISPATCH=0
run pkt0_activewait $B ROC_ACTIVE_WAIT_TIMEOUT=200
run pkt0_twolaunch $B DTF_PIPELINED_STEP=0
run twolaunch DTF_PIPELINED_STEP=0
