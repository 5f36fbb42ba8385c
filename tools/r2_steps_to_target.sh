#!/bin/bash
# Steps to target loss with the round-2 engines, 1/2/4 ranks sharing the one GPU of the box (arithmetic of an N-GPU run,
# not its timing): sync on both peer-mapped transports, async with a pull every step and every 5th step.
set -o pipefail
out=gpurun_out/r2_steps.txt
: > $out
export DTF_DIST_BACKEND=gloo DTF_SHARE_GPU=1 DTF_P2P_ALLREDUCE=1 DTF_PEER_ASYNC=1
run() { local n=$1; shift; timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) tools/steps_to_target_multi.py --lr 0.1 "$@" 2>/dev/null | grep '^{' >> $out || { echo "failed: $n $*"; exit 1; }; }
for n in 1 2 4; do
  DTF_P2P_MODE=replicated run $n --mode sync
  [ $n -gt 1 ] && DTF_P2P_MODE=sharded run $n --mode sync
  run $n --mode async --staleness 0
  [ $n -gt 1 ] && run $n --mode async --staleness 4
done
cat $out
