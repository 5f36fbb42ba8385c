#!/bin/bash
# Rehearsal on ONE GPU: the training script as 1 ps + 2 workers (processes sharing the GPU, gloo process group, peer-mapped
# sync step), once with the plain loop and once with the loop that launches ahead (the default); prints whether the
# workers' outputs are identical.   tools/cli_two_workers.sh [train_steps]
cd "$(dirname "$0")/.."
steps=${1:-3000}
out=gpurun_out/cli2
mkdir -p $out
export DTF_DIST_BACKEND=gloo DTF_SHARE_GPU=1 DTF_P2P_ALLREDUCE=1
run() {
  mode=$1
  p0=$((29700 + RANDOM % 200)); p1=$((p0 + 1)); p2=$((p0 + 2))
  common="--ps_hosts=127.0.0.1:$p2 --worker_hosts=127.0.0.1:$p0,127.0.0.1:$p1 --sync_replicas --train_steps=$steps --synthetic_data --synthetic_train_size=4000 --learning_rate=0.5 --seed=7 --validation_every=1000 --log_every=50 --dist_backend=gloo"
  DTF_LOOP_OVERLAP=$mode timeout -k 10 200 python distributed.py --job_name=ps --task_index=0 $common > $out/ps_$mode.txt 2> $out/ps_$mode.err &
  PS=$!
  DTF_LOOP_OVERLAP=$mode timeout -k 10 200 python distributed.py --job_name=worker --task_index=1 $common > $out/w1_$mode.txt 2> $out/w1_$mode.err &
  W1=$!
  DTF_LOOP_OVERLAP=$mode timeout -k 10 200 python distributed.py --job_name=worker --task_index=0 $common > $out/w0_$mode.txt 2> $out/w0_$mode.err
  rc=$?
  wait $W1 || rc=1
  wait $PS || rc=1
  return $rc
}
run 0 || { echo "plain loop run failed"; tail -3 $out/w0_0.err; exit 1; }
run 1 || { echo "launch-ahead run failed"; tail -3 $out/w0_1.err; exit 1; }
for w in w0 w1; do
  if diff <(grep -v "elapsed\|examples/sec" $out/${w}_0.txt) <(grep -v "elapsed\|examples/sec" $out/${w}_1.txt) > /dev/null; then echo "$w: identical output"; else echo "$w: outputs DIFFER"; fi
  grep "elapsed" $out/${w}_0.txt $out/${w}_1.txt
done
tail -3 $out/w0_1.txt
