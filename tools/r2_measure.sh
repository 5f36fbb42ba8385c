#!/bin/bash
# Round-2 measurement run on one MI355X: writes everything under gpurun_out/r2/.  Each GPU step has its own time limit
# and the script stops at the first step that fails.
set -o pipefail
out=gpurun_out/r2
mkdir -p $out
step() { echo "== $*" >&2; }
run() { local lim=$1; shift; timeout -k 10 $lim "$@" || { echo "FAILED: $*" >&2; exit 1; }; }

step "one worker, driver line (20/5) x3 and long run"
for i in 1 2 3; do run 120 python bench.py --gpus 1 --steps 20 --warmup 5 >> $out/bench_n1_20_5.jsonl 2>> $out/err.log; done
run 120 python bench.py --gpus 1 --steps 2000 --warmup 200 > $out/bench_n1_2000.json 2>> $out/err.log

step "multi-rank code paths on ONE rank (cost of each transport without peers)"
DTF_P2P_ALLREDUCE=1 DTF_P2P_MODE=replicated run 120 python bench.py --gpus 1 --steps 2000 --warmup 200 --force_collectives > $out/bench_n1_forced_replicated.json 2>> $out/err.log
DTF_P2P_ALLREDUCE=1 DTF_P2P_MODE=sharded run 120 python bench.py --gpus 1 --steps 2000 --warmup 200 --force_collectives > $out/bench_n1_forced_sharded.json 2>> $out/err.log
DTF_PEER_ASYNC=1 run 120 python bench.py --gpus 1 --steps 2000 --warmup 200 --config mlp_async --force_collectives > $out/bench_n1_forced_async.json 2>> $out/err.log

step "two ranks sharing the GPU, self-launched: sync (transport selection) + async second segment"
DTF_SHARE_GPU=1 DTF_DIST_BACKEND=gloo DTF_P2P_ALLREDUCE=1 DTF_PEER_ASYNC=1 run 300 python bench.py --gpus 2 --steps 2000 --warmup 200 > $out/bench_n2_shared_gpu.json 2>> $out/err.log
DTF_SHARE_GPU=1 DTF_DIST_BACKEND=gloo DTF_P2P_ALLREDUCE=1 DTF_PEER_ASYNC=1 run 300 python bench.py --gpus 2 --steps 2000 --warmup 200 --config mlp_async --staleness 4 > $out/bench_n2_shared_gpu_async_staleness4.json 2>> $out/err.log
DTF_SHARE_GPU=1 DTF_DIST_BACKEND=gloo DTF_P2P_ALLREDUCE=1 run 300 python bench.py --gpus 3 --steps 400 --warmup 50 --replicas_to_aggregate 2 --no_second_mode > $out/bench_n3_shared_gpu_R2.json 2>> $out/err.log

step "kernel trace of the one-worker step and of the exchange kernels (one rank)"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_n1 -- python3 $R/bench.py --gpus 1 --steps 2000 --warmup 200 > $R/$out/prof_n1.json 2>> $R/$out/err.log
DTF_P2P_ALLREDUCE=1 DTF_P2P_MODE=sharded run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_n1_sharded -- python3 $R/bench.py --gpus 1 --steps 2000 --warmup 200 --force_collectives > $R/$out/prof_n1_sharded.json 2>> $R/$out/err.log
cd $R
find $out -name "*kernel_stats.csv" | head
echo done
