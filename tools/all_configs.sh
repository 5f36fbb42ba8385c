#!/bin/bash
# One run of every benchmark configuration on one GPU (run on the GPU box); JSON lines under gpurun_out/all_configs/.
cd "$(dirname "$0")/.."
out=gpurun_out/all_configs
mkdir -p $out
( while true; do date >> $out/heartbeat.txt; sleep 45; done ) &
HB=$!
rc=0
run() { name=$1; shift; timeout -k 10 420 python bench.py "$@" 2> $out/$name.err > $out/$name.json || rc=1; python -c "import json;d=json.load(open('$out/$name.json'));print('$name', round(d['ms_per_step'],5), 'ms/step', round(d['value']), d['unit'])"; }
run mlp_sync_20_5 --gpus 1 --steps 20 --warmup 5
run mlp_sync_2000 --gpus 1 --steps 2000 --warmup 200
run mlp_async_2000 --config mlp_async --steps 2000 --warmup 200
run wide_mlp_sync_bf16 --config wide_mlp_sync --dtype bf16 --steps 60 --warmup 20
run wide_mlp_async_bf16 --config wide_mlp_async --dtype bf16 --steps 60 --warmup 20
run resnet50_sync --config resnet50_sync --steps 30 --warmup 10
kill $HB
exit $rc
