#!/bin/bash
# Every headline configuration on ONE box, one JSON line each, plus the rocprofv3 kernel table of the flagship step:
#     bash tools/final_numbers.sh OUTDIR
set -e
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/final}
mkdir -p "$out"
run() { name=$1; shift; echo "[final] $name"; timeout -k 10 "${LIMIT:-300}" python3 bench.py "$@" > "$out/$name.json" 2> "$out/$name.err"; }
run mlp_sync_n1_20_5 --gpus 1 --steps 20 --warmup 5
run mlp_sync_n1_2000 --gpus 1 --steps 2000 --warmup 200
export DTF_P2P_ALLREDUCE=1
run mlp_sync_forced_pipelined_2000 --gpus 1 --steps 2000 --warmup 200 --force_collectives --collective p2p_pipelined --target_loss 0
run mlp_sync_forced_replicated_2000 --gpus 1 --steps 2000 --warmup 200 --force_collectives --collective p2p_replicated --target_loss 0
run mlp_sync_forced_sharded_2000 --gpus 1 --steps 2000 --warmup 200 --force_collectives --collective p2p_sharded --target_loss 0
run mlp_async_forced_2000 --gpus 1 --steps 2000 --warmup 200 --config mlp_async --force_collectives --target_loss 0
unset DTF_P2P_ALLREDUCE
run mlp_256_256 --gpus 1 --steps 2000 --warmup 200 --hidden_units 256 --batch_size 256 --target_loss 0
run wide_bf16_n1 --gpus 1 --steps 30 --warmup 10 --config wide_mlp_sync --dtype bf16
run wide_bf16_forced --gpus 1 --steps 30 --warmup 10 --config wide_mlp_sync --dtype bf16 --force_collectives
LIMIT=600 run resnet50_n1 --gpus 1 --steps 20 --warmup 15 --config resnet50_sync
LIMIT=600 run resnet50_forced --gpus 1 --steps 20 --warmup 15 --config resnet50_sync --force_collectives
echo "[final] two ranks sharing the GPU"
DTF_DIST_BACKEND=gloo DTF_SHARE_GPU=1 DTF_P2P_ALLREDUCE=1 DTF_PEER_ASYNC=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 2000 --warmup 200 > "$out/mlp_sync_n2_sharing_one_gpu.json" 2> "$out/mlp_sync_n2_sharing_one_gpu.err"
python3 - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    for line in open(f):
        if line.startswith("{"):
            d = json.loads(line)
            print(f"{os.path.basename(f):44s} {d['ms_per_step'] * 1e3:10.2f} us/step  {d['value']:14.0f} examples/s  {d['config'].get('transport') or ''}")
PY
