#!/bin/bash
# Steady-state kernel table of the ResNet-50 step (run on the GPU box): rocprofv3 kernel trace of a short bench, the
# last 10 steps summarised by tools/steady_state_stats.py into gpurun_out/resnet_table/steady_state_kernels.txt.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/resnet_table"
mkdir -p "$OUT"
( while true; do date >> "$OUT/heartbeat.txt"; sleep 45; done ) &
HB=$!
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/resnet_trace -o step -- python3 "$ROOT/bench.py" --config resnet50_sync --steps 10 --warmup 5 "$@" > "$OUT/trace.log" 2>&1
rc=$?
kill $HB
[ $rc = 0 ] || exit $rc
ms=$(python3 -c "import json,sys; print(json.loads([l for l in open('$OUT/trace.log') if l.startswith('{')][-1])['ms_per_step'])")
python3 "$ROOT/tools/steady_state_stats.py" $(find /tmp/resnet_trace -name "*kernel_trace.csv" | head -1) --window_ms $(python3 -c "print(10*$ms)") --steps 10 --top 45 > "$OUT/steady_state_kernels.txt"
echo "ms_per_step under the profiler: $ms" >> "$OUT/steady_state_kernels.txt"
