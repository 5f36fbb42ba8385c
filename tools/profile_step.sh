#!/bin/bash
# Per-kernel profile + hardware counters of the flagship step (run on the GPU box; output under gpurun_out/).
#   tools/profile_step.sh [extra bench.py flags]
# Pass 1: kernel trace + stats.  Pass 2/3: PMC counters, each in a run of its own (counter blocks have small limits).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/profile_step"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o step -- python3 "$ROOT/bench.py" --steps 1000 --warmup 100 "$@" > "$OUT/trace.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -o step -- python3 "$ROOT/bench.py" --steps 200 --warmup 20 "$@" > "$OUT/pmc_sq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mem" -o step -- python3 "$ROOT/bench.py" --steps 200 --warmup 20 "$@" > "$OUT/pmc_mem.log" 2>&1
ls "$OUT"/*
