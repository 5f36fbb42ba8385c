#!/bin/bash
# MFMA-pipe utilisation of the 8-wave GEMM as a function of K (M = N = 4096: one 256x256 block per CU), to separate the
# main loop from the per-launch fixed part: one rocprofv3 --pmc run per K, each under its own time limit.
#   bash tools/pmc_ksweep.sh   (on the GPU box; writes gpurun_out/pmc_ksweep/)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_ksweep
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for K in 1024 4096 16384; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $out/k$K -- python3 $R/tools/gemm_pmc_target.py 1 4096 4096 $K > $out/k$K.log 2>&1 || { echo "pmc run K=$K failed"; tail -5 $out/k$K.log; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for K in (1024, 4096, 16384):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/k{K}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_pp_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    if not m:
        print(K, "no rows"); continue
    cyc = m["SQ_BUSY_CYCLES"] / 32.0                     # per shader engine -> kernel duration in cycles
    busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0        # per SIMD (256 CUs x 4)
    print(f"K={K}: kernel cycles {cyc:.0f}, MFMA busy cycles per SIMD {busy:.0f}, MFMA pipes busy {100 * busy / cyc:.1f} %, "
          f"wait-any share of wave cycles {100 * m.get('SQ_WAIT_INST_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1):.0f} %")
PY
