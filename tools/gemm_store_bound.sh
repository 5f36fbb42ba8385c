#!/bin/bash
# How much of a launch of the 8-wave bf16 GEMM is the final burst of stores?  Times 4096 x 4096 x 4096 (one 256x256
# block per CU) with the stamped diagnostic build (tools/build_diag.sh) while leaving out the stores of C, of the
# transposed copy, of both; then the MFMA-pipe utilisation of the production kernel (rocprofv3 --pmc).
#   bash tools/gemm_store_bound.sh      (on the GPU box; writes gpurun_out/gemm_store_bound.txt)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/gemm_store_bound.txt
: > $out
for skip in 0 1 2 3; do
  DTF_KERNELS_LIB=$R/distributed_tensorflow_amd/lib/libdtf_kernels_diag.so DTF_GEMM_SKIP_STORES=$skip \
    timeout -k 10 120 python3 $R/tools/gemm_store_bound.py $skip >> $out 2>&1 || { tail -5 $out; exit 1; }
done
cat $out
