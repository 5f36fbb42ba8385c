cd tools
for v in "$@"; do echo "== $v"; DTF_KERNELS_LIB=$GRAFT_REPO_ROOT/distributed_tensorflow_amd/lib/libdtf_kernels_$v.so timeout -k 10 100 python kbench.py | grep -E "fwd|head|wgrad\+sgd\+tail|full"; done
