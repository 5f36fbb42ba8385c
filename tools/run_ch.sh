cd tools
for ch in 1 2 4; do echo "== CH=$ch"; DTF_KERNELS_LIB=$GRAFT_REPO_ROOT/distributed_tensorflow_amd/lib/libdtf_kernels_ch$ch.so timeout -k 10 100 python sweep_fwd.py | head -6; done
