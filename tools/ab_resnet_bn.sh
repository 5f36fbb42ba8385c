#!/bin/bash
# A/B of builds of the kernel library inside the ResNet-50 step: tools/ab_resnet_bn.sh <variant>...
# ("new" = the default library, anything else = distributed_tensorflow_amd/lib/variants/libdtf_kernels_<variant>.so)
cd "$(dirname "$0")/.."
out=gpurun_out/bn
mkdir -p $out
( while true; do date >> $out/heartbeat.txt; sleep 45; done ) &
HB=$!
rc=0
for v in "$@"; do
  if [ "$v" = new ]; then L=""; else L=$PWD/distributed_tensorflow_amd/lib/variants/libdtf_kernels_$v.so; fi
  DTF_KERNELS_LIB=$L timeout -k 10 420 python bench.py --config resnet50_sync --steps 30 --warmup 10 2> $out/err_$v.txt > $out/resnet_$v.json || { rc=1; break; }
  python -c "import json;d=json.load(open('$out/resnet_$v.json'));print('$v',d['ms_per_step'],d['value'])" | tee -a $out/resnet_ab.txt
done
kill $HB
exit $rc
