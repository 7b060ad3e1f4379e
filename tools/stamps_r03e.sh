#!/bin/bash
# Round 3: the direct epilogue's store tail under cache-policy bits (two diagnostic builds in /tmp: default policy, nt).
set -x
for aux in 0 2; do
  D=/tmp/smoe_diag$aux
  rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
  rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
  make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG="-DSMOE_DIAG -DPS_STORE_AUX=$aux" > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
done
O=gpurun_out/r03_store_policy.txt
: > $O
for aux in 0 2 0 2; do
  echo "== PS_STORE_AUX=$aux" >> $O
  SMOE_LIB=/tmp/smoe_diag$aux/slim-switch-moe-vit_amd/libslimmoe_hip.so python3 tools/gemm_stamps.py 9 fc1 >> $O 2>&1
done
grep -v amdgpu.ids $O
