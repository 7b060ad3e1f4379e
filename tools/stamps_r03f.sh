#!/bin/bash
# Round 3: deferred stores of the 256-row direct-store tile (qkv shape; GEMM-1 forced onto 256-row tiles = variant 11): two diagnostic
# builds in /tmp, -DPS_ND=0 (every store at the boundary) and the default (the last 7 under the next tile's K-tiles).
set -x
for nd in 0 7; do
  D=/tmp/smoe_diag_nd$nd
  rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
  rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
  make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG="-DSMOE_DIAG -DPS_ND=$nd" > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
done
O=gpurun_out/r03_deferred_stores.txt
: > $O
for nd in 0 7 0 7; do
  echo "== PS_ND=$nd" >> $O
  SMOE_LIB=/tmp/smoe_diag_nd$nd/slim-switch-moe-vit_amd/libslimmoe_hip.so python3 tools/gemm_stamps.py 9 qkv >> $O 2>&1
  SMOE_LIB=/tmp/smoe_diag_nd$nd/slim-switch-moe-vit_amd/libslimmoe_hip.so python3 tools/gemm_stamps.py 11 fc1 >> $O 2>&1
done
for nd in 0 7; do
  echo "== PS_ND=$nd: gemm_ab 9 11 (qkv: 9 = 256-row tiles; gemm1: 9 = 320-row, 11 = 256-row)" >> $O
  SMOE_LIB=/tmp/smoe_diag_nd$nd/slim-switch-moe-vit_amd/libslimmoe_hip.so python3 tools/gemm_ab.py --cold 9 11 >> $O 2>&1
done
grep -v amdgpu.ids $O
