#!/bin/bash
# Round 3: what do the ping-pong's barrier hand-overs cost?  Main-loop stamps with the interval barriers skipped
# (SMOE_DIAG_FLAGS=2; =3 also without operand DMA).  Results are garbage in these modes; only the cycle counts mean something.
set -x
D=/tmp/smoe_diag
rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG=-DSMOE_DIAG > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
export SMOE_LIB=$D/slim-switch-moe-vit_amd/libslimmoe_hip.so
O=gpurun_out/r03_barrier_cost.txt
: > $O
for f in 0 1 2 3; do
  SMOE_DIAG_FLAGS=$f timeout -k 10 120 python3 tools/gemm_stamps.py 9 fc1 >> $O 2>&1
  SMOE_DIAG_FLAGS=$f timeout -k 10 120 python3 tools/gemm_stamps.py 9 fc2 >> $O 2>&1
done
grep -v amdgpu.ids $O
