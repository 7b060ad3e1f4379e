#!/bin/bash
# Round 3: what bounds the main loop?  Tile stamps of GEMM-1 / GEMM-2 as shipped, and with the operand DMA switched off
# (SMOE_DIAG_FLAGS=1: the MFMAs and LDS reads run on whatever the buffers hold -- the schedule's own floor).
set -x
D=/tmp/smoe_diag
rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG=-DSMOE_DIAG > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
export SMOE_LIB=$D/slim-switch-moe-vit_amd/libslimmoe_hip.so
O=gpurun_out/r03_mainloop_floor.txt
: > $O
for f in 0 1; do
  SMOE_DIAG_FLAGS=$f python3 tools/gemm_stamps.py 9 fc1 >> $O 2>&1
  SMOE_DIAG_FLAGS=$f python3 tools/gemm_stamps.py 9 fc2 >> $O 2>&1
done
grep -v amdgpu.ids $O
