#!/bin/bash
# Round-3 tile-stamp recipe (GPU box, from the repo root): a diagnostic build of the library in /tmp (the production .so in the
# tree stays untouched), then per-tile phase times of GEMM-1 with the direct-store epilogue (variant 9) and with the LDS-staged
# epilogue (variant 14), and of GEMM-2.
set -x
D=/tmp/smoe_diag
rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG=-DSMOE_DIAG > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
export SMOE_LIB=$D/slim-switch-moe-vit_amd/libslimmoe_hip.so
O=gpurun_out/r03_gemm_tile_stamps.txt
: > $O
python3 tools/gemm_stamps.py 9 fc1 >> $O 2>&1
python3 tools/gemm_stamps.py 14 fc1 >> $O 2>&1
SMOE_EPI=none python3 tools/gemm_stamps.py 9 fc1 >> $O 2>&1
python3 tools/gemm_stamps.py 9 fc2 >> $O 2>&1
cat $O
