#!/bin/bash
# Round-3 tile stamps of GEMM-2 (f32 row-mapped store + residual): buffer-addressed staged epilogue (variant 9) against the
# flat-addressed one (variant 14).  GPU box, from the repo root; diagnostic build in /tmp.
set -x
D=/tmp/smoe_diag
rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG=-DSMOE_DIAG > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
export SMOE_LIB=$D/slim-switch-moe-vit_amd/libslimmoe_hip.so
O=gpurun_out/r03_gemm2_tile_stamps.txt
: > $O
python3 tools/gemm_stamps.py 9 fc2 >> $O 2>&1
python3 tools/gemm_stamps.py 14 fc2 >> $O 2>&1
cat $O
