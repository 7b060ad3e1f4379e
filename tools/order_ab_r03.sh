#!/bin/bash
# Round-3 A/B of the persistent GEMM's tile order (GPU box, from the repo root): diagnostic build in /tmp, then the bench
# model's four GEMM shapes with the strided order (n_block 0) against XCD-contiguous runs over n-blocks of 3 / 4 / 6 / 9 / 12 n-tiles
# (widths that do not divide a shape's n-tiles fall back to the strided order).
set -x
D=/tmp/smoe_diag
rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG=-DSMOE_DIAG > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
export SMOE_LIB=$D/slim-switch-moe-vit_amd/libslimmoe_hip.so
python3 tools/gemm_ab.py 9:0 9:3 9:4 9:6 9:9 9:12 > gpurun_out/r03_tile_order_ab.txt 2>&1
python3 tools/gemm_ab.py --cold 9:0 9:3 9:4 9:6 9:12 > gpurun_out/r03_tile_order_ab_cold.txt 2>&1
cat gpurun_out/r03_tile_order_ab.txt gpurun_out/r03_tile_order_ab_cold.txt
# is the tile boundary bound by the chip (all 256 workgroups store at once) or by the CU?  the same tiles on 256 / 128 / 64 workgroups
O=gpurun_out/r03_boundary_vs_grid.txt
: > $O
for g in 256 128 64; do
  echo "== SMOE_PS_GRID=$g" >> $O
  SMOE_PS_GRID=$g python3 tools/gemm_stamps.py 9 fc2 >> $O 2>&1
  SMOE_PS_GRID=$g python3 tools/gemm_stamps.py 9 fc1 >> $O 2>&1
done
grep -v amdgpu.ids $O
