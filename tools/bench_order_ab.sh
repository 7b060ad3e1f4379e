#!/bin/bash
# Round 3: the tile order inside the model (bench.py's per-kernel HIP-event times), same box, diagnostic build:
# SMOE_PS_NBLOCK=0 (strided order) against the launcher's choice, two alternating repeats.
set -x
D=/tmp/smoe_diag
rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG=-DSMOE_DIAG > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
export SLIMMOE_LIB=$D/slim-switch-moe-vit_amd/libslimmoe_hip.so
O=gpurun_out/r03_bench_order_ab.txt
: > $O
for rep in 1 2; do
  for nb in 0 default; do
    if [ $nb = default ]; then unset SMOE_PS_NBLOCK; else export SMOE_PS_NBLOCK=$nb; fi
    echo "== SMOE_PS_NBLOCK=$nb (repeat $rep)" >> $O
    python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels']
print(d['ms_per_step'], d['roofline']['frac'], {n: k[n]['avg_ms'] for n in ('grouped_gemm_fc1', 'grouped_gemm_fc2', 'qkv_gemm', 'attn_proj_gemm')})
" >> $O
  done
done
cat $O
