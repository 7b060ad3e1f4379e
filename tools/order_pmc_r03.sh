#!/bin/bash
# Round-3: L2 counters of GEMM-1 under the strided tile order (SMOE_PS_NBLOCK=0) and XCD-contiguous runs over n-blocks of 4 and 12
# n-tiles (GPU box, from the repo root; diagnostic build in /tmp; one counter set per pass).
set -x
export TMPDIR=/tmp
D=/tmp/smoe_diag
rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG=-DSMOE_DIAG > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
export SMOE_LIB=$D/slim-switch-moe-vit_amd/libslimmoe_hip.so
O=gpurun_out/prof_r03c
mkdir -p $O
for nb in 0 4 12; do
  export SMOE_PS_NBLOCK=$nb
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_nb$nb -o f -- python3 tools/gemm_prof.py 9 fc1 3 > $O/fetch_nb$nb.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc_nb$nb -o t -- python3 tools/gemm_prof.py 9 fc1 3 > $O/tcc_nb$nb.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/ea_nb$nb -o e -- python3 tools/gemm_prof.py 9 fc1 3 > $O/ea_nb$nb.log 2>&1
done
find $O -name "*counter_collection.csv"
