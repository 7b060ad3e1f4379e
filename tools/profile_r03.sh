#!/bin/bash
# Round-3 profile recipe (GPU box, from the repo root; output under gpurun_out/prof_r03, summaries copied to profiles/ by hand).
# Counters in their own passes (--kernel-trace / --stats alone; --pmc alone).
set -x
export TMPDIR=/tmp
O=gpurun_out/prof_r03
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 tools/train_bench.py model 128 6 > $O/train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dispatch -o dispatch -- python3 tools/dispatch_prof.py 20 > $O/dispatch.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_dispatch -o f -- python3 tools/dispatch_prof.py 3 > $O/fetch_dispatch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_dispatch -o w -- python3 tools/dispatch_prof.py 3 > $O/write_dispatch.log 2>&1
for shape in fc1 fc2; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmcA_$shape -o a -- python3 tools/gemm_prof.py 9 $shape 3 > $O/pmcA_$shape.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $O/pmcB_$shape -o b -- python3 tools/gemm_prof.py 9 $shape 3 > $O/pmcB_$shape.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$shape -o f -- python3 tools/gemm_prof.py 9 $shape 3 > $O/fetch_$shape.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$shape -o w -- python3 tools/gemm_prof.py 9 $shape 3 > $O/write_$shape.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc_$shape -o t -- python3 tools/gemm_prof.py 9 $shape 3 > $O/tcc_$shape.log 2>&1
done
find $O -name "*.csv" | wc -l
