#!/bin/bash
# One recipe file for the GPU-box diagnostics of the persistent GEMM and the bench profiles (run from the repo root inside a
# gpurun call; output under gpurun_out/, summaries are copied to profiles/ by hand).  Replaces the one-off stamps_r03*.sh /
# order_*_r03.sh / profile_r0*.sh scripts of earlier rounds (their output files are the ones profiles/README.md cites).
#
#   tools/diag.sh stamps  OUT [ENV=VAL ...] -- VARIANT:SHAPE [VARIANT:SHAPE ...]   per-tile phase times (s_memtime stamps)
#        e.g.  tools/diag.sh stamps gpurun_out/r04_stamps.txt -- 9:fc1 14:fc1 9:fc2
#              tools/diag.sh stamps gpurun_out/r04_nobar.txt SMOE_DIAG_FLAGS=2 -- 9:fc1 9:fc2      (bit 0 no DMA, bit 1 no barriers)
#              tools/diag.sh stamps gpurun_out/r04_grid64.txt SMOE_PS_GRID=64 -- 9:fc2
#   tools/diag.sh order-ab OUT                       tile orders of the four bench GEMM shapes, warm and cold (tools/gemm_ab.py)
#   tools/diag.sh order-pmc OUTDIR                   L2 counters of GEMM-1 under three tile orders
#   tools/diag.sh bench-order-ab OUT                 tile order inside the model (bench.py per-kernel times)
#   tools/diag.sh train-profile TAG                  rocprofv3 kernel stats of the two training steps only
#   tools/diag.sh profile TAG                        rocprofv3 kernel stats of bench / train / dispatch + PMC sets of both GEMMs
#                                                    -> gpurun_out/prof_TAG   (counters in passes of their own: --pmc alone)
#   tools/diag.sh tiny-profile TAG                   the reference's own DeiT-Tiny MoE models: eval tables + rocprofv3 kernel stats
#   tools/diag.sh cfg4-profile TAG                   BASELINE cfg 4's model (ViT-L/16 @384, E 32) at batch 64 + rocprofv3 kernel stats
# `stamps`, `order-*` and `bench-order-ab` use a DIAGNOSTIC build of the library made in /tmp (the production .so stays untouched).
set -e
export TMPDIR=/tmp
cmd=$1; shift || true

build_diag() {
  D=/tmp/smoe_diag
  rm -rf $D && mkdir -p $D/slim-switch-moe-vit_amd && cp -r slim-switch-moe-vit_amd/csrc $D/slim-switch-moe-vit_amd/ && cp -r include $D/
  rm -f $D/slim-switch-moe-vit_amd/csrc/*.o
  make -C $D/slim-switch-moe-vit_amd/csrc -j16 DIAG=-DSMOE_DIAG > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
  export SMOE_LIB=$D/slim-switch-moe-vit_amd/libslimmoe_hip.so SLIMMOE_LIB=$D/slim-switch-moe-vit_amd/libslimmoe_hip.so
}

PMC_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16"
PMC_B="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"

case "$cmd" in
  stamps)
    O=$1; shift
    while [ "$1" != "--" ] && [ -n "$1" ]; do export "$1"; shift; done
    shift
    build_diag
    : > $O
    for vs in "$@"; do
      timeout -k 10 180 python3 tools/gemm_stamps.py ${vs%%:*} ${vs##*:} >> $O 2>&1
    done
    grep -v amdgpu.ids $O ;;
  order-ab)
    O=$1; build_diag
    python3 tools/gemm_ab.py 9:0 9:3 9:4 9:6 9:9 9:12 > $O 2>&1
    python3 tools/gemm_ab.py --cold 9:0 9:3 9:4 9:6 9:12 > ${O%.txt}_cold.txt 2>&1
    cat $O ${O%.txt}_cold.txt ;;
  order-pmc)
    O=$1; mkdir -p $O; build_diag
    for nb in 0 4 12; do
      export SMOE_PS_NBLOCK=$nb
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_nb$nb -o f -- python3 tools/gemm_prof.py 9 fc1 3 > $O/fetch_nb$nb.log 2>&1
      rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc_nb$nb -o t -- python3 tools/gemm_prof.py 9 fc1 3 > $O/tcc_nb$nb.log 2>&1
      rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/ea_nb$nb -o e -- python3 tools/gemm_prof.py 9 fc1 3 > $O/ea_nb$nb.log 2>&1
    done
    find $O -name "*counter_collection.csv" ;;
  bench-order-ab)
    O=$1; build_diag
    : > $O
    for rep in 1 2; do
      for nb in 0 default; do
        if [ $nb = default ]; then unset SMOE_PS_NBLOCK; else export SMOE_PS_NBLOCK=$nb; fi
        echo "== SMOE_PS_NBLOCK=$nb (repeat $rep)" >> $O
        python3 bench.py --no-cpu-baseline --clock-seconds 0 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['ms_per_step'], {k: v['avg_ms'] for k, v in d['kernels'].items() if 'gemm' in k or 'ffn' in k})" >> $O
      done
    done
    cat $O ;;
  profile)
    TAG=$1; O=gpurun_out/prof_$TAG; mkdir -p $O
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py --no-cpu-baseline --clock-seconds 0 --steps 20 --warmup 5 > $O/bench.log 2>&1
    python3 tools/train_bench.py model 128 10 > $O/train_unprofiled.log 2>&1
    python3 tools/train_bench.py model 128 10 resmoe_base_patch16_224_expert8_top1 > $O/train_resmoe_unprofiled.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 tools/train_bench.py model 128 6 > $O/train.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_resmoe -o train -- python3 tools/train_bench.py model 128 6 resmoe_base_patch16_224_expert8_top1 > $O/train_resmoe.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/dispatch -o dispatch -- python3 tools/dispatch_prof.py 20 > $O/dispatch.log 2>&1
    for set in A B; do   # attention forward + backward (counters before any change)
      if [ $set = A ]; then C="$PMC_A"; else C="$PMC_B"; fi
      rocprofv3 --pmc $C --output-format csv -d $O/pmc${set}_attn -o a -- python3 tools/attn_prof.py > $O/pmc${set}_attn.log 2>&1
    done
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_attn -o f -- python3 tools/attn_prof.py > $O/fetch_attn.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_attn -o w -- python3 tools/attn_prof.py > $O/write_attn.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_dispatch -o f -- python3 tools/dispatch_prof.py 3 > $O/fetch_dispatch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_dispatch -o w -- python3 tools/dispatch_prof.py 3 > $O/write_dispatch.log 2>&1
    for shape in fc1 fc2; do      # (`ffn` = the fused launch: only with a library built with `make FFN=-DSMOE_FFN_FUSED`)
      rocprofv3 --pmc $PMC_A --output-format csv -d $O/pmcA_$shape -o a -- python3 tools/gemm_prof.py 9 $shape 3 > $O/pmcA_$shape.log 2>&1
      rocprofv3 --pmc $PMC_B --output-format csv -d $O/pmcB_$shape -o b -- python3 tools/gemm_prof.py 9 $shape 3 > $O/pmcB_$shape.log 2>&1
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$shape -o f -- python3 tools/gemm_prof.py 9 $shape 3 > $O/fetch_$shape.log 2>&1
      rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$shape -o w -- python3 tools/gemm_prof.py 9 $shape 3 > $O/write_$shape.log 2>&1
      rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc_$shape -o t -- python3 tools/gemm_prof.py 9 $shape 3 > $O/tcc_$shape.log 2>&1
    done
    find $O -name "*.csv" | wc -l ;;
  profile2)        # second half (a gpurun call of its own): the reference's own models, cfg 4, the forced expert-parallel line
    TAG=$1; O=gpurun_out/prof_$TAG; mkdir -p $O
    "$0" tiny-profile $TAG > $O/tiny_profile.log 2>&1
    "$0" cfg4-profile $TAG > $O/cfg4_profile.log 2>&1
    python3 bench.py --force-ep --no-cpu-baseline --clock-seconds 0 --steps 20 --warmup 5 > $O/force_ep_line.json 2> $O/force_ep.err
    tail -2 $O/tiny_profile.log $O/cfg4_profile.log; tail -c 600 $O/force_ep_line.json ;;
  train-profile)   # the two training steps only (kernel tables): tools/diag.sh train-profile TAG
    TAG=$1; O=gpurun_out/prof_$TAG; mkdir -p $O
    python3 tools/train_bench.py model 128 10 > $O/train_unprofiled.log 2>&1        # the step's wall time WITHOUT the profiler
    python3 tools/train_bench.py model 128 10 resmoe_base_patch16_224_expert8_top1 > $O/train_resmoe_unprofiled.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o train -- python3 tools/train_bench.py model 128 6 > $O/train.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_resmoe -o train -- python3 tools/train_bench.py model 128 6 resmoe_base_patch16_224_expert8_top1 > $O/train_resmoe.log 2>&1
    grep "train step" $O/train_unprofiled.log $O/train_resmoe_unprofiled.log ;;
  tiny-profile)    # the reference's own models (DeiT-Tiny, E 8, top-2, batch 128): tools/diag.sh tiny-profile TAG
    TAG=$1; O=gpurun_out/prof_$TAG; mkdir -p $O
    for m in resmoe_tiny_patch16_224_expert8 moe_tiny_patch16_224_expert8; do
      python3 tools/tiny_bench.py $m 128 20 > $O/tiny_$m.json 2> $O/tiny_$m.txt
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/tiny_$m -o tiny -- python3 tools/tiny_bench.py $m 128 20 > $O/tiny_${m}_prof.log 2>&1
    done
    TRAIN_BENCH_GRAPH=1 python3 tools/train_bench.py model 128 10 resmoe_tiny_patch16_224_expert8 > $O/tiny_train_unprofiled.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/tiny_train -o train -- python3 tools/train_bench.py model 128 6 resmoe_tiny_patch16_224_expert8 > $O/tiny_train.log 2>&1
    grep -h "images/s" $O/tiny_*.txt $O/tiny_train_unprofiled.log ;;
  cfg4-profile)    # BASELINE cfg 4's model at its per-rank batch (512 / 8 = 64): tools/diag.sh cfg4-profile TAG
    TAG=$1; O=gpurun_out/prof_$TAG; mkdir -p $O
    python3 tools/cfg4_bench.py 64 > $O/cfg4_b64.txt 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4 -o cfg4 -- python3 tools/cfg4_bench.py 64 > $O/cfg4_prof.log 2>&1
    grep -v amdgpu $O/cfg4_b64.txt ;;
  *)
    sed -n 2,24p "$0"; exit 1 ;;
esac
