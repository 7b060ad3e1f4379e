#!/bin/bash
# Smoke matrix of bench.py flag combinations on one GPU (three timed steps each): expert counts, forced expert parallelism with fixed
# pipeline shapes / exchanges / slot sizes / reserved CUs, odd batch sizes, two compute streams, bf16.  Prints rc and ms per step.
#   bash tools/bench_matrix.sh > gpurun_out/bench_matrix.txt
run() { echo "== $*"; timeout -k 10 240 python3 bench.py --no-cpu-baseline --clock-seconds 0 --steps 3 --warmup 2 "$@" > /tmp/line.json 2> /tmp/err.txt; rc=$?; echo "rc=$rc"; if [ $rc -ne 0 ]; then grep -v "amdgpu.ids\|socket.cpp" /tmp/err.txt | tail -4 | cut -c1-300; else python3 -c "
import json
d=json.loads(open('/tmp/line.json').read().strip().splitlines()[-1]); print('   ', d['ms_per_step'], d['config']['parallelism'][:110])"; fi; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; }
run --experts 16 --force-ep --ep-micro-batches 3 --ep-static 1
run --experts 16 --force-ep --ep-micro-batches 2 --ep-static 1 --ep-alpha 1.0
run --experts 32 --force-ep --ep-micro-batches 3 --ep-static 1
run --experts 8 --force-ep --ep-micro-batches 2 --ep-chunks 2 --ep-static 0
run --experts 8 --force-ep --ep-static 1 --ep-alpha 1.0 --ep-reserve-cus 16 --ep-micro-batches 2
run --batch 64 --force-ep
run --batch 31 --force-ep --ep-micro-batches 2 --ep-static 1
run --compute-streams 2
run --compute-dtype bf16 --force-ep
run --experts 4 --force-ep
