#!/usr/bin/env python3
"""Where the GPU idles in the expert-parallel code path (one GPU, world of one rank: `bench.py --force-ep`'s model).

Runs a few eval forwards under the torch profiler (CPU + GPU activities), orders the GPU kernels of ONE step by start time and
reports: wall time of the step, GPU busy time, and the idle gaps > 15 us aggregated by (kernel before -> kernel after) -- the
places where the launch queue ran dry (host syncs of the count exchange, host work between launches).

usage: python3 tools/ep_gaps.py [micro_batches=1] [batch=256]
       EP_GAPS_STATIC=1.25 python3 tools/ep_gaps.py ...     the speculative static exchange (alpha; raised in warm-up as needed)"""
import argparse
import os
import sys
from collections import defaultdict

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    force = os.environ.get("EP_GAPS_PLAIN", "0") != "1"
    args = argparse.Namespace(experts=8, compute_dtype="f16", gemm_variant=9, ep_chunks=1, force_ep=force, ep_micro_batches=mb,
                              compute_streams=1, no_cpu_baseline=True, batch=batch)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if force:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1, device_id=dev)
    model, _ = bench.build_model(args, 1, 0, dev)
    images = torch.randn(batch, 3, 224, 224, generator=torch.Generator().manual_seed(100)).to(dev)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model(images)
    if os.environ.get("EP_GAPS_RESERVE"):
        from slim_switch_moe_vit_amd import ops
        ops.set_reserved_cus(int(os.environ["EP_GAPS_RESERVE"]))
        print(f"the persistent GEMM leaves {os.environ['EP_GAPS_RESERVE']} CUs free")
    alpha = os.environ.get("EP_GAPS_STATIC")
    if force and alpha:
        from slim_switch_moe_vit_amd import ep
        ep.set_speculative(model, float(alpha))
        reps = sum(int(ep.run_guarded(step)[1]) for _ in range(5))
        print(f"speculative static exchange: alpha {[round(float(b.mlp.ep_speculative), 3) for b in model.blocks]}, "
              f"{reps} warm-up step(s) repeated on the counted exchange")
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    ks = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range is not None),
                key=lambda e: e.time_range.start)
    ks = [e for e in ks if e.time_range.end > e.time_range.start]
    t0, t1 = ks[0].time_range.start, max(e.time_range.end for e in ks)
    busy, gaps, cur_end = 0.0, defaultdict(lambda: [0, 0.0]), ks[0].time_range.start
    prev = None
    for e in ks:
        s, en = e.time_range.start, e.time_range.end
        if s > cur_end:
            g = s - cur_end
            if g > 15 and prev is not None:
                key = (prev.name[:48], e.name[:48])
                gaps[key][0] += 1
                gaps[key][1] += g
        busy += max(0.0, en - max(s, cur_end))
        if en > cur_end:
            cur_end, prev = en, e
    print(f"micro-batches {mb}, batch {batch}, force_ep {force}: step wall (first kernel start -> last kernel end) {(t1 - t0) / 1e3:.3f} ms, "
          f"GPU busy {busy / 1e3:.3f} ms, idle {(t1 - t0 - busy) / 1e3:.3f} ms, {len(ks)} GPU activities")
    print("idle gaps > 15 us by (kernel before -> kernel after): count, total us")
    for key, (n, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {n:4d} {tot:9.1f}   {key[0]}  ->  {key[1]}")
    per = defaultdict(lambda: [0, 0.0])
    for e in ks:
        per[e.name[:72]][0] += 1
        per[e.name[:72]][1] += e.time_range.end - e.time_range.start
    print(f"sum of kernel durations {sum(e.time_range.end - e.time_range.start for e in ks) / 2 / 1e3:.3f} ms (nccl kernels are listed twice; "
          f"sum > busy = kernels of different streams ran side by side)")
    print("GPU time by kernel: count, total us, avg us")
    for name, (n, tot) in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"  {n:4d} {tot:9.1f} {tot / n:8.1f}   {name}")
    if force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
