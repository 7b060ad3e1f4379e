#!/usr/bin/env python3
"""Where does the HOST spend its time in an expert-parallel forward?  One-rank RCCL group, the bench model at batch 256; for the
counted and the speculative static exchange at 1 and 2 micro-batches: host enqueue time per step with the overflow watch's lag taken
out of the way (ep.OVERFLOW_LAG huge: the host never waits for an event inside the loop), then a cProfile listing of the static
1 x 1 step.

    python tools/ep_host_profile.py [steps=20]            (SLIMMOE_EP_TRANSPORT=cabi: the library's own transport)"""
import cProfile
import io
import os
import pstats
import sys
import time
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from slim_switch_moe_vit_amd import ep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29572", rank=0, world_size=1, device_id=dev)
    args = types.SimpleNamespace(experts=8, compute_dtype="f16", gemm_variant=None, ep_chunks=1, ep_micro_batches=1,
                                 compute_streams=1, force_ep=True, no_cpu_baseline=True, batch=256)
    model, _ = bench.build_model(args, 1, 0, dev)
    images = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(100)).to(dev)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model(images)

    def measure(tag, prof=None):
        for _ in range(3):
            ep.run_guarded(step)
        torch.cuda.synchronize()
        lag, ep.OVERFLOW_LAG = ep.OVERFLOW_LAG, 1 << 30
        if prof is not None:
            prof.enable()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        t1 = time.perf_counter()
        if prof is not None:
            prof.disable()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ep.OVERFLOW_LAG = lag
        try:
            ep.check_static_overflow(flush=True)
        except ep.StaticExchangeOverflow:
            tag += " (OVERFLOWED)"
        print(f"{tag:44s} host enqueue {1e3 * (t1 - t0) / steps:7.3f} ms/step   wall {1e3 * (t2 - t0) / steps:7.3f} ms/step", flush=True)

    for static in (False, True):
        ep.set_speculative(model, 1.25 if static else None)
        for mb in (1, 2):
            model.ep_micro_batches = mb
            measure(f"{'static' if static else 'counted'} exchange, {mb} micro-batch(es)")
    model.ep_micro_batches = 1
    prof = cProfile.Profile()
    measure("static exchange, 1 micro-batch, cProfile on", prof)
    out = io.StringIO()
    pstats.Stats(prof, stream=out).sort_stats("tottime").print_stats(28)
    print(out.getvalue())
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
