#!/usr/bin/env python3
"""Is the forced expert-parallel step bound by the HOST?  For the bench model on one GPU (world of one rank) and each exchange
(counted / speculative static) prints the time the host needs to ENQUEUE a step (no synchronisation inside the loop) beside the
wall time per step (synchronised at the end), and the same for the plain single-rank path.

    python tools/ep_host_probe.py [steps=20]          (SLIMMOE_EP_TRANSPORT=cabi: the library's own RCCL transport)"""
import os
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
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29571", rank=0, world_size=1, device_id=dev)
    args = types.SimpleNamespace(experts=8, compute_dtype="f16", gemm_variant=None, ep_chunks=1, ep_micro_batches=1,
                                 compute_streams=1, force_ep=True, no_cpu_baseline=True, batch=256)
    model, _ = bench.build_model(args, 1, 0, dev)
    images = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(100)).to(dev)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model(images)

    def measure(tag):
        step = cell[0]
        for _ in range(3):
            ep.run_guarded(step)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        try:
            ep.check_static_overflow(flush=True)
        except ep.StaticExchangeOverflow:
            tag += " (OVERFLOWED)"
        print(f"{tag:34s} host enqueue {1e3 * (t1 - t0) / steps:7.3f} ms/step   wall {1e3 * (t2 - t0) / steps:7.3f} ms/step", flush=True)

    cell = [step]
    for blk in model.blocks:
        blk.mlp.force_ep = False
    measure("single-rank path")
    for blk in model.blocks:
        blk.mlp.force_ep = True
    ep.set_speculative(model, None)
    measure("EP, counted exchange")
    ep.set_speculative(model, 1.25)
    measure("EP, speculative static exchange")
    measure("EP, speculative static (again)")
    print("slot rows per layer / routed rows:", [round(b.mlp.__dict__["_ep_slots"][1].table.rows / (256 * 197), 3) for b in model.blocks])
    # the same static forward replayed from ONE HIP graph (engine.GraphedForward: the exchanges sit on the compute stream)
    import slim_switch_moe_vit_amd as sm
    if sm.GraphedForward.supported(model, dev):
        gf = sm.GraphedForward(model)
        eager = step().float()

        cell[0] = lambda: gf(images)
        measure("EP, static, HIP graph replay")
        measure("EP, static, HIP graph (again)")
        print("graph captures:", gf.captures, " failed:", gf.failed, " replay == eager:", bool(torch.equal(gf(images).float(), eager)))
        gf = None
    for blk in model.blocks:
        blk.mlp.force_ep = False
    if sm.GraphedForward.supported(model, dev):
        gf = sm.GraphedForward(model)

        cell[0] = lambda: gf(images)
        measure("single-rank path, HIP graph")
        gf = None
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
