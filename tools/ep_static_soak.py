#!/usr/bin/env python3
"""How often does the speculative static exchange have to repeat a step when EVERY batch is different?  One-rank RCCL group on one
GPU, `moe_base_patch16_224_expert8_top1` (depth 4) and the reference's `resmoe_tiny_patch16_224_expert8`, fresh random images per step;
per step the static result is compared with the counted exchange's (bit for bit) and the repeats / the slot rows are recorded.

    python tools/ep_static_soak.py [steps=40] [batch=64] [graph]      graph: the static forward replayed from engine.GraphedForward's
                                                                      HIP graph (re-captured whenever the slots are re-sized)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slim_switch_moe_vit_amd as sm  # noqa: E402
from slim_switch_moe_vit_amd import ep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29581", rank=0, world_size=1, device_id=dev)
    for name, kw in (("moe_base_patch16_224_expert8_top1", dict(depth=4)),
                     ("resmoe_tiny_patch16_224_expert8", dict(starting_threshold=0.55, target_threshold=0.5))):
        torch.manual_seed(0)
        model = sm.create_model(name, num_classes=100, **kw).eval()
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for blk in model.blocks:
                m = blk.mlp
                m.gate.gate.weight.copy_(torch.randn(m.gate.gate.weight.shape, generator=g) * 0.02)
                m.experts.htoh4.weight.normal_(0, 0.02, generator=g)
                m.experts.h4toh.weight.normal_(0, 0.02, generator=g)
                for gt in (getattr(blk, "dense_gate", None), getattr(blk, "moe_gate", None)):
                    if gt is not None:
                        gt.head[1].weight.normal_(0, 0.05, generator=g)
        model = model.to(dev)
        for blk in model.blocks:
            blk.mlp.force_ep = True
        model.ep_micro_batches = 1
        repeats, mism = 0, 0
        ratios = []
        ep.set_speculative(model, 1.25)
        gf = sm.GraphedForward(model) if graph else None
        assert gf is None or sm.GraphedForward.supported(model, dev)
        for s in range(steps):
            images = torch.randn(batch, 3, 224, 224, generator=g).to(dev)

            def fwd():
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                    return model(images).float()
            with ep.dynamic_only():
                ref = fwd()
            out, again = ep.run_guarded(fwd if gf is None else (lambda: gf(images).float()))
            repeats += int(again)
            mism += int(not torch.equal(out, ref))
            T = batch * 197 * model.blocks[0].mlp.top_k
            ratios.append(max(b.mlp.__dict__["_ep_slots"][1].table.rows / T for b in model.blocks))
        print(f"{name}: {steps} steps of {batch} fresh images: {repeats} repeated on the counted exchange, {mism} results differ from "
              f"the counted exchange's; send-buffer rows / routed rows (largest layer) first {ratios[0]:.3f}, last {ratios[-1]:.3f}, "
              f"max {max(ratios):.3f}" + (f"; HIP graph: {gf.captures} captures, failed = {gf.failed}" if gf is not None else ""), flush=True)
        gf = None
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
