#!/usr/bin/env python3
"""Images/s of the residual-MoE model (token-skip gates firing) on the fused HIP path vs the module-composed path.
usage: resmoe_bench.py [batch] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slim_switch_moe_vit_amd as sm  # noqa: E402
from slim_switch_moe_vit_amd import resmoe  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = "cuda:0"
    torch.manual_seed(0)
    model = sm.create_model("resmoe_base_patch16_224_expert8_top1", num_classes=1000).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for blk in model.blocks:
            m = blk.mlp
            m.gate.gate.weight.copy_(torch.randn(m.gate.gate.weight.shape, generator=g) * 0.02)
            m.experts.htoh4.weight.copy_(torch.nn.init.trunc_normal_(torch.empty_like(m.experts.htoh4.weight), std=0.02, generator=g))
            m.experts.h4toh.weight.copy_(torch.nn.init.trunc_normal_(torch.empty_like(m.experts.h4toh.weight), std=0.02, generator=g))
            for gt in (blk.dense_gate, blk.moe_gate):
                gt.head[1].weight.copy_(torch.randn(gt.head[1].weight.shape, generator=g) * 0.05)
                gt.head[1].bias.fill_(1.5)       # sigmoid(1.5 +- ...) vs threshold 0.9: a visible fraction of tokens skips
    model = model.to(dev)
    images = torch.randn(batch, 3, 224, 224, generator=g).to(dev)

    def run(fused: bool):
        orig = resmoe._fused_ok
        if not fused:
            resmoe._fused_ok = lambda blk, x: False
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                for _ in range(3):
                    model(images)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    model(images)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / steps
        finally:
            resmoe._fused_ok = orig
    tf = run(True)
    skipped = [(b.dense_gate._skipped_tokens / max(1, b.dense_gate._total_tokens),
                b.moe_gate._skipped_tokens / max(1, b.moe_gate._total_tokens)) for b in model.blocks]
    tc = run(False)
    print(f"resmoe_base_patch16_224_expert8_top1, batch {batch}: fused {tf * 1e3:.2f} ms/step = {batch / tf:.0f} images/s; "
          f"composed {tc * 1e3:.2f} ms/step = {batch / tc:.0f} images/s; skipped fraction per block (dense, moe): "
          f"{[(round(a, 2), round(b, 2)) for a, b in skipped[:4]]} ...")


if __name__ == "__main__":
    main()
