#!/usr/bin/env python3
"""fwd + bwd timing of ONE MoE layer at BASELINE cfg-5 per-GPU size (ViT-B/16, E=8, SwitchGate, capacity_factor 1.0,
aux loss; T = 256 x 197 tokens), single rank.  Prints one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slim_switch_moe_vit_amd as sm  # noqa: E402
from slim_switch_moe_vit_amd import _lib  # noqa: E402
if os.environ.get("SMOE_LIB"):
    _lib.LIB_PATH = os.environ["SMOE_LIB"]


def main():
    dev = "cuda:0"
    torch.manual_seed(0)
    d, h, E, T = 768, 3072, 8, 256 * 197
    mod = sm.FMoETransformerMLP(E, d, h, torch.nn.GELU(), top_k=1, gate="switch", capacity_factor=1.0).to(dev).train()
    with torch.no_grad():
        for p in mod.experts.parameters():
            if p.dim() == 3:
                p.copy_(torch.randn_like(p) * 0.02)
    x = torch.randn(T, d, device=dev, requires_grad=True)
    gout = torch.randn(T, d, device=dev) * 1e-2

    def step():
        out = mod(x)
        loss = (out * gout).sum() + 0.01 * mod.gate.get_loss()
        loss.backward()
        for p in mod.parameters():
            p.grad = None
        x.grad = None

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    s.record()
    for _ in range(n):
        step()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    kept = int(mod.last_plan[3][-1])
    flops = 3 * 4.0 * kept * d * h  # fwd (2 GEMMs) + dgrad (2) + wgrad (2)
    print(json.dumps({"what": "MoE layer fwd+bwd, cfg-5 size, 1 GPU", "ms": round(ms, 3), "kept_tokens": kept,
                      "dropped": T - kept, "gemm_tflops": round(flops / ms / 1e9, 1)}))


if __name__ == "__main__":
    main()
