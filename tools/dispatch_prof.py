#!/usr/bin/env python3
"""Driver for rocprofv3 --kernel-trace --stats of the dispatch-side kernels at BASELINE cfg-2 size (T = 50,432 tokens,
d = 768, E = 8): router (plain and LayerNorm-fused), dispatch plan, token scatter, gather + combine (k = 1 with the f32
residual, and k = 2), the token-skip gate passes.  usage: dispatch_prof.py [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = "cuda:0"
    torch.manual_seed(0)
    T, d, E = 256 * 197, 768, 8
    x = torch.randn(T, d, device=dev)
    wg = torch.randn(E, d, device=dev) * 0.02
    bg = torch.zeros(E, device=dev)
    g, b = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    gw, gb, thr = torch.randn(1, d, device=dev) * 0.05, torch.zeros(1, device=dev), torch.tensor(0.6, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    dl = torch.randn(T, E, device=dev)
    for _ in range(iters):
        idx, score, _, _ = ops.router_topk(x, wg, bg, 1)
        xn16, _, idx, score, _, _ = ops.ln_router_topk(x, g, b, 1e-6, wg, bg, 1)
        counts, offsets, pos, inv_pos, _ = ops.dispatch_plan(idx, E)
        buf = ops.scatter_rows(x, pos, 1, torch.float16)
        out = ops.gather_combine(buf, inv_pos, score, T, 1, torch.float32, residual=x)
        idx2, score2, _, _ = ops.router_topk(x, wg, bg, 2)
        c2, o2, pos2, inv2, _ = ops.dispatch_plan(idx2, E)
        buf2 = ops.scatter_rows(x, pos2, 2, torch.float16)
        out2 = ops.gather_combine(buf2, inv2, score2, T, 2, torch.float32)
        ops.gate_ln_router(x, gw, gb, thr, ln=(g, b, 1e-6), xn16_dtype=torch.float16, want_xn32=True, skip_count=cnt)
        ops.gate_ln_router(x, gw, gb, thr, ln=(g, b, 1e-6), wg=wg, bg=bg, k=1, xn16_dtype=torch.float16, want_xn32=True,
                           skip_count=cnt)
        ops.layernorm(x, g, b, 1e-6, torch.float16)
        # round 3: combine + residual + the next LayerNorm in one pass (expert-parallel return path), the padded plan of the
        # static exchange, the backward-side streaming kernels
        ops.gather_combine_ln(buf, inv_pos, score, T, 1, x, g, b, 1e-6, torch.float16)
        ops.dispatch_plan_padded(idx, E, T // E + 64)
        dy16 = buf if buf.shape[0] == T else buf[:T]
        ops.layernorm_bwd(x, dy16, g, 1e-6, dres=x)
        ops.gate_dgrad(dl, wg, torch.float32)
    torch.cuda.synchronize()
    print("done", float(out.sum()), float(out2.sum()))


if __name__ == "__main__":
    main()
