#!/usr/bin/env python3
"""What the N > 256 fallback costs: torch's scaled_dot_product_attention forward + backward against the library's attention
forward / backward kernels (N <= 256 only for the backward) at the training shapes.

    python tools/attn_bwd_probe.py        prints one line per (B, N, H) x implementation: forward us, backward us, TFLOP/s"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops  # noqa: E402

DEV = "cuda:0"


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    torch.manual_seed(0)
    for B, N, H in ((128, 197, 12), (32, 577, 16), (64, 577, 16), (128, 197, 3)):
        hd = 64
        scale = hd ** -0.5
        qkv = (torch.randn(B, N, 3, H, hd, device=DEV) * 0.5).half()
        dout = (torch.randn(B, N, H * hd, device=DEV) * 0.1).half()
        f_flop, b_flop = 4.0 * B * H * N * N * hd, 10.0 * B * H * N * N * hd
        # torch: q, k, v as [B, H, N, hd] views of the same buffer (what vit.py's fallback does)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).detach().requires_grad_(True) for i in range(3))

        def t_fwd():
            return F.scaled_dot_product_attention(q, k, v, scale=scale)
        o = t_fwd()
        go = dout.view(B, N, H, hd).permute(0, 2, 1, 3)

        def t_bwd():
            return torch.autograd.grad(o, (q, k, v), go, retain_graph=True)
        tf, tb = timed(t_fwd), timed(t_bwd)
        print(f"B {B:4d} N {N:4d} H {H:3d}  torch sdpa   fwd {tf:8.1f} us ({f_flop / tf * 1e-6:6.1f} TF)   bwd {tb:8.1f} us ({b_flop / tb * 1e-6:6.1f} TF)",
              flush=True)
        flat = qkv.reshape(B * N, 3 * H * hd)
        own_bwd = ops.attention_bwd_supported(N, hd)          # (the forward keeps its log-sum-exp output for those N only)
        if own_bwd:
            out, lse = ops.attention(flat, B, N, H, hd, scale, want_lse=True)
        of = timed(lambda: ops.attention(flat, B, N, H, hd, scale, want_lse=own_bwd))
        if own_bwd:
            ob = timed(lambda: ops.attention_bwd(flat, out, dout, lse, B, N, H, hd, scale))
            print(f"B {B:4d} N {N:4d} H {H:3d}  own kernels  fwd {of:8.1f} us ({f_flop / of * 1e-6:6.1f} TF)   bwd {ob:8.1f} us ({b_flop / ob * 1e-6:6.1f} TF)",
                  flush=True)
        else:
            print(f"B {B:4d} N {N:4d} H {H:3d}  own kernels  fwd {of:8.1f} us ({f_flop / of * 1e-6:6.1f} TF)   bwd not supported", flush=True)


if __name__ == "__main__":
    main()
