#!/usr/bin/env python3
"""Attention kernel driver for timing / rocprofv3 (ViT-B shape: B 256, N 197, H 12, D 64, f16)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops, _lib
if os.environ.get("SMOE_LIB"):
    _lib.LIB_PATH = os.environ["SMOE_LIB"]
B, N, H, D = 256, 197, 12, 64
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
qkv = torch.randn(B, N, 3, H, D, device="cuda:0").half()
for _ in range(3): ops.attention(qkv, B, N, H, D, D ** -0.5)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): ops.attention(qkv, B, N, H, D, D ** -0.5)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print(f"attention B{B} N{N} H{H}: {ms*1e3:.1f} us  {4.0*B*H*N*N*D/ms/1e9:.1f} TFLOP/s  {(B*N*4*H*D*2)/ms/1e6:.0f} GB/s", flush=True)
# backward (the training step's kernel; B 128 as tools/train_bench.py runs it)
Bb = 128
qb = qkv[:Bb].contiguous()
o, lse = ops.attention(qb, Bb, N, H, D, D ** -0.5, want_lse=True)
do = torch.randn_like(o)
for _ in range(3): ops.attention_bwd(qb, o, do, lse, Bb, N, H, D, D ** -0.5)
torch.cuda.synchronize()
s.record()
for _ in range(iters): ops.attention_bwd(qb, o, do, lse, Bb, N, H, D, D ** -0.5)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print(f"attention backward B{Bb} N{N} H{H}: {ms*1e3:.1f} us  {10.0*Bb*H*N*N*D/ms/1e9:.1f} TFLOP/s", flush=True)
