#!/usr/bin/env python3
"""Fused LayerNorm + router driver for timing / rocprofv3 (ViT-B shape: T = 256 x 197, d 768, E 8, k 1).
usage: lnrouter_prof.py [iters] [images] [d] [E] [tokens per image]; SMOE_LIB=<path> selects an alternative build of the library."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slim_switch_moe_vit_amd  # noqa: F401
from slim_switch_moe_vit_amd import ops, _lib
if os.environ.get("SMOE_LIB"):
    _lib.LIB_PATH = os.environ["SMOE_LIB"]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
images = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
E = int(sys.argv[4]) if len(sys.argv) > 4 else 8
T = images * (int(sys.argv[5]) if len(sys.argv) > 5 else 197)
g = torch.Generator().manual_seed(0)
x = torch.randn(T, d, generator=g).cuda()
gamma = (1 + 0.1 * torch.randn(d, generator=g)).cuda(); beta = (0.1 * torch.randn(d, generator=g)).cuda()
wg = (torch.randn(E, d, generator=g) * 0.02).cuda(); bg = torch.zeros(E).cuda()
for _ in range(3): ops.ln_router_topk(x, gamma, beta, 1e-6, wg, bg, 1, ops.GATE_NAIVE, xn16_dtype=torch.float16)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters): ops.ln_router_topk(x, gamma, beta, 1e-6, wg, bg, 1, ops.GATE_NAIVE, xn16_dtype=torch.float16)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
print(f"ln_router T{T} d{d} E{E} [{os.path.basename(_lib.LIB_PATH)}]: {ms*1e3:.1f} us  {(T*d*6)/ms/1e6:.0f} GB/s", flush=True)
