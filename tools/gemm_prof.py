#!/usr/bin/env python3
"""Single-variant grouped-GEMM driver for rocprofv3 runs and calibration.
usage: gemm_prof.py <variant> <shape: fc1|fc1n|qkv|fc2|fc2h|sq8k|sq4k|ffn> [iters] [images (default 256)]
(ffn = smoe_expert_ffn: both expert GEMMs of a layer, gathered rows / combine / residual as the model runs them, one launch)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops, _lib  # noqa: E402
if os.environ.get("SMOE_LIB"):
    _lib.LIB_PATH = os.environ["SMOE_LIB"]          # a diagnostic build (make DIAG=-DSMOE_DIAG): SMOE_PS_NBLOCK / SMOE_PS_GRID apply


def main():
    variant = int(sys.argv[1])
    shape = sys.argv[2]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = "cuda:0"
    torch.manual_seed(0)
    if shape == "ffn":
        T, d, h, E = (int(sys.argv[4]) if len(sys.argv) > 4 else 256) * 197, 768, 3072, 8
        idx = torch.randint(0, E, (T, 1), device=dev)
        counts, offsets, pos, inv_pos, _ = ops.dispatch_plan(idx, E)
        x16 = torch.randn(T, d, device=dev).half()
        w1, w2 = (torch.randn(E, h, d, device=dev) * 0.02).half(), (torch.randn(E, d, h, device=dev) * 0.02).half()
        b1, b2 = torch.randn(E, h, device=dev) * 0.02, torch.randn(E, d, device=dev) * 0.02
        score, res = torch.rand(T, device=dev), torch.randn(T, d, device=dev)
        hbuf, out = torch.empty(T, h, device=dev, dtype=torch.float16), torch.empty(T, d, device=dev)

        def run():
            ops.expert_ffn(x16, w1, b1, w2, b2, offsets, out, a_gather=pos, row_map=pos, row_scale=score, residual=res, H=hbuf)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            run()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        print(f"fused expert FFN T={T} d={d} h={h} E={E}: {ms:.4f} ms  {4.0*T*d*h/ms/1e9:.1f} TFLOP/s", flush=True)
        return
    if shape.startswith("sq"):
        n = 8192 if shape == "sq8k" else 4096
        E, M, K, N = 1, n, n, n
        counts = [M]
    else:
        E, M = 8, (int(sys.argv[4]) if len(sys.argv) > 4 else 256) * 197
        K, N = (768, 3072) if shape in ("fc1", "fc1n") else ((768, 2304) if shape == "qkv" else (3072, 768))
        if shape.startswith("k"):  # k<K>n<N>: plain f16-out GEMM of that depth / width (per-tile constant cost probes)
            K, N = (int(v) for v in shape[1:].split("n"))
        if shape == "qkv":
            E = 1
        base = M // E
        counts = [base] * E
        counts[-1] += M - base * E
    offsets = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=dev)
    data = os.environ.get("SMOE_DATA", "randn")  # clocks depend on the operand bits: compare like with like
    if data == "zeros":
        A, W = torch.zeros(M, K, device=dev).half(), torch.zeros(E, N, K, device=dev).half()
    elif data == "uniform":
        A = (torch.rand(M, K, device=dev) * 2 - 1).half()
        W = (torch.rand(E, N, K, device=dev) * 2 - 1).half()
    else:
        A = torch.randn(M, K, device=dev).half()
        W = (torch.randn(E, N, K, device=dev) * 0.02).half()
    b = torch.randn(E, N, device=dev) * 0.02
    epi = ops.EPI_GELU if shape == "fc1" else ops.EPI_NONE
    odt = torch.float32 if shape == "fc2" else torch.float16
    out = torch.empty(M, N, device=dev, dtype=odt)
    for _ in range(3):
        ops.grouped_gemm(A, W, b, offsets, epi, out=out, variant=variant)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.grouped_gemm(A, W, b, offsets, epi, out=out, variant=variant)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"variant {variant} shape {shape} M={M} K={K} N={N}: {ms:.4f} ms  {2.0*M*K*N/ms/1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
