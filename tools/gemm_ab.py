#!/usr/bin/env python3
"""A/B of grouped-GEMM variants on the shapes of the bench model (BASELINE cfg 2), interleaved rounds in one process on
random data (cdna_hip_programming.md rules 24 / 25).  Shapes: GEMM-1 as the model runs it (gathered rows, bias + GELU, f16
out), the qkv projection (one group, N 2304), GEMM-2 (row-mapped f32 store + residual).
usage: gemm_ab.py [variants ...]      default: 9 14 (direct-store epilogue vs LDS-staged epilogue)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops  # noqa: E402


def main():
    variants = [int(v) for v in sys.argv[1:]] or [9, 14]
    dev = "cuda:0"
    torch.manual_seed(0)
    T, d, h, E = 256 * 197, 768, 3072, 8
    idx = torch.randint(0, E, (T, 1), device=dev)
    counts, offsets, pos, inv_pos, _ = ops.dispatch_plan(idx, E)
    one = torch.tensor([0, T], dtype=torch.int32, device=dev)
    x16 = torch.randn(T, d, device=dev).half()
    w1 = (torch.randn(E, h, d, device=dev) * 0.02).half()
    w2 = (torch.randn(E, d, h, device=dev) * 0.02).half()
    wq = (torch.randn(1, 3 * d, d, device=dev) * 0.02).half()
    b1 = torch.randn(E, h, device=dev) * 0.02
    b2 = torch.randn(E, d, device=dev) * 0.02
    bq = torch.randn(1, 3 * d, device=dev) * 0.02
    score = torch.rand(T, device=dev)
    hbuf = torch.empty(T, h, device=dev, dtype=torch.float16)
    qkv = torch.empty(T, 3 * d, device=dev, dtype=torch.float16)
    out = torch.zeros(T, d, device=dev)
    res = torch.randn(T, d, device=dev)
    cases = {
        "gemm1 (K 768, N 3072, gathered rows, bias + GELU, f16)": (
            lambda v: ops.grouped_gemm(x16, w1, b1, offsets, ops.EPI_GELU, torch.float16, variant=v, a_gather=pos, out=hbuf),
            lambda: hbuf, 2.0 * T * d * h),
        "qkv (K 768, N 2304, one group, f16)": (
            lambda v: ops.grouped_gemm(x16, wq, bq, one, ops.EPI_NONE, torch.float16, variant=v, out=qkv), lambda: qkv,
            2.0 * T * d * 3 * d),
        "gemm2 (K 3072, N 768, row-mapped f32 + residual)": (
            lambda v: ops.grouped_gemm(hbuf, w2, b2, offsets, ops.EPI_NONE, row_map=pos, row_scale=score, out=out, variant=v,
                                       residual=res), lambda: out, 2.0 * T * d * h),
    }
    for name, (fn, get, flops) in cases.items():
        fn(4)
        ref = get().clone()
        line = {}
        for v in variants:
            get().zero_()
            fn(v)
            torch.cuda.synchronize()
            line[f"v{v} bit-equal to v4"] = bool(torch.equal(get(), ref))
        times = {v: [] for v in variants}
        for _ in range(100):          # let the clock settle under load
            fn(variants[0])
        for rnd in range(8):
            for v in variants:
                for _ in range(3):
                    fn(v)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    fn(v)
                e.record()
                torch.cuda.synchronize()
                times[v].append(s.elapsed_time(e) / 20)
        for v in variants:
            ts = sorted(times[v])
            med = ts[len(ts) // 2]
            line[f"v{v}"] = {"median_ms": round(med, 4), "min_ms": round(ts[0], 4), "tflops": round(flops / med / 1e9, 1)}
        print(name, json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
