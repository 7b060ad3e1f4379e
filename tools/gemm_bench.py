#!/usr/bin/env python3
"""Micro-benchmark of the grouped expert GEMM variants at BASELINE cfg-2 shapes (interleaved rounds in one
process, random data; cdna_hip_programming.md rules 24/25).  Usage: python tools/gemm_bench.py [variants...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops  # noqa: E402


def main():
    variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3]
    dev = "cuda:0"
    torch.manual_seed(0)
    T, d, h, E = 256 * 197, 768, 3072, 8
    idx = torch.randint(0, E, (T, 1), device=dev)
    counts, offsets, pos, inv_pos, _ = ops.dispatch_plan(idx, E)
    x16 = torch.randn(T, d, device=dev).half()
    w1 = (torch.randn(E, h, d, device=dev) * 0.02).half()
    w2 = (torch.randn(E, d, h, device=dev) * 0.02).half()
    b1 = torch.randn(E, h, device=dev) * 0.02
    b2 = torch.randn(E, d, device=dev) * 0.02
    score = torch.rand(T, device=dev)
    hbuf = ops.grouped_gemm(x16, w1, b1, offsets, ops.EPI_GELU, torch.float16, variant=0)
    ref_h = hbuf.clone()
    out0 = torch.zeros(T, d, device=dev)
    ops.grouped_gemm(ref_h, w2, b2, offsets, ops.EPI_NONE, row_map=pos, row_scale=score, out=out0, variant=0)
    cases = {
        "fc1 K768 N3072 gelu f16out": lambda v, o: ops.grouped_gemm(x16, w1, b1, offsets, ops.EPI_GELU, torch.float16, variant=v, out=o),
        "fc2 K3072 N768 f32out rowmap": lambda v, o: ops.grouped_gemm(ref_h, w2, b2, offsets, ops.EPI_NONE, row_map=pos, row_scale=score, out=o, variant=v),
        "fc2 K3072 N768 f16out": lambda v, o: ops.grouped_gemm(ref_h, w2, b2, offsets, ops.EPI_NONE, torch.float16, variant=v, out=o),
    }
    outs = {"fc1 K768 N3072 gelu f16out": lambda: torch.empty(T, h, device=dev, dtype=torch.float16),
            "fc2 K3072 N768 f32out rowmap": lambda: torch.zeros(T, d, device=dev),
            "fc2 K3072 N768 f16out": lambda: torch.empty(T, d, device=dev, dtype=torch.float16)}
    flops = 2.0 * T * d * h
    res = {}
    for name, fn in cases.items():
        # correctness vs variant 0
        o_ref = outs[name](); fn(0, o_ref)
        for v in variants:
            o = outs[name](); fn(v, o)
            torch.cuda.synchronize()
            diff = (o.float() - o_ref.float()).abs().max().item()
            res[f"{name} | v{v} maxdiff_vs_v0"] = diff
        times = {v: [] for v in variants}
        o = outs[name]()
        for rnd in range(6):
            for v in variants:
                for _ in range(2):
                    fn(v, o)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    fn(v, o)
                e.record()
                torch.cuda.synchronize()
                times[v].append(s.elapsed_time(e) / 10)
        for v in variants:
            ts = sorted(times[v])
            res[f"{name} | v{v}"] = {"median_ms": round(ts[len(ts) // 2], 4), "min_ms": round(ts[0], 4),
                                     "tflops_median": round(flops / (ts[len(ts) // 2] * 1e-3) / 1e12, 1)}
    for k, v in res.items():
        print(k, json.dumps(v))


if __name__ == "__main__":
    main()
