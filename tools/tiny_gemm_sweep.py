#!/usr/bin/env python3
"""Which grouped-GEMM variant serves the DeiT-Tiny shapes (d 192 / h 768, batch 128, top-2: models/resMoE.py:151-187) best?
Interleaved rounds in one process, random data.  Variants: 1 = 128 x 128 tiles, 2 = 256 x 128, 3 = 256 x 256 (one workgroup per
tile, LDS-DMA), 4 = 256 / 320 x 256 ping-pong (one workgroup per tile), 9 = the same persistent, 10 / 11 = persistent 320 / 256 rows.

    python tools/tiny_gemm_sweep.py [batch=128]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = "cuda:0"
    torch.manual_seed(0)
    T, d, h, E, k = batch * 197, 192, 768, 8, 2
    n = T * k
    idx = torch.stack([torch.randperm(E, device=dev)[:k] for _ in range(64)]).repeat(T // 64 + 1, 1)[:T].contiguous()
    counts, offsets, pos, inv_pos, _ = ops.dispatch_plan(idx, E)
    one = torch.tensor([0, T], dtype=torch.int32, device=dev)
    xn16 = torch.randn(T, d, device=dev).half()
    buf = ops.scatter_rows(xn16, pos, k, torch.float16)
    wqkv = (torch.randn(1, 3 * d, d, device=dev) * 0.05).half(); bqkv = torch.randn(1, 3 * d, device=dev) * 0.02
    wproj = (torch.randn(1, d, d, device=dev) * 0.05).half(); bproj = torch.randn(1, d, device=dev) * 0.02
    w1 = (torch.randn(E, h, d, device=dev) * 0.05).half(); b1 = torch.randn(E, h, device=dev) * 0.02
    w2 = (torch.randn(E, d, h, device=dev) * 0.05).half(); b2 = torch.randn(E, d, device=dev) * 0.02
    res32 = torch.randn(T, d, device=dev)
    hbuf = ops.grouped_gemm(buf, w1, b1, offsets, ops.EPI_GELU, torch.float16, variant=9)
    cases = {
        "qkv   M %d K 192 N 576 f16 out" % T: (lambda v: ops.grouped_gemm(xn16, wqkv, bqkv, one, ops.EPI_NONE, torch.float16, variant=v),
                                               2.0 * T * d * 3 * d, T * d * 2 + T * 3 * d * 2, (1, 2, 3, 4, 9, 10, 11)),
        "proj  M %d K 192 N 192 f32 out + residual" % T: (lambda v: ops.grouped_gemm(xn16, wproj, bproj, one, ops.EPI_NONE, torch.float32, residual=res32, variant=v),
                                                          2.0 * T * d * d, T * d * (2 + 4 + 4), (1, 2, 3, 4, 9, 10, 11)),
        "fc1   M %d K 192 N 768 GELU f16 out (gathered rows)" % n: (lambda v: ops.grouped_gemm(xn16, w1, b1, offsets, ops.EPI_GELU, torch.float16, variant=v, a_gather=pos, a_div=k),
                                                                    2.0 * n * d * h, n * d * 2 + n * h * 2, (4, 9, 10, 11)),
        "fc1   M %d K 192 N 768 GELU f16 out (scattered buffer)" % n: (lambda v: ops.grouped_gemm(buf, w1, b1, offsets, ops.EPI_GELU, torch.float16, variant=v),
                                                                       2.0 * n * d * h, n * d * 2 + n * h * 2, (1, 2, 3, 4, 9, 10, 11)),
        "fc2   M %d K 768 N 192 f16 out" % n: (lambda v: ops.grouped_gemm(hbuf, w2, b2, offsets, ops.EPI_NONE, torch.float16, variant=v),
                                               2.0 * n * d * h, n * h * 2 + n * d * 2, (1, 2, 3, 4, 9, 10, 11)),
    }
    out = {}
    for name, (fn, flops, nbytes, variants) in cases.items():
        ref = fn(variants[-1]).float()
        times = {v: [] for v in variants}
        errs = {}
        for v in variants:
            errs[v] = float((fn(v).float() - ref).abs().max())
        for rnd in range(5):
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
        row = {}
        for v in variants:
            ts = sorted(times[v])
            us = 1e3 * ts[len(ts) // 2]
            row[f"v{v}"] = {"us": round(us, 1), "tflops": round(flops / us / 1e6, 1), "tb_s": round(nbytes / us / 1e6, 2), "maxdiff": errs[v]}
        out[name] = {"hbm_floor_us_at_6.3TBs": round(nbytes / 6.3e6, 1), "variants": row}
        print(name, f"(HBM floor {nbytes / 6.3e6:.1f} us):", "  ".join(f"v{v} {row[f'v{v}']['us']:.1f}" for v in variants), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
