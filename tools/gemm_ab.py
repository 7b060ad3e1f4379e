#!/usr/bin/env python3
"""A/B of grouped-GEMM variants on the shapes of the bench model (BASELINE cfg 2), interleaved rounds in one process on
random data (cdna_hip_programming.md rules 24 / 25).  Shapes: GEMM-1 as the model runs it (gathered rows, bias + GELU, f16
out), the qkv projection (one group, N 2304), GEMM-2 (row-mapped f32 store + residual), the attention projection (one group, f32 + residual).
usage: gemm_ab.py [arms ...]      default: 9 14 (direct-store / buffer-addressed epilogues vs the flat LDS-staged epilogue)
An arm is  variant[:n_block] ; with a diagnostic build (make DIAG=-DSMOE_DIAG, SMOE_LIB=<its .so>) `:n_block` sets SMOE_PS_NBLOCK for
that arm's launches (0 = the strided tile order, else the n-block width of the XCD-contiguous order), e.g.  9:0 9:6 9:3
--cold: one timed launch at a time, each behind a 512-MB write (as inside the model: the operands were last touched a layer ago)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops, _lib  # noqa: E402
if os.environ.get("SMOE_LIB"):
    _lib.LIB_PATH = os.environ["SMOE_LIB"]


class Arm:
    def __init__(self, spec):
        self.spec = spec
        v, _, nb = spec.partition(":")
        self.variant, self.n_block = int(v), (nb if nb != "" else None)

    def __call__(self, fn):
        if self.n_block is None:
            os.environ.pop("SMOE_PS_NBLOCK", None)
        else:
            os.environ["SMOE_PS_NBLOCK"] = self.n_block
        fn(self.variant)

    def __repr__(self):
        return "v" + self.spec


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cold = "--cold" in sys.argv          # every timed launch behind a 512-MB write (operands leave the L2s and the Infinity Cache)
    variants = [Arm(v) for v in (args or ["9", "14"])]
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
    wp = (torch.randn(1, d, d, device=dev) * 0.02).half()
    bp = torch.randn(1, d, device=dev) * 0.02
    out_p = torch.zeros(T, d, device=dev)
    cases["proj (K 768, N 768, one group, f32 + residual)"] = (
        lambda v: ops.grouped_gemm(x16, wp, bp, one, ops.EPI_NONE, torch.float32, variant=v, out=out_p, residual=res),
        lambda: out_p, 2.0 * T * d * d)
    for name, (fn, get, flops) in cases.items():
        fn(4)
        ref = get().clone()
        line = {}
        for v in variants:
            get().zero_()
            v(fn)
            torch.cuda.synchronize()
            line[f"{v} bit-equal to v4"] = bool(torch.equal(get(), ref))
        times = {v: [] for v in variants}
        for _ in range(100):          # let the clock settle under load
            variants[0](fn)
        flush = torch.empty(128 << 20, device=dev) if cold else None
        for rnd in range(24 if cold else 0):
            for v in variants:
                flush.zero_()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                v(fn)
                e.record()
                torch.cuda.synchronize()
                times[v].append(s.elapsed_time(e))
        for rnd in range(0 if cold else 8):
            for v in variants:
                for _ in range(3):
                    v(fn)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    v(fn)
                e.record()
                torch.cuda.synchronize()
                times[v].append(s.elapsed_time(e) / 20)
        for v in variants:
            ts = sorted(times[v])
            med = ts[len(ts) // 2]
            line[f"{v}"] = {"median_ms": round(med, 4), "min_ms": round(ts[0], 4), "tflops": round(flops / med / 1e9, 1)}
        print(name, json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
