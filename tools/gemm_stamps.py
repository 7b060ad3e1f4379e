#!/usr/bin/env python3
"""Diagnostic (library built with `make DIAG=-DSMOE_DIAG`): per-tile phase times of the persistent grouped GEMM from
s_memtime stamps (wave 0 / lane 0 of every workgroup).  usage: gemm_stamps.py <variant 9..13> <fc1|fc2|qkv>"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops, _lib  # noqa: E402
if os.environ.get("SMOE_LIB"):
    _lib.LIB_PATH = os.environ["SMOE_LIB"]          # a diagnostic build (make DIAG=-DSMOE_DIAG) next to the production library


def main():
    variant, shape = int(sys.argv[1]), sys.argv[2]
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
    hbuf = ops.grouped_gemm(x16, w1, b1, offsets, ops.EPI_GELU, torch.float16, variant=4, a_gather=pos)
    out = torch.zeros(T, d, device=dev)
    res = torch.randn(T, d, device=dev)

    one = torch.tensor([0, T], dtype=torch.int32, device=dev)
    wq = (torch.randn(1, 3 * d, d, device=dev) * 0.02).half()
    bq = torch.randn(1, 3 * d, device=dev) * 0.02
    qkv = torch.empty(T, 3 * d, device=dev, dtype=torch.float16)

    def run():
        if shape == "qkv":      # one group, N 2304, f16 out: the 256-row direct-store instantiation
            ops.grouped_gemm(x16, wq, bq, one, ops.EPI_NONE, torch.float16, variant=variant, out=qkv)
        elif shape == "ffn":     # smoe_expert_ffn (fused launch): the stamps left are those of its GEMM-1 runs' tiles 2.. (the GEMM-2 runs
            #                      overwrite tiles 0-1 of every workgroup)
            ops.expert_ffn(x16, w1, b1, w2, b2, offsets, out, a_gather=pos, row_map=pos, row_scale=score, residual=res, H=hbuf)
        elif shape == "fc1":
            ops.grouped_gemm(x16, w1, b1, offsets, ops.EPI_NONE if os.environ.get("SMOE_EPI") == "none" else ops.EPI_GELU,
                             torch.float16, variant=variant, a_gather=pos, out=hbuf)
        else:
            ops.grouped_gemm(hbuf, w2, b2, offsets, ops.EPI_NONE, row_map=pos, row_scale=score, out=out, variant=variant,
                             residual=res)
    lib = _lib.load()
    rd = lib.__getattr__("smoe_diag_read_stamps"); rd.restype = ctypes.c_int; rd.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    clr = lib.__getattr__("smoe_diag_clear_stamps"); clr.restype = ctypes.c_int; clr.argtypes = []
    if os.environ.get("SMOE_DIAG_FLAGS"):   # bit 0: no operand DMA in the main loop (MFMA + LDS reads alone; results are garbage)
        sf = lib.__getattr__("smoe_diag_set_flags"); sf.restype = ctypes.c_int; sf.argtypes = [ctypes.c_int]
        assert sf(int(os.environ["SMOE_DIAG_FLAGS"])) == 0
        print("SMOE_DIAG_FLAGS =", os.environ["SMOE_DIAG_FLAGS"])
    for _ in range(200):   # let the clock settle under load
        run()
    torch.cuda.synchronize()
    clr()
    run()
    torch.cuda.synchronize()
    n = 256 * 12 * 16
    buf = np.zeros(n, dtype=np.uint64)
    rc = rd(buf.ctypes.data, n)
    assert rc == 0, rc
    st = buf.reshape(256, 12, 16).astype(np.int64)
    names = {(0, 1): "main loop", (1, 2): "advance", (2, 3): "setup (gather addresses)", (3, 4): "issue kt0",
             (4, 5): "pass0: rows + GELU + LDS write", (5, 6): "pass0: wait + barrier", (6, 7): "pass0: LDS read + stores",
             (8, 9): "last pass: vmcnt(0)", (9, 10): "last pass: barrier", (10, 11): "last pass: stores", (11, 12): "last barrier",
             (0, 12): "whole tile", (1, 12): "tile boundary (everything but the main loop)"}
    if shape in ("fc1", "qkv", "ffn") and variant != 14:   # the direct-store epilogue (16-bit output, no row map): stamps 5 and 12 only
        names = {(0, 1): "main loop", (1, 2): "advance", (2, 3): "setup (gather addresses)", (3, 4): "issue kt0 + kt1",
                 (4, 5): "direct epilogue: bias + GELU + pack + swaps + 20 stores issued", (5, 12): "wait for K-tile 0 + barrier",
                 (0, 12): "whole tile", (1, 12): "tile boundary (everything but the main loop)"}
    sel = st[:, (2 if shape == "ffn" else 1):6, :]          # tiles 1..5 of every workgroup (steady state, a next tile exists)
    ok = (sel[:, :, 12] > 0) & (sel[:, :, 0] > 0)
    print(f"variant {variant} {shape} grid={os.environ.get('SMOE_PS_GRID', 'all')} epi={os.environ.get('SMOE_EPI', 'default')}: {int(ok.sum())} tiles sampled; s_memtime = shader cycles (k = 1000 cycles)")
    for (a, b), nm in names.items():
        dlt = (sel[:, :, b] - sel[:, :, a])[ok & (sel[:, :, a] > 0) & (sel[:, :, b] > 0)]
        if dlt.size:
            print(f"  {nm:46s} median {np.median(dlt) / 1000:7.2f} k   p10 {np.percentile(dlt, 10) / 1000:7.2f}   p90 {np.percentile(dlt, 90) / 1000:7.2f}")
    t0 = st[:, 0, 0][st[:, 0, 0] > 0]
    tend = st[:, :, 12].max(axis=1)
    print(f"  tiles per workgroup (histogram): {np.bincount((st[:, :, 12] > 0).sum(axis=1))}")


if __name__ == "__main__":
    main()
