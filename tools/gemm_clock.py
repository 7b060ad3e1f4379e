#!/usr/bin/env python3
"""Clock the chip holds under the two expert GEMMs of the bench workload (MI355X_MICROARCH.md 'DVFS give-back' item 6).

Runs in a process of its own on the clock-probe build of the library (`make -C slim-switch-moe-vit_amd/csrc clock` ->
libslimmoe_hip_clock.so: two s_memtime / s_memrealtime stamps per workgroup of the persistent GEMM, nothing else differs):
>= SECONDS of back-to-back launches of each GEMM on random operands, then the stamps of the last launch:
clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups.  Prints ONE JSON line:

    {"gemm1_mhz": ..., "gemm2_mhz": ..., "fused_mhz": ..., "gemm1_ms": ..., "gemm2_ms": ..., "fused_ms": ..., "seconds_each": ..., ...}
(fused = smoe_expert_ffn, both GEMMs in one persistent launch)

bench.py starts it as a child process (SLIMMOE_LIB selects the library at import) and reports `frac_of_clocked_peak` from it.
usage: SLIMMOE_LIB=slim-switch-moe-vit_amd/libslimmoe_hip_clock.so python3 tools/gemm_clock.py [seconds_each=2.0] [batch=256]"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slim_switch_moe_vit_amd import ops, _lib  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = "cuda:0"
    torch.manual_seed(0)
    T, d, h, E = batch * 197, 768, 3072, 8
    lib = _lib.load()
    try:
        rd = lib.__getattr__("smoe_clock_read_stamps")
    except AttributeError:
        sys.exit("gemm_clock.py needs the clock-probe build: SLIMMOE_LIB=.../libslimmoe_hip_clock.so (make clock)")
    rd.restype = ctypes.c_int
    rd.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    want, got = _lib.source_build_id(" -DSMOE_CLOCK"), _lib.binary_build_id()
    if want is not None and got != want:     # a probe library left over from other sources would clock another kernel
        sys.exit(f"gemm_clock.py: the clock-probe library reports build id {got}, these sources with -DSMOE_CLOCK hash to {want}: "
                 "rebuild it (make -C slim-switch-moe-vit_amd/csrc clock)")
    idx = torch.randint(0, E, (T, 1), device=dev)
    counts, offsets, pos, inv_pos, _ = ops.dispatch_plan(idx, E)
    x16 = torch.randn(T, d, device=dev).half()                     # random operands: zeros would clock ~20 % higher
    w1 = (torch.randn(E, h, d, device=dev) * 0.02).half()
    w2 = (torch.randn(E, d, h, device=dev) * 0.02).half()
    b1 = torch.randn(E, h, device=dev) * 0.02
    b2 = torch.randn(E, d, device=dev) * 0.02
    score = torch.rand(T, device=dev)
    hbuf = torch.empty(T, h, device=dev, dtype=torch.float16)
    out = torch.zeros(T, d, device=dev)
    res = torch.randn(T, d, device=dev)

    def gemm1():
        ops.grouped_gemm(x16, w1, b1, offsets, ops.EPI_GELU, torch.float16, variant=9, a_gather=pos, out=hbuf)

    def gemm2():
        ops.grouped_gemm(hbuf, w2, b2, offsets, ops.EPI_NONE, row_map=pos, row_scale=score, out=out, variant=9, residual=res)

    def fused():
        ops.expert_ffn(x16, w1, b1, w2, b2, offsets, out, a_gather=pos, row_map=pos, row_scale=score, residual=res, H=hbuf)

    result = {"seconds_each": seconds, "batch": batch}
    gemm1()
    torch.cuda.synchronize()
    legs = [("gemm1", gemm1), ("gemm2", gemm2)]
    if ops.ffn_fused_available():       # (the fused launch is in optional builds only: make FFN=-DSMOE_FFN_FUSED)
        legs.append(("fused", fused))
    for name, fn in legs:
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:                  # back to back: the clock settles under THIS kernel's load
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if name == "fused":                                        # per-workgroup counts of ONE fused launch
            clr = lib.__getattr__("smoe_clock_clear_stamps"); clr.restype = ctypes.c_int; clr.argtypes = []
            clr()
            fn()
            torch.cuda.synchronize()
        buf = np.zeros(1024 * 8, dtype=np.uint64)
        assert rd(buf.ctypes.data, buf.size) == 0
        st = buf.reshape(1024, 8).astype(np.int64)
        ok = (st[:, 0] > 0) & (st[:, 2] > st[:, 0]) & (st[:, 3] > st[:, 1])
        st = st[ok]
        mhz = (st[:, 2] - st[:, 0]) / (st[:, 3] - st[:, 1]) * 100.0
        result[f"{name}_mhz"] = round(float(np.median(mhz)), 1)
        result[f"{name}_mhz_p10_p90"] = [round(float(np.percentile(mhz, 10)), 1), round(float(np.percentile(mhz, 90)), 1)]
        result[f"{name}_ms"] = round(e0.elapsed_time(e1) / 20, 4)
        result[f"{name}_kernel_cycles_median"] = int(np.median(st[:, 2] - st[:, 0]))
        result["workgroups"] = int(ok.sum())
        if name == "fused":
            result["fused_spins_per_wg_median_max"] = [int(np.median(st[:, 4])), int(st[:, 4].max())]
            result["fused_gemm1_tiles_per_wg_min_max"] = [int(st[:, 5].min()), int(st[:, 5].max())]
            result["fused_gemm2_tiles_per_wg_hist"] = np.bincount(st[:, 6]).tolist()
            g2 = st[:, 7] / np.maximum(st[:, 6], 1)
            g1 = (st[:, 2] - st[:, 0] - st[:, 7]) / np.maximum(st[:, 5], 1)
            result["fused_cycles_per_gemm1_tile_p10_p50_p90"] = [int(np.percentile(g1, q)) for q in (10, 50, 90)]
            result["fused_cycles_per_gemm2_tile_p10_p50_p90"] = [int(np.percentile(g2, q)) for q in (10, 50, 90)]
            result["fused_wg_cycles_p10_p50_p90_max"] = [int(np.percentile(st[:, 2] - st[:, 0], q)) for q in (10, 50, 90, 100)]
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
