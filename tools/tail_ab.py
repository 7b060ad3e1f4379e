#!/usr/bin/env python3
"""A/B inside ONE process: how a MoE half ends on a single rank (fmoe.TAIL_MODE).

    "epilogue"  GEMM-2's row-mapped f32 epilogue (combine + residual in the store), then the next block's norm1 as its own pass
    "ln"        GEMM-2 with contiguous 16-bit direct stores, then smoe_gather_combine_ln (combine + residual + next norm1)

Alternates the two modes R times over the bench model (ViT-B/16, E = 8, top-1, batch 256), S timed steps each, and prints the
per-mode median step time, the per-launch HIP-event averages of the kernels that differ, and the difference of the logits.

    python tools/tail_ab.py [rounds=4] [steps=10] [batch=256]
"""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import slim_switch_moe_vit_amd as sm  # noqa: E402
from slim_switch_moe_vit_amd import fmoe, ops  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    args = types.SimpleNamespace(experts=8, compute_dtype="f16", gemm_variant=None, ep_chunks=1, ep_micro_batches=1,
                                 compute_streams=1, force_ep=False, no_cpu_baseline=True, batch=batch)
    dev = torch.device("cuda", 0)
    model, _ = bench.build_model(args, 1, 0, dev)
    images = torch.randn(batch, 3, 224, 224, generator=torch.Generator().manual_seed(100)).to(dev)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model(images)

    res = {m: {"ms": [], "kern": {}} for m in ("epilogue", "ln")}
    logits = {}
    for r in range(rounds):
        for mode in ("epilogue", "ln"):
            fmoe.TAIL_MODE = mode
            for _ in range(3):
                out = step()
            torch.cuda.synchronize()
            logits[mode] = out.float()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            ev[0].record()
            for i in range(steps):
                step()
                ev[i + 1].record()
            torch.cuda.synchronize()
            res[mode]["ms"] += [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
            ops.profile_begin()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            for name, meta, ms in ops.profile_end():
                key = name
                if name == "grouped_gemm":
                    key = "gemm1" if meta.get("epilogue") == ops.EPI_GELU else "gemm2"
                a = res[mode]["kern"].setdefault(key, [0, 0.0])
                a[0] += 1
                a[1] += ms
    out = {}
    for mode, r in res.items():
        ms = sorted(r["ms"])
        out[mode] = {"ms_per_step_median": round(ms[len(ms) // 2], 4), "min": round(ms[0], 4), "max": round(ms[-1], 4),
                     "images_per_s": round(batch / ms[len(ms) // 2] * 1e3, 1),
                     "kernel_avg_us (HIP events, profiled steps)": {k: round(v[1] / v[0] * 1e3, 2) for k, v in sorted(r["kern"].items())},
                     "kernel_launches_per_step": {k: v[0] / (2 * rounds) for k, v in sorted(r["kern"].items())}}
    dl = (logits["ln"] - logits["epilogue"]).abs()
    out["logits_ln_vs_epilogue"] = {"max_abs": float(dl.max()), "per_image_median_max": float(dl.amax(dim=1).median()),
                                    "scale_max_abs": float(logits["epilogue"].abs().max())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
