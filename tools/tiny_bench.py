#!/usr/bin/env python3
"""The reference's OWN models (models/resMoE.py:151-209: DeiT-Tiny, d 192 / h 768, E = 8, top-2, batch 128 as cmd.sh:7-13 runs them)
on the eval forward: images/s and a per-kernel table (HIP events around every launch of a few extra steps: launches per step,
average us, TFLOP/s or GB/s, fraction of the 2.5 PFLOP/s MFMA / 6.3 TB/s achievable-HBM roof).

    python tools/tiny_bench.py [model=resmoe_tiny_patch16_224_expert8] [batch=128] [steps=20]
    rocprofv3 --kernel-trace --stats -d ... -- python3 tools/tiny_bench.py ...      (the same table by kernel symbol)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slim_switch_moe_vit_amd as sm  # noqa: E402
from slim_switch_moe_vit_amd import ops  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "resmoe_tiny_patch16_224_expert8"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    dev = "cuda:0"
    torch.manual_seed(0)
    kw = dict(num_classes=1000)
    if name.startswith("resmoe"):
        kw.update(starting_threshold=0.55, target_threshold=0.5)     # gates that fire (the defaults 1.0 / 0.9 never skip at init)
    model = sm.create_model(name, **kw).eval()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for blk in model.blocks:
            m = blk.mlp
            m.gate.gate.weight.copy_(torch.randn(m.gate.gate.weight.shape, generator=g) * 0.02)
            m.experts.htoh4.weight.copy_(torch.nn.init.trunc_normal_(torch.empty_like(m.experts.htoh4.weight), std=0.02, generator=g))
            m.experts.h4toh.weight.copy_(torch.nn.init.trunc_normal_(torch.empty_like(m.experts.h4toh.weight), std=0.02, generator=g))
            for gt in (getattr(blk, "dense_gate", None), getattr(blk, "moe_gate", None)):
                if gt is not None:
                    gt.head[1].weight.copy_(torch.randn(gt.head[1].weight.shape, generator=g) * 0.05)
        torch.nn.init.trunc_normal_(model.head.weight, std=0.02, generator=g)
    model = model.to(dev)
    images = torch.randn(batch, 3, 224, 224, generator=g).to(dev)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model(images)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    med = ms[len(ms) // 2]
    side = 3
    ops.profile_begin()
    for _ in range(side):
        step()
    torch.cuda.synchronize()
    agg = {}
    for nm, meta, t in ops.profile_end():
        key = nm
        if nm == "grouped_gemm":
            key = f"expert GEMM K={meta['K']} N={meta['N']}" + (" +GELU" if meta.get("epilogue") == ops.EPI_GELU else "")
        elif nm.endswith("_gemm"):
            key = f"{nm} K={meta['K']} N={meta['N']}"
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += t; a[2] += meta.get("flops", 0.0); a[3] += meta.get("bytes", 0.0)
    rows = []
    for key, (n, t, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        r = {"kernel": key, "launches_per_step": n / side, "avg_us": round(1e3 * t / n, 1), "ms_per_step": round(t / side, 3)}
        if fl:
            r["tflops"] = round(fl / (t * 1e-3) / 1e12, 1)
            r["frac_of_2.5PF"] = round(fl / (t * 1e-3) / 1e12 / 2500.0, 3)
        if by:
            r["tb_s"] = round(by / (t * 1e-3) / 1e12, 2)
            r["frac_of_6.3TBs"] = round(by / (t * 1e-3) / 1e12 / 6.3, 3)
        rows.append(r)
    # the same forward replayed from ONE HIP graph: what the launch gaps between ~150 short kernels cost
    graph_ms = None
    try:
        side_s = torch.cuda.Stream()
        side_s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side_s):
            step()
        torch.cuda.current_stream().wait_stream(side_s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            gout = step()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        graph_ms = e0.elapsed_time(e1) / steps
        graph_same = bool(torch.equal(gout, step()))
    except Exception as exc:      # a report, not the measurement
        graph_ms, graph_same = None, f"{type(exc).__name__}: {exc}"
    gates = [m for m in model.modules() if isinstance(m, sm.Gate)]
    out = {"model": name, "batch": batch, "ms_per_step_median": round(med, 3), "min_max": [round(ms[0], 3), round(ms[-1], 3)],
           "images_per_s": round(batch / med * 1e3, 1),
           "hip_graph_replay_ms_per_step": None if graph_ms is None else round(graph_ms, 3), "hip_graph_replay_equals_eager": graph_same,
           "timed_kernel_ms_per_step": round(sum(r["ms_per_step"] for r in rows), 3),
           "skipped_token_fraction": (round(sum(gt._skipped_tokens for gt in gates) / max(1, sum(gt._total_tokens for gt in gates)), 3)
                                      if gates else None),
           "kernels": rows}
    print(json.dumps(out))
    print(f"\n{name}, batch {batch}: {med:.3f} ms per eval forward = {batch / med * 1e3:.0f} images/s"
          + (f"; replayed from one HIP graph {graph_ms:.3f} ms = {batch / graph_ms * 1e3:.0f} images/s (same bits: {graph_same})"
             if graph_ms else f"; HIP graph: {graph_same}"), file=sys.stderr)
    for r in rows:
        print(f"  {r['kernel']:44s} x{r['launches_per_step']:5.1f}  {r['avg_us']:8.1f} us  {r['ms_per_step']:7.3f} ms/step  "
              + (f"{r['tflops']:7.1f} TF/s ({r['frac_of_2.5PF']:.3f})" if "tflops" in r else "")
              + (f"{r['tb_s']:6.2f} TB/s ({r['frac_of_6.3TBs']:.3f})" if "tb_s" in r else ""), file=sys.stderr)


if __name__ == "__main__":
    main()
