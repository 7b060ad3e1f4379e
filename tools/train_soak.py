#!/usr/bin/env python3
"""A short training soak on one GPU: N optimizer steps of a small residual-MoE / switch-MoE ViT on ONE fixed synthetic batch (the loss
must fall: the whole stack -- own forward / backward kernels, loss scaling, clipping, the fused AdamW with its 16-bit weight images --
is in the loop), peak memory per 10 steps (no growth), and the same run with the optimizer's weight images switched off
(SLIMMOE_ADAMW_SHADOW=0 semantics): the two loss curves must be IDENTICAL, since the images hold the same rounded weights either way.
usage: python3 tools/train_soak.py [model=resmoe_tiny_patch16_224_expert8] [steps=40] [batch=32]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slim_switch_moe_vit_amd as sm
from slim_switch_moe_vit_amd import optim as smo

name = sys.argv[1] if len(sys.argv) > 1 else "resmoe_tiny_patch16_224_expert8"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda", 0)


def run(shadow: bool):
    smo.SHADOW_STEP = shadow
    torch.manual_seed(0)
    kw = (dict(starting_threshold=0.55, target_threshold=0.5) if name.startswith("resmoe")
          else (dict(gate="switch", capacity_factor=1.25) if "top1" in name else {}))     # (the top-2 factories keep the naive gate)
    model = sm.create_model(name, num_classes=100, drop_path_rate=0.1, **kw).to(dev).train()
    opt = smo.AdamW(model.parameters(), lr=3e-4, weight_decay=0.05)
    scaler = smo.NativeScaler()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 3, 224, 224, generator=g).to(dev)
    y = torch.randint(0, 100, (batch,), generator=g).to(dev)
    crit = torch.nn.CrossEntropyLoss()
    moes = [m for m in model.modules() if isinstance(m, sm.FMoETransformerMLP)]
    losses, peaks = [], []
    for i in range(steps):
        torch.manual_seed(1000 + i)            # the same stochastic-depth / gate-noise draws in both runs
        with torch.autocast("cuda", dtype=torch.float16):
            out = model(x)
            loss = crit(out, y)
            aux = [a for a in (m.gate.get_loss() for m in moes) if a is not None]
            if aux:
                loss = loss + 0.01 * torch.stack([a.reshape(()) for a in aux]).sum()
        opt.zero_grad()
        scaler(loss, opt, clip_grad=1.0, parameters=model.parameters())
        losses.append(float(loss.detach()))
        if i % 10 == 9:
            peaks.append(torch.cuda.max_memory_allocated(dev) >> 20)
            torch.cuda.reset_peak_memory_stats(dev)
    return losses, peaks


on, peaks = run(True)
off, _ = run(False)
print(f"{name}, batch {batch}, {steps} steps: loss {on[0]:.4f} -> {on[-1]:.4f} (every 10th: {[round(v, 4) for v in on[::10]]}); "
      f"peak MiB per 10 steps {peaks}")
print(f"peak MiB per 10 steps: {peaks}", flush=True)
assert all(v == v and abs(v) < 1e4 for v in on), "non-finite loss"
assert on[-1] < on[0] - 0.3, "the loss did not fall on a fixed batch"
assert max(peaks[1:]) - min(peaks[1:]) <= 0.02 * max(peaks), "memory grows"     # (the routing, hence a few buffer sizes, moves with the weights)
assert on == off, f"the optimizer's 16-bit weight images change the trajectory: first difference at step {next(i for i, (a, b) in enumerate(zip(on, off)) if a != b)}"
print("ok: identical loss curves with and without the optimizer-written weight images")
