#!/usr/bin/env python3
"""Harness fuzz on a one-rank RCCL group: loaders whose batch sizes GROW after the exchange buffers were agreed (8, 16, 4, 16, 12 images),
through engine.evaluate (speculative static exchange, HIP graph on / off) and engine.train_one_epoch (speculative / counted), for the
reference's two model families.  Whatever overflows on the way -- rows beyond the agreed count, groups beyond their slots -- the numbers
must be the counted exchange's: evaluate's metrics exactly, the trained parameters bit for bit.

    python tools/ep_harness_fuzz.py"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import slim_switch_moe_vit_amd as sm  # noqa: E402
from test_gpu_model import _init  # noqa: E402

DEV = "cuda:0"


def build(name, seed=41):
    torch.manual_seed(0)
    kw = dict(starting_threshold=0.5, target_threshold=0.5) if name.startswith("resmoe") else {}
    model = _init(sm.create_model(name, num_classes=10, depth=2, drop_path_rate=0.0, **kw), seed)
    if name.startswith("resmoe"):
        with torch.no_grad():
            for blk in model.blocks:
                for gt in (blk.dense_gate, blk.moe_gate):
                    gt.head[1].weight.normal_(0, 0.5, generator=torch.Generator().manual_seed(3))
    model = model.to(DEV)
    for blk in model.blocks:
        blk.mlp.force_ep = True
    model.ep_micro_batches = 1
    return model


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29593", rank=0, world_size=1, device_id=torch.device(DEV))
    ok = True
    try:
        g = torch.Generator().manual_seed(9)
        loader = [(torch.randn(b, 3, 224, 224, generator=g), torch.randint(0, 10, (b,), generator=g)) for b in (8, 16, 4, 16, 12)]
        for name in ("moe_tiny_patch16_224_expert8", "resmoe_tiny_patch16_224_expert8"):
            ref = sm.evaluate(loader, build(name), DEV, ep_speculative=None, hip_graph=False)
            for graph in (False, True):
                got = sm.evaluate(loader, build(name), DEV, ep_speculative=1.25, hip_graph=graph)
                same = all(got[k] == ref[k] for k in ("loss", "acc1", "acc5"))
                ok &= same
                print(f"{name}: evaluate, growing batches, speculative static, graph {got['hip_graph']}: metrics == counted: {same}; "
                      f"{got['ep_repeated_steps']} of {len(loader)} steps repeated", flush=True)
            finals = {}
            for tag, alpha in (("counted", None), ("speculative", 1.25)):
                model = build(name)
                opt = sm.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05)
                st = sm.train_one_epoch(model, torch.nn.CrossEntropyLoss(), loader, opt, DEV, 0, sm.NativeScaler(), max_norm=1.0,
                                        ep_speculative=alpha)
                torch.cuda.synchronize()
                finals[tag] = ([p.detach().clone() for p in model.parameters()], st)
            same = all(torch.equal(a, b) for a, b in zip(finals["counted"][0], finals["speculative"][0]))
            ok &= same
            print(f"{name}: train_one_epoch, growing batches: parameters == counted: {same}; loss {finals['speculative'][1]['loss']:.6f} / "
                  f"{finals['counted'][1]['loss']:.6f}; {finals['speculative'][1]['ep_repeated_steps']} of {len(loader)} forwards repeated",
                  flush=True)
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()
    print("ALL OK" if ok else "MISMATCH", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
