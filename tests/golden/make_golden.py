"""Generates the committed fixtures in tests/golden/.  Run HERE (build container), never on the GPU box:

    python tests/golden/make_golden.py

1. ``ref_mlp_tiny.npz`` / ``ref_layernorm_tiny.npz`` -- outputs of the REFERENCE's own importable code:
   /root/reference/models/layers.py (``Mlp`` 391-414 = the dense FFN every expert generalises, and the manual
   ``LayerNorm`` 160-224; ``Attention`` 227-269 -> ``ref_attention_tiny.npz``), loaded standalone by file path
   (the rest of the reference needs timm / fmoe, which are absent: SURVEY.md 8c).  These pin the per-expert FFN of the oracle and of the HIP path (E = 1 <=> Mlp).
2. ``oracle_moe_small.npz`` -- regression vectors produced by oracle/moe_oracle.py (NOT reference outputs;
   the MoE operator itself is 'parity unpinned', see the oracle header).
Only data is stored: inputs, parameters, expected outputs.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# where the .npz files go: this directory, or -- tests/test_oracle.py::test_committed_fixtures_regenerate_bit_identically -- a scratch one
OUT = os.environ.get("SLIMMOE_GOLDEN_OUT", HERE)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_LAYERS = "/root/reference/models/layers.py"


def load_ref_layers():
    spec = importlib.util.spec_from_file_location("ref_layers", REF_LAYERS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    ref = load_ref_layers()

    # ---- 1a. reference Mlp at ViT-Ti dims (d=192, h=768), 3 images x 37 tokens
    g = torch.Generator().manual_seed(1234)
    mlp = ref.Mlp(192, 768)
    with torch.no_grad():
        torch.nn.init.trunc_normal_(mlp.fc1.weight, std=0.02, a=-2, b=2, generator=g)
        torch.nn.init.trunc_normal_(mlp.fc2.weight, std=0.02, a=-2, b=2, generator=g)
        mlp.fc1.bias.copy_(torch.randn(768, generator=g) * 0.02)
        mlp.fc2.bias.copy_(torch.randn(192, generator=g) * 0.02)
    mlp.eval()
    x = torch.randn(3, 37, 192, generator=g)
    with torch.no_grad():
        y = mlp(x)
    np.savez_compressed(os.path.join(OUT, "ref_mlp_tiny.npz"), x=x.numpy(), fc1_w=mlp.fc1.weight.detach().numpy(),
                        fc1_b=mlp.fc1.bias.detach().numpy(), fc2_w=mlp.fc2.weight.detach().numpy(),
                        fc2_b=mlp.fc2.bias.detach().numpy(), y=y.numpy())

    # ---- 1b. reference manual LayerNorm
    ln = ref.LayerNorm(192, eps=1e-6)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(192, generator=g))
        ln.bias.copy_(0.1 * torch.randn(192, generator=g))
        yl = ln(x)
    np.savez_compressed(os.path.join(OUT, "ref_layernorm_tiny.npz"), x=x.numpy(), w=ln.weight.detach().numpy(),
                        b=ln.bias.detach().numpy(), y=yl.numpy())

    # ---- 1c. reference Attention (models/layers.py:227-269; the same computation as models/vision_transformer.py:248-280)
    #          at ViT-Ti dims (d = 192, 3 heads of 64), 2 images x 50 tokens and 1 image x 197 tokens
    att = ref.Attention(192, num_heads=3, qkv_bias=True)
    g_keep, g = g, torch.Generator().manual_seed(4321)   # own stream: the vectors of section 2 stay as committed
    with torch.no_grad():
        torch.nn.init.trunc_normal_(att.qkv.weight, std=0.1, a=-2, b=2, generator=g)   # std 0.1: a softmax that is not flat
        torch.nn.init.trunc_normal_(att.proj.weight, std=0.05, a=-2, b=2, generator=g)
        att.qkv.bias.copy_(torch.randn(576, generator=g) * 0.05)
        att.proj.bias.copy_(torch.randn(192, generator=g) * 0.05)
    att.eval()
    xa = torch.randn(2, 50, 192, generator=g)
    xb = torch.randn(1, 197, 192, generator=g)
    with torch.no_grad():
        ya, yb = att(xa), att(xb)
    np.savez_compressed(os.path.join(OUT, "ref_attention_tiny.npz"), xa=xa.numpy(), xb=xb.numpy(), ya=ya.numpy(),
                        yb=yb.numpy(), qkv_w=att.qkv.weight.detach().numpy(), qkv_b=att.qkv.bias.detach().numpy(),
                        proj_w=att.proj.weight.detach().numpy(), proj_b=att.proj.bias.detach().numpy(),
                        num_heads=np.array(3))
    g = g_keep

    # ---- 2. oracle regression vectors (small dims)
    from oracle import moe_oracle as mo

    d, h, E = 64, 128, 4
    cases = {}
    for name, (T, k, gate, cap) in {"naive_k2": (97, 2, mo.GATE_NAIVE, -1), "naive_k1": (130, 1, mo.GATE_NAIVE, -1),
                                    "switch_cap": (111, 1, mo.GATE_SWITCH, 20)}.items():
        xs = torch.randn(T, d, generator=g)
        wg = torch.randn(E, d, generator=g) * 0.2
        bg = torch.randn(E, generator=g) * 0.1
        w1 = torch.randn(E, h, d, generator=g) * 0.05
        b1 = torch.randn(E, h, generator=g) * 0.05
        w2 = torch.randn(E, d, h, generator=g) * 0.05
        b2 = torch.randn(E, d, generator=g) * 0.05
        r = mo.moe_forward(xs, wg, bg, w1, b1, w2, b2, k, gate, cap)
        for key, val in dict(x=xs, wg=wg, bg=bg, w1=w1, b1=b1, w2=w2, b2=b2, out=r.out, idx=r.idx, score=r.score,
                             counts=r.plan.counts, offsets=r.plan.offsets, pos=r.plan.pos, inv_pos=r.plan.inv_pos,
                             idx_pruned=r.plan.idx_pruned).items():
            cases[f"{name}.{key}"] = val.numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
        cases[f"{name}.meta"] = np.array([T, k, gate, cap], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "oracle_moe_small.npz"), **cases)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
