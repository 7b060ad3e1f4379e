"""Fixtures from the reference's OWN hot-path code that lives in-tree: the token-skip ``Gate`` (models/resMoE.py:32-85) and the
block wrapper ``forward_residule_moe`` (models/resMoE.py:126-145).  Run HERE (build container), never on the GPU box:

    python tests/golden/make_golden_resmoe.py

``import models.resMoE`` fails on the module's top-level imports of ``fmoe`` / ``timm`` (absent from the image: SURVEY.md 8c), but
neither the class nor the function uses anything from those packages.  So this script reads the reference file, keeps the two
definitions (and the file's own plain ``math`` / ``typing`` / ``torch`` imports) from its syntax tree, and executes exactly that
code -- nothing of ``fmoe`` / ``timm`` is stubbed or imitated.  The block's sub-modules are the reference's own importable
``models/layers.py`` classes (``Attention`` 227-269, ``Mlp`` 391-414: the dense FFN = the E = 1 MoE) and ``nn.LayerNorm``.

Written (data only -- inputs, parameters, expected outputs, gradients):
  ref_gate_tiny.npz      Gate.forward in eval / train-hard / train-soft / disabled, rows placed within a few float32 ulp of the
                         threshold, with d(mask)/d(x, weight, bias) of the two training modes and the token counters
  ref_resblock_tiny.npz  forward_residule_moe in eval (both gates skipping ~30-40 %) and in train-hard mode with its gradients
"""
import ast
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# where the .npz files go: this directory, or -- tests/test_oracle.py::test_committed_fixtures_regenerate_bit_identically -- a scratch one
OUT = os.environ.get("SLIMMOE_GOLDEN_OUT", HERE)
REF_RESMOE = "/root/reference/models/resMoE.py"
REF_LAYERS = "/root/reference/models/layers.py"


def load_reference_defs():
    """{'Gate': class, 'forward_residule_moe': function} compiled from the reference file's own text."""
    src = open(REF_RESMOE).read()
    tree = ast.parse(src, REF_RESMOE)
    keep = []
    for node in tree.body:
        if isinstance(node, ast.Import) and all(a.name.split(".")[0] in ("math", "typing", "torch") for a in node.names):
            keep.append(node)
        elif isinstance(node, ast.ClassDef) and node.name == "Gate":
            keep.append(node)
        elif isinstance(node, ast.FunctionDef) and node.name == "forward_residule_moe":
            keep.append(node)
    names = [getattr(n, "name", None) for n in keep]
    assert "Gate" in names and "forward_residule_moe" in names, names
    ns = {}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF_RESMOE, "exec"), ns)
    return ns


def load_ref_layers():
    spec = importlib.util.spec_from_file_location("ref_layers", REF_LAYERS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def near_threshold_rows(x, w, b, thr, rows, g):
    """Moves ``rows`` of x [T, d] along w so that their logit lands at logit(thr) + delta for deltas from 0 to a few 1e-6 (both
    signs): float32 rounding of the dot product / the sigmoid decides these tokens."""
    z_thr = np.log(thr / (1 - thr))
    deltas = [0.0, 2e-8, -2e-8, 6e-8, -6e-8, 1.5e-7, -1.5e-7, 4e-7, -4e-7, 1e-6, -1e-6, 3e-6, -3e-6, 1e-5, -1e-5, 1e-4, -1e-4]
    wd = w.double().reshape(-1)
    for i, t in enumerate(rows):
        xt = x[t].double()
        z = float(xt @ wd + b.double())
        want = z_thr + deltas[i % len(deltas)]
        xt = xt + (want - z) * wd / float(wd @ wd)
        x[t] = xt.float()
    return x


def gate_fixture(ref, out):
    Gate = ref["Gate"]
    d, B, N = 192, 3, 41
    g = torch.Generator().manual_seed(2024)
    w = torch.randn(1, d, generator=g) * 0.15
    b = torch.randn(1, generator=g) * 0.1
    x = torch.randn(B * N, d, generator=g)
    thr_eval, thr_train = 0.62, 0.71
    rows = torch.randperm(B * N, generator=g)[:34].tolist()
    x = near_threshold_rows(x, w, b[0], thr_eval, rows[:17], g)
    x = near_threshold_rows(x, w, b[0], thr_train, rows[17:], g)
    x = x.reshape(B, N, d)
    dret = torch.randn(B, N, 2, generator=g)
    out.update(x=x.numpy(), w=w.numpy(), b=b.numpy(), thr_eval=np.float32(thr_eval), thr_train=np.float32(thr_train),
               dret=dret.numpy(), near_rows_eval=np.array(rows[:17]), near_rows_train=np.array(rows[17:]))

    def make(is_hard=True):
        gt = Gate(d, 1.0, dropout=0.0, target_threshold=thr_eval, starting_threshold=thr_train, is_hard=is_hard)
        with torch.no_grad():
            gt.head[1].weight.copy_(w)
            gt.head[1].bias.copy_(b)
        return gt

    # eval: compares against `threshold`
    gt = make().eval()
    with torch.no_grad():
        m = gt(x)
    out.update(eval_mask=m.numpy(), eval_total=np.int64(gt._total_tokens), eval_skipped=np.float64(gt._skipped_tokens))
    # the f32 sigmoid the reference's comparison saw (for attributing near-threshold rows)
    with torch.no_grad():
        out["prob_f32"] = torch.sigmoid(gt.head(x)).numpy()
    # train, hard: compares against `_threshold`; straight-through gradients
    for mode, hard in (("train_hard", True), ("train_soft", False)):
        gt = make(hard).train()
        xg = x.clone().requires_grad_(True)
        m = gt(xg)
        m.backward(dret)
        out.update({f"{mode}_mask": m.detach().numpy(), f"{mode}_dx": xg.grad.numpy(),
                    f"{mode}_dw": gt.head[1].weight.grad.numpy(), f"{mode}_db": gt.head[1].bias.grad.numpy(),
                    f"{mode}_total": np.int64(gt._total_tokens), f"{mode}_skipped": np.float64(gt._skipped_tokens)})
    # disabled
    gt = make().eval()
    gt.disable = True
    with torch.no_grad():
        out["disabled_mask"] = gt(x).numpy()
    # step(): _threshold anneals down to threshold
    gt = make()
    seq = []
    for _ in range(4):
        gt.step(torch.tensor(0.04))
        seq.append(float(gt._threshold))
    out["step_sequence"] = np.array(seq, dtype=np.float64)


def resblock_fixture(ref, layers, out):
    Gate, fwd = ref["Gate"], ref["forward_residule_moe"]
    d, heads, B, N = 192, 3, 2, 50
    g = torch.Generator().manual_seed(777)

    class Holder(torch.nn.Module):
        """What the reference's factory leaves on a Block (models/resMoE.py:163-186): the attributes forward_residule_moe reads."""

        def __init__(self):
            super().__init__()
            self.norm1 = torch.nn.LayerNorm(d, eps=1e-6)
            self.attn = layers.Attention(d, num_heads=heads, qkv_bias=True)
            self.drop_path = torch.nn.Identity()
            self.norm2 = torch.nn.LayerNorm(d, eps=1e-6)
            self.mlp = layers.Mlp(d, 4 * d)
            self.dense_gate = Gate(d, 1.0, target_threshold=0.55, starting_threshold=0.6)
            self.moe_gate = Gate(d, 1.0, target_threshold=0.55, starting_threshold=0.6)

    blk = Holder()
    with torch.no_grad():
        for name, p in blk.named_parameters():
            if "norm" in name and name.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif "gate" in name:
                p.copy_(torch.randn(p.shape, generator=g) * (0.6 if p.dim() > 1 else 0.1))
            elif "qkv.weight" in name:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.04)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    blk.forward = fwd.__get__(blk, Holder)            # the reference's own bind idiom (models/resMoE.py:185-186)
    x = torch.randn(B, N, d, generator=g) * 1.5
    dy = torch.randn(B, N, d, generator=g) * 0.1
    out.update({"p." + k: v.detach().numpy() for k, v in blk.state_dict().items()})
    out.update(x=x.numpy(), dy=dy.numpy(), num_heads=np.array(heads))
    masks = {}
    hooks = [blk.dense_gate.register_forward_hook(lambda m, i, o: masks.__setitem__("dense", o.detach().clone())),
             blk.moe_gate.register_forward_hook(lambda m, i, o: masks.__setitem__("moe", o.detach().clone()))]
    blk.eval()
    with torch.no_grad():
        y = blk.forward(x)
    out.update(eval_y=y.numpy(), eval_dense_mask=masks["dense"].numpy(), eval_moe_mask=masks["moe"].numpy())
    blk.train()
    xg = x.clone().requires_grad_(True)
    y = blk.forward(xg)
    y.backward(dy)
    out.update(train_y=y.detach().numpy(), train_dx=xg.grad.numpy(), train_dense_mask=masks["dense"].numpy(),
               train_moe_mask=masks["moe"].numpy())
    out.update({"g." + k: p.grad.numpy() for k, p in blk.named_parameters()})
    for h in hooks:
        h.remove()
    print("resblock: eval skip fractions", float(out["eval_dense_mask"][..., 0].mean()), float(out["eval_moe_mask"][..., 0].mean()),
          "train", float(out["train_dense_mask"][..., 0].mean()), float(out["train_moe_mask"][..., 0].mean()))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    ref = load_reference_defs()
    layers = load_ref_layers()
    gate = {}
    gate_fixture(ref, gate)
    np.savez_compressed(os.path.join(OUT, "ref_gate_tiny.npz"), **gate)
    blk = {}
    resblock_fixture(ref, layers, blk)
    np.savez_compressed(os.path.join(OUT, "ref_resblock_tiny.npz"), **blk)
    print("gate: eval skipped", gate["eval_skipped"], "of", gate["eval_total"], "| train-hard skipped", gate["train_hard_skipped"])
    print("wrote ref_gate_tiny.npz, ref_resblock_tiny.npz")


if __name__ == "__main__":
    sys.exit(main())
