"""GPU tests of the token-skip gate (models/resMoE.py:32-85 `Gate`) and of the fused residual-MoE block
(models/resMoE.py:126-145 `forward_residule_moe`) on the HIP path: ops.gate_ln_router / ops.zero_row_output through
the C-ABI, the `Gate` module, `FMoETransformerMLP.forward_norm_gate_add`, and the `resmoe_*` models end to end."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import moe_oracle as mo  # noqa: E402
import slim_switch_moe_vit_amd as sm  # noqa: E402
from slim_switch_moe_vit_amd import ops  # noqa: E402

DEV = "cuda:0"


def _gen(s):
    return torch.Generator().manual_seed(s)


@pytest.mark.parametrize("d", [192, 384, 768, 1024])
@pytest.mark.parametrize("thr", [0.5, 0.9, 0.27])
def test_skip_gate_decisions_are_bit_exact_and_counted(d, thr):
    """mask == oracle.skip_gate (decision on the float64 logit), incl. rows placed within 1e-7 of the threshold (they
    must take the f64 redo path), a row exactly ON it (not skipped: strict >), zero rows (decided by the bias);
    the device counter equals the number of skipped tokens; the disabled gate passes everything."""
    g = _gen(d + int(thr * 100))
    B, N = 3, 211
    x = torch.randn(B, N, d, generator=g)
    w = torch.randn(1, d, generator=g) * 0.08
    b = torch.randn(1, generator=g) * 0.1
    zt = mo.skip_logit_threshold(thr)
    # rows engineered onto / next to the decision boundary: scale a row so that its f64 logit is zt + delta
    flat = x.reshape(-1, d)
    for i, delta in enumerate([0.0, 3e-8, -3e-8, 2e-7, -2e-7, 1e-6, -1e-6]):
        r = flat[10 + i]
        z0 = float(r.double() @ w[0].double())
        flat[10 + i] = (r.double() * ((zt + delta - float(b)) / z0)).float()
    flat[40:45] = 0.0
    thr_t = torch.tensor(thr, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    r = ops.gate_ln_router(flat.to(DEV), w.to(DEV), b.to(DEV), thr_t, want_mask=True, skip_count=cnt,
                           xn16_dtype=torch.float16, want_xn32=True)
    ref = mo.skip_gate(x, w, b, float(torch.tensor(thr)))          # the f32 value the module's buffer holds
    assert torch.equal(r["mask"].cpu().reshape(B, N, 2), ref)
    n_skip = int(ref[..., 0].sum())
    assert 0 < n_skip < B * N
    assert int(cnt.item()) == n_skip
    keep = ref[..., 1].reshape(-1, 1)
    assert torch.equal(r["xn32"].cpu(), flat)                       # no LayerNorm, no zero-row constant: the rows themselves
    assert torch.equal(r["xn16"].cpu(), (flat * keep).half())       # masked 16-bit operand image
    off = ops.gate_ln_router(flat.to(DEV), w.to(DEV), b.to(DEV), None, want_mask=True)
    assert torch.all(off["mask"][:, 1] == 1) and torch.all(off["mask"][:, 0] == 0)


def test_gate_module_eval_uses_the_kernel_and_matches_the_training_composition():
    d = 192
    gate = sm.Gate(d, 1.0, target_threshold=0.6, starting_threshold=0.8).to(DEV)
    with torch.no_grad():
        gate.head[1].weight.mul_(3.0)
        gate.head[1].bias.fill_(0.4)
    x = torch.randn(4, 50, d, generator=_gen(1)).to(DEV)
    gate.eval()
    with torch.no_grad():
        m_eval = gate(x)
    ref = mo.skip_gate(x.cpu(), gate.head[1].weight.detach().cpu(), gate.head[1].bias.detach().cpu(), float(gate.threshold))
    assert torch.equal(m_eval.cpu(), ref)
    assert gate._total_tokens == 200 and gate._skipped_tokens == float(ref[..., 0].sum())
    gate.train()                                  # training: _threshold, differentiable straight-through masks
    xg = x.clone().requires_grad_(True)
    m_tr = gate(xg)
    ref_tr = mo.skip_gate(x.cpu(), gate.head[1].weight.detach().cpu(), gate.head[1].bias.detach().cpu(), float(gate._threshold))
    assert torch.equal(m_tr.detach().cpu(), ref_tr)
    (m_tr[..., 1] * torch.arange(50, device=DEV)).sum().backward()
    assert gate.head[1].weight.grad is not None and float(gate.head[1].weight.grad.abs().sum()) > 0
    gate.step(torch.tensor(0.15, device=DEV))
    assert abs(float(gate._threshold) - 0.65) < 1e-6
    gate.step(torch.tensor(0.15, device=DEV))
    assert abs(float(gate._threshold) - 0.6) < 1e-6  # never below the target
    gate.disable = True
    with torch.no_grad():
        m_off = gate(x)
    assert torch.all(m_off[..., 1] == 1) and torch.all(m_off[..., 0] == 0)


@pytest.mark.parametrize("E,k,bias_tie", [(8, 2, False), (4, 1, True), (8, 1, False), (5, 3, False)])
def test_zero_row_output_is_the_moe_of_a_zero_row(E, k, bias_tie):
    d, h = 192, 768
    g = _gen(E * 10 + k)
    bg = torch.randn(E, generator=g) * 0.3
    if bias_tie:
        bg[:] = 0.0                                              # all tied: lowest ids win
    w1 = torch.randn(E, h, d, generator=g) * 0.05
    b1 = torch.randn(E, h, generator=g) * 0.5
    w2 = torch.randn(E, d, h, generator=g) * 0.05
    b2 = torch.randn(E, d, generator=g) * 0.1
    got = ops.zero_row_output(bg.to(DEV), k, w2.to(DEV), b1.to(DEV), b2.to(DEV)).cpu()
    ref = mo.moe_forward(torch.zeros(1, d), torch.randn(E, d, generator=g), bg, w1, b1, w2, b2, k).out[0]
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, float(ref.abs().max()))


def _mk_moe(d, h, E, k, seed, cd=None):
    g = _gen(seed)
    mod = sm.CustomizedMoEMLP(d, h, E, k, 0.0, **({"compute_dtype": cd} if cd is not None else {}))
    with torch.no_grad():
        mod.gate.gate.weight.copy_(torch.randn(E, d, generator=g) * 0.1)
        mod.gate.gate.bias.copy_(torch.randn(E, generator=g) * 0.05)
        mod.experts.htoh4.weight.copy_(torch.randn(E, h, d, generator=g) * 0.02)
        mod.experts.htoh4.bias.copy_(torch.randn(E, h, generator=g) * 0.2)
        mod.experts.h4toh.weight.copy_(torch.randn(E, d, h, generator=g) * 0.02)
        mod.experts.h4toh.bias.copy_(torch.randn(E, d, generator=g) * 0.1)
    return mod.to(DEV).eval()


@pytest.mark.parametrize("d,h,E,k", [(192, 768, 8, 2), (768, 3072, 8, 1), (384, 768, 4, 1)])
def test_fused_moe_half_with_skip_gate(d, h, E, k):
    """FMoETransformerMLP.forward_norm_gate_add (LayerNorm + skip gate + router in one pass, skipped tokens not
    dispatched, residual image updated in place) against (a) the oracle's block half on the same weights,
    (b) the unfused composition of the same kernels fed with the fused pass's own normed rows: bit for bit on the
    tokens that enter the experts, to f16 rounding on the skipped ones (their constant comes from an f32 GEMV)."""
    T = 3000
    g = _gen(d + E + k)
    x = (torch.randn(T, d, generator=g) * 1.5 + 0.2)
    ln = torch.nn.LayerNorm(d, eps=1e-6)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.2 * torch.randn(d, generator=g)); ln.bias.copy_(0.1 * torch.randn(d, generator=g))
    ln = ln.to(DEV)
    mod = _mk_moe(d, h, E, k, seed=11)
    gate = sm.Gate(d, 1.0, target_threshold=0.55).to(DEV).eval()
    with torch.no_grad():
        gate.head[1].weight.copy_(torch.randn(1, d, generator=g) * 0.05)
        gate.head[1].bias.fill_(0.1)
    xg = x.to(DEV)
    with torch.no_grad():
        assert mod.norm_gate_fusable(xg, ln)
        fused = mod.forward_norm_gate_add(xg, ln, gate)
        idx, score, counts, offsets, pos, inv_pos = mod.last_plan
        # the same pass again, for its intermediate images
        r = ops.gate_ln_router(xg, gate.head[1].weight, gate.head[1].bias, gate.threshold,
                               ln=(ln.weight, ln.bias, ln.eps), wg=mod.gate.gate.weight.float(), bg=mod.gate.gate.bias.float(),
                               k=k, xn16_dtype=torch.float16, want_xn32=True, want_mask=True)
        xn = r["xn32"]
        keep = r["mask"][:, 1:2]
        skipped = (keep[:, 0] == 0)
        n_skip = int(skipped.sum())
        assert 0 < n_skip < T and gate._skipped_tokens == n_skip and gate._total_tokens == T
        assert int(counts.sum()) == (T - n_skip) * k, "skipped tokens are not dispatched"
        # (b) unfused composition of the same kernels on the same rows
        unfused = mod.forward_add(xn * keep, xn)
        assert torch.equal(fused[~skipped], unfused[~skipped])
        tol = 1e-3 * max(1.0, float(unfused.abs().max()))
        assert (fused[skipped] - unfused[skipped]).abs().max().item() <= tol
        assert torch.equal(idx, mod.last_plan[0]), "a skipped token routes like the zero row it is"
    # (a) the oracle's half block: xn = LN(x); m = gate(xn); out = moe(xn * keep) + xn
    sd = {kk: v.detach().cpu() for kk, v in mod.state_dict().items()}
    xn_c = xn.cpu()
    m = mo.skip_gate(xn_c[None], gate.head[1].weight.detach().cpu(), gate.head[1].bias.detach().cpu(), float(gate.threshold))[0]
    assert torch.equal(m[:, 1:2], keep.cpu())
    o = mo.moe_forward(xn_c * m[:, 1:2], sd["gate.gate.weight"], sd["gate.gate.bias"], sd["experts.htoh4.weight"],
                       sd["experts.htoh4.bias"], sd["experts.h4toh.weight"], sd["experts.h4toh.bias"], k)
    assert torch.equal(idx.cpu(), o.idx)
    ref = o.out + xn_c
    diff = fused.cpu() - ref
    assert diff.abs().max().item() <= 1e-3 * max(1.0, float(o.out.abs().max())) + 1e-6
    assert (diff.norm() / o.out.norm()).item() <= 1e-3


def _init_resmoe(model, seed):
    g = _gen(seed)
    with torch.no_grad():
        for blk in model.blocks:
            m = blk.mlp
            m.gate.gate.weight.copy_(torch.randn(m.gate.gate.weight.shape, generator=g) * 0.1)
            m.gate.gate.bias.copy_(torch.randn(m.gate.gate.bias.shape, generator=g) * 0.05)
            m.experts.htoh4.weight.copy_(torch.randn(m.experts.htoh4.weight.shape, generator=g) * 0.02)
            m.experts.h4toh.weight.copy_(torch.randn(m.experts.h4toh.weight.shape, generator=g) * 0.02)
            m.experts.htoh4.bias.copy_(torch.randn(m.experts.htoh4.bias.shape, generator=g) * 0.1)
            for gt in (blk.moe_gate, blk.dense_gate):   # make the skip gates fire on a visible fraction of tokens
                gt.head[1].weight.copy_(torch.randn(gt.head[1].weight.shape, generator=g) * 0.05)
                gt.head[1].bias.fill_(1.8)
        model.head.weight.copy_(torch.randn(model.head.weight.shape, generator=g) * 0.02)
    return model


@pytest.mark.parametrize("name,heads", [("resmoe_tiny_patch16_224_expert8", 3), ("resmoe_base_patch16_224_expert8_top1", 12)])
def test_resmoe_model_fused_path_matches_oracle_and_composed_path(name, heads):
    """The reference's live model (resmoe_tiny_patch16_224_expert8: E = 8, top-2, token-skip gates, residual on the
    normed activations; and its ViT-B top-1 sibling) under fp16 autocast: every block takes the fused path
    (gate counters move, no module-level gate call), gates fire, and the logits agree with the oracle's
    vit_forward(residual_moe=True) and with the module-composed path of the same model."""
    torch.manual_seed(0)
    depth = 12 if heads == 3 else 3
    model = _init_resmoe(sm.create_model(name, num_classes=10, depth=depth, starting_threshold=1.0,
                                         target_threshold=0.9), 3).eval()
    k = model.blocks[0].mlp.top_k
    sd = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    images = torch.randn(4, 3, 224, 224, generator=_gen(5))
    ref = mo.vit_forward(images, sd, depth=depth, num_heads=heads, k=k, residual_moe=True)
    model = model.to(DEV)
    from slim_switch_moe_vit_amd import resmoe
    calls = {"fused": 0}
    orig = resmoe._residual_block_fused

    def spy(blk, x):
        calls["fused"] += 1
        return orig(blk, x)

    resmoe._residual_block_fused = spy
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            out = model(images.to(DEV)).float().cpu()
    finally:
        resmoe._residual_block_fused = orig
    assert calls["fused"] == depth, "every block must take the fused path under fp16 autocast"
    T = 4 * 197
    for blk in model.blocks:
        for gt in (blk.dense_gate, blk.moe_gate):
            assert gt._total_tokens == T and 0 < gt._skipped_tokens < T
    assert (out - ref).abs().max().item() <= 5e-2
    # the module-composed path of the same model (f32 activations, HIP gate + MoE operator, torch glue)
    for blk in model.blocks:
        for gt in (blk.dense_gate, blk.moe_gate):
            gt._skipped_tokens = 0
    with torch.no_grad():
        comp = model(images.to(DEV)).float().cpu()
    assert (comp - ref).abs().max().item() <= 5e-2
    assert (out - comp).abs().max().item() <= 5e-2
