"""GPU tests of the token-skip gate (models/resMoE.py:32-85 `Gate`) and of the fused residual-MoE block
(models/resMoE.py:126-145 `forward_residule_moe`) on the HIP path: ops.gate_ln_router / ops.zero_row_output through
the C-ABI, the `Gate` module, `FMoETransformerMLP.forward_norm_gate_add`, and the `resmoe_*` models end to end."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from _mp import dtype_factor  # noqa: E402

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
    assert diff.abs().max().item() <= 1e-3 * dtype_factor() * max(1.0, float(o.out.abs().max())) + 1e-6
    assert (diff.norm() / o.out.norm()).item() <= 1e-3 * dtype_factor()


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


# ---- against outputs of the reference's OWN code (tests/golden/make_golden_resmoe.py: models/resMoE.py:32-85, 126-145) ----------
import os  # noqa: E402

F32_EPS = float(np.finfo(np.float32).eps)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixture(name):
    return {k: v for k, v in np.load(os.path.join(GOLDEN, name)).items()}


def _gate_from_fixture(g, is_hard=True):
    gate = sm.Gate(192, 1.0, target_threshold=float(g["thr_eval"]), starting_threshold=float(g["thr_train"]), is_hard=is_hard)
    with torch.no_grad():
        gate.head[1].weight.copy_(torch.from_numpy(g["w"])); gate.head[1].bias.copy_(torch.from_numpy(g["b"]))
    return gate.to(DEV)


def _explained(ours_skip, ref_mask, prob_f32, thr):
    """Tokens whose decision differs from the reference's ``sigmoid_f32(z_f32) > thr``: each within 4 f32 ulp of the threshold."""
    ref_skip = np.rint(ref_mask[..., 0]).astype(bool).reshape(-1)
    differ = np.nonzero(ours_skip.reshape(-1) != ref_skip)[0]
    gaps = [(int(t), float(prob_f32.reshape(-1)[t]) - float(np.float32(thr))) for t in differ]
    for t, gap in gaps:
        assert abs(gap) <= 4 * F32_EPS * max(float(thr), 0.25), (t, gap)
    return gaps


def test_hip_gate_against_the_reference_gate_fixture():
    """The HIP ``Gate`` (smoe_gate_ln_router) in eval and disabled mode against the reference ``Gate``'s own outputs on rows placed
    within a few f32 ulp of the threshold: decisions equal except tokens whose f32 sigmoid is within rounding of the threshold
    (listed; the kernel decides on the f64-accurate logit), masks exactly 0 / 1 where the reference is within one ulp of that, the
    device counter = the number of skipped tokens."""
    g = _fixture("ref_gate_tiny.npz")
    x = torch.from_numpy(g["x"]).to(DEV)
    gate = _gate_from_fixture(g).eval()
    with torch.no_grad():
        m = gate(x).cpu().numpy()
    assert np.all((m == 0) | (m == 1)) and np.all(m.sum(-1) == 1)
    flips = _explained(m[..., 0] > 0.5, g["eval_mask"], g["prob_f32"], g["thr_eval"])
    print(f"HIP gate, eval: {int(m[..., 0].sum())} skipped (reference {g['eval_skipped']}), decisions that differ: {flips}")
    assert len(flips) <= len(g["near_rows_eval"])
    same = np.ones(m.shape[0] * m.shape[1], dtype=bool)
    same[[t for t, _ in flips]] = False
    assert np.array_equal(m.reshape(-1, 2)[same], np.rint(g["eval_mask"]).reshape(-1, 2)[same])
    assert gate._total_tokens == int(g["eval_total"]) and gate._skipped_tokens == float(m[..., 0].sum())
    assert abs(gate._skipped_tokens - float(g["eval_skipped"])) <= len(flips) + 1e-3
    # the oracle takes the same decisions as the kernel (both on the f64 logit): bit for bit
    ref_o = mo.skip_gate(torch.from_numpy(g["x"]), torch.from_numpy(g["w"]), torch.from_numpy(g["b"]), float(g["thr_eval"]))
    assert np.array_equal(m, ref_o.numpy())
    gate.disable = True
    with torch.no_grad():
        assert np.array_equal(gate(x).cpu().numpy(), g["disabled_mask"])


def _block_from_fixture(g, compute_dtype=None):
    """A Block wired as the reference's factory wires it (models/resMoE.py:163-186), holding the fixture's parameters; the
    dense ``Mlp`` of the fixture is the single expert of an E = 1, top-1 ``CustomizedMoEMLP``."""
    from slim_switch_moe_vit_amd.vit import Block
    from slim_switch_moe_vit_amd.resmoe import forward_residule_moe
    from functools import partial
    d, heads = 192, int(g["num_heads"])
    blk = Block(d, heads, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6))
    blk.dense_gate = sm.Gate(d, 1.0, target_threshold=0.55, starting_threshold=0.6)
    blk.moe_gate = sm.Gate(d, 1.0, target_threshold=0.55, starting_threshold=0.6)
    kw = {"compute_dtype": compute_dtype} if compute_dtype is not None else {}
    blk.mlp = sm.CustomizedMoEMLP(d, 4 * d, 1, 1, 0.0, **kw)
    blk.forward = forward_residule_moe.__get__(blk, Block)
    p = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}
    sd = {k: v for k, v in p.items() if not k.startswith("mlp.")}
    sd["mlp.gate.gate.weight"], sd["mlp.gate.gate.bias"] = torch.zeros(1, d), torch.zeros(1)
    sd["mlp.experts.htoh4.weight"], sd["mlp.experts.htoh4.bias"] = p["mlp.fc1.weight"][None], p["mlp.fc1.bias"][None]
    sd["mlp.experts.h4toh.weight"], sd["mlp.experts.h4toh.bias"] = p["mlp.fc2.weight"][None], p["mlp.fc2.bias"][None]
    missing, unexpected = blk.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return blk.to(DEV)


@pytest.mark.parametrize("mode", ["fused_fp16_autocast", "composed_f32"])
def test_residual_block_against_the_reference_forward_residule_moe_fixture(mode):
    """``forward_residule_moe`` on the HIP path against the output of the reference's own function run on reference modules
    (models/resMoE.py:126-145; LayerNorm, layers.Attention, layers.Mlp = the E = 1 MoE, two Gates skipping 40-50 %): the fused
    fp16-autocast inference path (``_residual_block_fused``: two LayerNorm + gate passes, qkv / attention / projection, the MoE
    GEMMs adding into the residual image) to fp16 rounding, and the module-composed f32 path (HIP gate + f32-operand MoE operator)
    to 1e-4; both take the reference's decision on every token."""
    from slim_switch_moe_vit_amd import resmoe
    g = _fixture("ref_resblock_tiny.npz")
    x = torch.from_numpy(g["x"]).to(DEV)
    ref = torch.from_numpy(g["eval_y"])
    fused = mode == "fused_fp16_autocast"
    blk = _block_from_fixture(g, None if fused else torch.float32).eval()
    calls = {"fused": 0}
    orig = resmoe._residual_block_fused

    def spy(b, t):
        calls["fused"] += 1
        return orig(b, t)
    resmoe._residual_block_fused = spy
    try:
        with torch.no_grad():
            if fused:
                with torch.autocast("cuda", dtype=torch.float16):
                    y = blk(x)
            else:
                y = blk(x)
    finally:
        resmoe._residual_block_fused = orig
    assert calls["fused"] == (1 if fused else 0)
    T = x.shape[0] * x.shape[1]
    for gt, key in ((blk.dense_gate, "eval_dense_mask"), (blk.moe_gate, "eval_moe_mask")):
        assert gt._total_tokens == T
        # the skip counts are the reference's: no decision differs on this fixture (none of its rows sits on the threshold)
        assert gt._skipped_tokens == float(np.rint(g[key])[..., 0].sum()), key
    diff = (y.float().cpu() - ref)
    scale = max(1.0, float(ref.abs().max()))
    # measured (gpurun_out/r5_t4_prints.log): fused fp16 autocast 1.356e-3 at scale 4.16 (3.3e-4 of it), rel L2 2.09e-4; composed f32
    # 1.79e-6 (4.3e-7 of the scale), rel L2 2.7e-7 -- the bars are 3 x that
    tol_abs, tol_l2 = (1e-3 * dtype_factor(), 6.3e-4 * dtype_factor()) if fused else (1.3e-6, 8.2e-7)
    print(f"{mode}: max |y - reference| = {float(diff.abs().max()):.3e} (scale {scale:.2f}), rel L2 {float(diff.norm() / ref.norm()):.2e}")
    assert float(diff.abs().max()) <= tol_abs * scale and float(diff.norm() / ref.norm()) <= tol_l2


# ---- training: the gate's backward kernel, the gated half's Function, the whole block against the reference's gradients ----------
def _rel(got, ref):
    return float((got.double().cpu() - ref.double().cpu()).norm() / ref.double().cpu().norm().clamp(min=1e-30))


@pytest.mark.parametrize("d,gdt", [(192, torch.float32), (768, torch.float16), (384, torch.float32), (1024, torch.float16)])
def test_skip_gate_backward_kernel_matches_float64_autograd_of_the_reference_formula(d, gdt):
    """smoe_skip_gate_bwd against float64 autograd through the reference's own expressions (resMoE.py:69-77, 131-136):
    masks with the straight-through terms, tk = x * m1, skip_tk = x * m0, L = <g_f, tk> + <g_out, tk + skip_tk>."""
    T = 777
    g = _gen(d)
    xn = torch.randn(T, d, generator=g)
    w, b = torch.randn(d, generator=g) * 0.1, torch.randn(1, generator=g) * 0.1
    g_f = (torch.randn(T, d, generator=g) * 0.3).to(gdt)
    g_out = torch.randn(T, d, generator=g) * 0.2
    thr = 0.55
    xr, wr, br = xn.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    prob = torch.sigmoid(xr @ wr + br)[:, None]
    _prob = 1 - prob
    skip_tk = (prob > thr).double() + _prob.detach() - _prob
    tk = (prob <= thr).double() + prob.detach() - prob
    loss = (g_f.double() * (xr * tk)).sum() + (g_out.double() * (xr * tk + xr * skip_tk)).sum()
    loss.backward()
    mask = torch.cat([(prob > thr).float(), (prob <= thr).float()], dim=1).detach().float()
    dxn, dz = ops.skip_gate_bwd(xn.to(DEV), g_f.to(DEV), g_out.to(DEV), w.to(DEV), b.to(DEV), mask.to(DEV))
    assert _rel(dxn, xr.grad) <= 2e-5
    dw = ops.gate_wgrad(dz.reshape(-1, 1), xn.to(DEV)).reshape(-1)
    assert _rel(dw, wr.grad) <= 2e-5 and abs(float(dz.sum()) - float(br.grad)) <= 2e-5 * max(1.0, abs(float(br.grad)))
    # a disabled gate: constant masks, no gate gradient
    dxn0, dz0 = ops.skip_gate_bwd(xn.to(DEV), g_f.to(DEV), g_out.to(DEV), w.to(DEV), b.to(DEV), None, gate_on=False)
    assert _rel(dxn0, g_f.double() + g_out.double()) <= 1e-6 and float(dz0.abs().max()) == 0.0


@pytest.mark.parametrize("d,gdt,with_out,T", [(192, torch.float32, True, 777), (768, torch.float16, True, 2500),
                                              (768, torch.float32, False, 1301), (1024, torch.float16, True, 333),
                                              (384, torch.float32, True, 64)])
def test_gate_ln_backward_in_one_pass_matches_float64_autograd_of_the_reference_formula(d, gdt, with_out, T):
    """smoe_gate_ln_bwd (LayerNorm backward + the gate's straight-through gradients + both modules' parameter gradients from the
    half's input x alone) against float64 autograd through LayerNorm and the reference's expressions (resMoE.py:69-77, 126-136):
    xn = norm(x), tk = xn * m1, skip_tk = xn * m0, L = <g_f, tk> + <g_out, tk + skip_tk>; and a disabled gate."""
    g = _gen(d + T)
    x = torch.randn(T, d, generator=g) * 1.7 + 0.3
    gam, bet = 1.0 + 0.2 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    w, b = torch.randn(d, generator=g) * 0.1, torch.randn(1, generator=g) * 0.1
    g_f = (torch.randn(T, d, generator=g) * 0.3).to(gdt)
    g_out = torch.randn(T, d, generator=g) * 0.2 if with_out else None
    thr, eps = 0.55, 1e-6
    leaves = [t.double().requires_grad_(True) for t in (x, gam, bet, w, b)]
    xr, gr, btr, wr, br = leaves
    xn = torch.nn.functional.layer_norm(xr, (d,), gr, btr, eps)
    prob = torch.sigmoid(xn @ wr + br)[:, None]
    _prob = 1 - prob
    skip_tk = (prob > thr).double() + _prob.detach() - _prob
    tk = (prob <= thr).double() + prob.detach() - prob
    loss = (g_f.double() * (xn * tk)).sum()
    if with_out:
        loss = loss + (g_out.double() * (xn * tk + xn * skip_tk)).sum()
    loss.backward()
    mask = torch.cat([(prob > thr).float(), (prob <= thr).float()], dim=1).detach().float()
    dev = lambda t: t.to(DEV) if t is not None else None  # noqa: E731
    dx, dg, db_, dgw, dgb, dz = ops.gate_ln_bwd(dev(x), dev(g_f), dev(g_out), dev(gam), dev(bet), eps, dev(w), dev(b), dev(mask),
                                                want_dz=True)
    worst = max(_rel(dx, xr.grad), _rel(dg, gr.grad), _rel(db_, btr.grad), _rel(dgw, wr.grad))
    print(f"gate_ln_bwd[{gdt}, with_out={with_out}]: worst relative L2 vs float64 autograd {worst:.2e}")
    tol = 8e-7       # measured <= 2.7e-7 for f32 and f16 upstream gradients alike (gpurun_out/r5_t5_prints.log): 3 x
    assert _rel(dx, xr.grad) <= tol, _rel(dx, xr.grad)
    assert _rel(dg, gr.grad) <= tol and _rel(db_, btr.grad) <= tol
    e_b = abs(float(dgb) - float(br.grad)) / max(1.0, abs(float(br.grad)))
    e_z = abs(float(dz.sum()) - float(br.grad)) / max(1.0, abs(float(br.grad)))
    print(f"  gate bias gradient {e_b:.2e}, sum of dz {e_z:.2e} (relative to max(1, |db|))")
    assert _rel(dgw, wr.grad) <= tol and e_b <= 1e-5 and e_z <= 1e-5     # (scalars: f32 sums over every row)
    again = ops.gate_ln_bwd(dev(x), dev(g_f), dev(g_out), dev(gam), dev(bet), eps, dev(w), dev(b), dev(mask))
    assert all(torch.equal(a, c) for a, c in zip(again[:5], (dx, dg, db_, dgw, dgb)))          # deterministic
    # the three-kernel composition it replaces (saved xn from the forward kernel's own LayerNorm)
    xn32 = ops.layernorm(dev(x), dev(gam), dev(bet), eps, torch.float32)
    dxn, dz3 = ops.skip_gate_bwd(xn32, dev(g_f), dev(g_out), dev(w), dev(b), dev(mask))
    dx3, dg3, db3 = ops.layernorm_bwd(dev(x), dxn, dev(gam), eps)
    assert _rel(dx, dx3) <= 1e-5 and _rel(dg, dg3) <= 1e-5 and _rel(db_, db3) <= 1e-5 and _rel(dz, dz3) <= 1e-5
    # a disabled gate: constant masks, no gate gradient; dx = LayerNorm backward of g_f + g_out
    leaves0 = [t.double().requires_grad_(True) for t in (x, gam, bet)]
    xn0 = torch.nn.functional.layer_norm(leaves0[0], (d,), leaves0[1], leaves0[2], eps)
    l0 = (g_f.double() * xn0).sum() + ((g_out.double() * xn0).sum() if with_out else 0.0)
    l0.backward()
    dx0, dg0, db0, dgw0, dgb0, _ = ops.gate_ln_bwd(dev(x), dev(g_f), dev(g_out), dev(gam), dev(bet), eps, dev(w), dev(b), None,
                                                   gate_on=False)
    assert _rel(dx0, leaves0[0].grad) <= tol and _rel(dg0, leaves0[1].grad) <= tol and _rel(db0, leaves0[2].grad) <= tol
    assert float(dgw0.abs().max()) == 0.0 and float(dgb0.abs().max()) == 0.0


def test_residual_block_training_against_the_reference_forward_residule_moe_gradients():
    """fp16-autocast TRAINING of the block on the library's kernels (resmoe._residual_block_train) against output AND gradients of
    the reference's own ``forward_residule_moe`` in train mode (hard gates on ``_threshold``, straight-through estimator;
    tests/golden/make_golden_resmoe.py): y, dL/dx and every parameter gradient -- LayerNorms, qkv / proj, the E = 1 expert
    (= the fixture's Mlp), both gates."""
    from slim_switch_moe_vit_amd import resmoe
    g = _fixture("ref_resblock_tiny.npz")
    blk = _block_from_fixture(g).train()
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    dy = torch.from_numpy(g["dy"]).to(DEV)
    calls = {"n": 0}
    orig = resmoe._residual_block_train

    def spy(b, t):
        calls["n"] += 1
        return orig(b, t)
    resmoe._residual_block_train = spy
    try:
        with torch.autocast("cuda", dtype=torch.float16):
            y = blk(x)
        y.backward(dy)
    finally:
        resmoe._residual_block_train = orig
    assert calls["n"] == 1, "the training block must take the own-kernel path"
    T = x.shape[0] * x.shape[1]
    for gt, key in ((blk.dense_gate, "train_dense_mask"), (blk.moe_gate, "train_moe_mask")):
        assert gt._total_tokens == T and gt._skipped_tokens == float(np.rint(g[key])[..., 0].sum()), key
    ref_y = torch.from_numpy(g["train_y"])
    err_y = float((y.detach().float().cpu() - ref_y).abs().max())
    print(f"residual block, training: max |y - reference| = {err_y:.3e} (scale {float(ref_y.abs().max()):.2f})")
    assert err_y <= 1e-3 * dtype_factor() * max(1.0, float(ref_y.abs().max()))     # measured 1.37e-3 at scale 4.16 (3.3e-4 of it): 3 x
    worst = [(_rel(x.grad, torch.from_numpy(g["train_dx"])), "x")]
    names = {"mlp.experts.htoh4.weight": "mlp.fc1.weight", "mlp.experts.htoh4.bias": "mlp.fc1.bias",
             "mlp.experts.h4toh.weight": "mlp.fc2.weight", "mlp.experts.h4toh.bias": "mlp.fc2.bias"}
    for n, p in blk.named_parameters():
        if n.startswith("mlp.gate."):      # the E = 1 router: score == 1, no gradient (the fixture's Mlp has no router)
            continue
        ref = torch.from_numpy(g["g." + names.get(n, n)]).reshape(p.shape)
        assert p.grad is not None, n
        worst.append((_rel(p.grad, ref), n))
    print("residual block, training, relative L2 gradient differences vs the reference:",
          {n: f"{e:.1e}" for e, n in sorted(worst, reverse=True)[:6]})
    assert max(worst)[0] <= 3.3e-3 * dtype_factor(), max(worst)     # measured <= 1.1e-3 (dense_gate.head.1.weight): 3 x


@pytest.mark.parametrize("E,k", [(8, 1), (8, 2), (4, 1)])
def test_zero_row_groups_change_no_result_of_the_training_operator(E, k):
    """The training path's hint ``zero_rows`` (tokens the skip gate masked) cuts the all-zero rows into row groups of their own
    (group -> expert map, rank-1 weight gradients): the output is bit for bit the output without the hint, every gradient agrees to
    summation order -- with ~45 % of the rows zero, i.e. one expert's group 4-5 x the size of the others without it."""
    d, h, T = 192, 768, 3000
    mod = _mk_moe(d, h, E, k, seed=5).train()
    g = _gen(9)
    x = torch.randn(T, d, generator=g)
    zero = torch.rand(T, generator=g) < 0.45
    x[zero] = 0.0
    res = torch.randn(T, d, generator=g).to(DEV)
    dy = (torch.randn(T, d, generator=g) * 0.1).to(DEV)

    def run(hint):
        mod.zero_grad(set_to_none=True)
        xg = x.to(DEV).requires_grad_(True)
        out = mod.forward_add(xg, res, zero_rows=zero.to(DEV) if hint else None)
        out.backward(dy)
        return out.detach(), xg.grad, {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}
    o1, dx1, g1 = run(True)
    o2, dx2, g2 = run(False)
    assert torch.equal(o1, o2) and torch.equal(dx1, dx2)
    assert set(g1) == set(g2)
    for n in g1:
        assert _rel(g1[n], g2[n]) <= 2e-3, (n, _rel(g1[n], g2[n]))


@pytest.mark.parametrize("d,with_ln", [(768, True), (192, False)])
def test_gate_kernel_writes_the_masked_f32_image_in_the_same_pass(d, with_ln):
    """``tk32`` of smoe_gate_ln_router = the (normed) f32 row with zeros for the skipped tokens: bit for bit ``xn32 * mask[:, 1:]``
    (models/resMoE.py:141 `x * mask[..., 1:]`), including the tokens the f64 redo pass decides."""
    T = 5000
    g = _gen(d + 1)
    x = torch.randn(T, d, generator=g).to(DEV)
    w, b = (torch.randn(d, generator=g) * 0.2).to(DEV), torch.zeros(1, device=DEV)
    thr = torch.tensor(0.5, device=DEV)
    ln = (torch.ones(d, device=DEV) + 0.1 * torch.randn(d, generator=g).to(DEV), 0.1 * torch.randn(d, generator=g).to(DEV), 1e-6) if with_ln else None
    r = ops.gate_ln_router(x, w, b, thr, ln=ln, want_xn32=True, want_mask=True, want_tk32=True)
    assert 0.2 < float(r["mask"][:, 0].mean()) < 0.8
    assert torch.equal(r["tk32"], r["xn32"] * r["mask"][:, 1:])
    both = ops.gate_ln_router(x, w, b, thr, ln=ln, xn16_dtype=torch.float16, want_xn32=True, want_mask=True, want_tk32=True)
    assert torch.equal(both["tk32"], r["tk32"]) and torch.equal(both["xn16"], r["tk32"].half())
