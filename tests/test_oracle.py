"""CPU tests of the oracle (oracle/moe_oracle.py): pinned against the reference's own importable code
(tests/golden/ref_*.npz, generated from /root/reference/models/layers.py) and checked for the properties
the MoE operator must have."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import moe_oracle as mo


def _load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name)).items()}


def test_expert_ffn_matches_reference_mlp(golden_dir):
    """E = 1 MoE expert == the reference's dense Mlp (models/layers.py:391-414), fp32."""
    g = _load(golden_dir, "ref_mlp_tiny.npz")
    x = torch.from_numpy(g["x"]).reshape(-1, 192)
    w1, b1 = torch.from_numpy(g["fc1_w"])[None], torch.from_numpy(g["fc1_b"])[None]
    w2, b2 = torch.from_numpy(g["fc2_w"])[None], torch.from_numpy(g["fc2_b"])[None]
    offs = np.array([0, x.shape[0]], dtype=np.int32)
    y, _ = mo.expert_ffn(x, offs, w1, b1, w2, b2)
    ref = torch.from_numpy(g["y"]).reshape(-1, 192)
    assert torch.allclose(y, ref, rtol=0, atol=2e-6), (y - ref).abs().max()


def test_moe_e1_k1_equals_reference_mlp(golden_dir):
    """whole operator with E = 1, k = 1: score == 1, permutation = identity, out == Mlp(x)."""
    g = _load(golden_dir, "ref_mlp_tiny.npz")
    x = torch.from_numpy(g["x"])
    r = mo.moe_forward(x, torch.zeros(1, 192), torch.zeros(1), torch.from_numpy(g["fc1_w"])[None],
                       torch.from_numpy(g["fc1_b"])[None], torch.from_numpy(g["fc2_w"])[None],
                       torch.from_numpy(g["fc2_b"])[None], k=1)
    assert torch.all(r.score == 1.0)
    assert np.array_equal(r.plan.pos, np.arange(x.shape[0] * x.shape[1]))
    assert torch.allclose(r.out, torch.from_numpy(g["y"]), rtol=0, atol=2e-6)


def test_layernorm_matches_reference(golden_dir):
    """F.layer_norm (used by the oracle's block) == the reference's manual LayerNorm (models/layers.py:160-224)."""
    g = _load(golden_dir, "ref_layernorm_tiny.npz")
    y = F.layer_norm(torch.from_numpy(g["x"]), (192,), torch.from_numpy(g["w"]), torch.from_numpy(g["b"]), 1e-6)
    assert torch.allclose(y, torch.from_numpy(g["y"]), rtol=0, atol=5e-6)


def test_attention_matches_reference_attention(golden_dir):
    """oracle.attention == the reference's own ``Attention`` module (models/layers.py:227-269, the computation of
    models/vision_transformer.py:248-280) on the committed inputs / parameters / outputs."""
    g = _load(golden_dir, "ref_attention_tiny.npz")
    t = lambda k: torch.from_numpy(g[k])
    for xk, yk in (("xa", "ya"), ("xb", "yb")):
        y = mo.attention(t(xk), t("qkv_w"), t("qkv_b"), t("proj_w"), t("proj_b"), int(g["num_heads"]))
        assert torch.allclose(y, t(yk), rtol=0, atol=3e-6), (y - t(yk)).abs().max()


def test_oracle_regression_vectors(golden_dir):
    g = _load(golden_dir, "oracle_moe_small.npz")
    for name in ("naive_k2", "naive_k1", "switch_cap"):
        T, k, gate, cap = [int(v) for v in g[f"{name}.meta"]]
        t = lambda key: torch.from_numpy(g[f"{name}.{key}"])
        r = mo.moe_forward(t("x"), t("wg"), t("bg"), t("w1"), t("b1"), t("w2"), t("b2"), k, gate, cap)
        assert np.array_equal(r.idx.numpy(), g[f"{name}.idx"])
        assert np.array_equal(r.plan.pos, g[f"{name}.pos"])
        assert np.array_equal(r.plan.inv_pos, g[f"{name}.inv_pos"])
        assert np.array_equal(r.plan.counts, g[f"{name}.counts"])
        assert torch.allclose(r.out, t("out"), rtol=0, atol=1e-6)


@pytest.mark.parametrize("E,k,n_tok", [(4, 1, 1), (8, 2, 257), (3, 3, 100), (16, 1, 1000)])
def test_plan_properties(E, k, n_tok):
    rng = np.random.default_rng(E * 100 + k)
    idx = rng.integers(-1, E, size=(n_tok, k))
    p = mo.dispatch_plan(idx, E)
    flat = idx.reshape(-1)
    kept = int(p.offsets[E])
    assert kept == (flat >= 0).sum() == p.counts.sum()
    # pos restricted to kept slots is a bijection onto the kept flat entries
    assert sorted(p.pos[:kept].tolist()) == np.nonzero(flat >= 0)[0].tolist()
    assert np.all(p.pos[kept:] == -1)
    for e in range(E):
        seg = p.pos[p.offsets[e]:p.offsets[e + 1]]
        assert np.all(flat[seg] == e)
        assert np.all(np.diff(seg) > 0)  # ascending flat index inside an expert (stable)
    assert np.all(p.inv_pos[p.pos[:kept]] == np.arange(kept))
    assert np.all(p.inv_pos[flat < 0] == -1)
    # brute-force restatement of assign_pos with loops
    slots = {e: [] for e in range(E)}
    for i, e in enumerate(flat):
        if e >= 0:
            slots[e].append(i)
    brute = [i for e in range(E) for i in slots[e]]
    assert brute == p.pos[:kept].tolist()


def test_capacity_prune_keeps_first_tokens():
    idx = np.array([0, 1, 0, 0, 1, 0, 2, 0], dtype=np.int64)
    p = mo.dispatch_plan(idx, 3, capacity=2)
    assert p.idx_pruned.tolist() == [0, 1, 0, -1, 1, -1, 2, -1]
    assert p.counts.tolist() == [2, 2, 1]
    assert p.pos[:5].tolist() == [0, 2, 1, 4, 6]
    assert mo.switch_capacity(1.0, 50432, 1, 8) == 6304


def test_naive_top1_score_is_exactly_one_and_ties_pick_lowest_id():
    x = torch.zeros(5, 16)  # all-zero rows: logits == bias
    wg = torch.randn(4, 16)
    bg = torch.tensor([0.5, 0.7, 0.7, 0.1])
    idx, score, _ = mo.naive_gate(x, wg, bg, 1)
    assert torch.all(score == 1.0) and torch.all(idx == 1)
    idx2, score2, _ = mo.naive_gate(x, wg, bg, 2)
    assert idx2[0].tolist() == [1, 2] and torch.allclose(score2.sum(-1), torch.ones(5))


def test_switch_gate_and_aux_loss():
    g = torch.Generator().manual_seed(3)
    x, wg = torch.randn(64, 32, generator=g), torch.randn(8, 32, generator=g)
    idx, score, p = mo.switch_gate(x, wg, None)
    assert torch.allclose(p.sum(-1), torch.ones(64), atol=1e-6)
    assert torch.equal(idx[:, 0], p.argmax(-1))
    aux = mo.switch_aux_loss(idx, p, 8)
    # perfectly uniform routing gives aux == 1; anything else is >= 1 up to sampling noise
    assert 0.9 < float(aux) < 8.0


def test_dropped_rows_are_zero_and_moe_is_rowwise():
    g = torch.Generator().manual_seed(5)
    d, h, E = 32, 64, 4
    x = torch.randn(50, d, generator=g)
    wg, bg = torch.randn(E, d, generator=g), torch.zeros(E)
    w1, b1 = torch.randn(E, h, d, generator=g) * 0.1, torch.randn(E, h, generator=g) * 0.1
    w2, b2 = torch.randn(E, d, h, generator=g) * 0.1, torch.randn(E, d, generator=g) * 0.1
    r = mo.moe_forward(x, wg, bg, w1, b1, w2, b2, 1, mo.GATE_SWITCH, capacity=5)
    dropped = torch.from_numpy(r.plan.idx_pruned < 0)
    assert dropped.any() and torch.all(r.out[dropped] == 0)
    # row-wise: a kept token's output only depends on itself
    t = int(torch.nonzero(~dropped)[0])
    e = int(r.idx[t, 0])
    y = F.gelu(x[t] @ w1[e].t() + b1[e]) @ w2[e].t() + b2[e]
    assert torch.allclose(r.out[t], r.score[t, 0] * y, atol=1e-5)


def test_block_with_disabled_gates_equals_stock_block_on_normed_residual():
    """forward_residule_moe with both Gates disabled == attn/moe residual taken from the normed x (resMoE.py:126-145)."""
    g = torch.Generator().manual_seed(7)
    d, heads, E, k = 32, 4, 4, 2
    p = {
        "norm1.weight": torch.ones(d), "norm1.bias": torch.zeros(d), "norm2.weight": torch.ones(d),
        "norm2.bias": torch.zeros(d),
        "attn.qkv.weight": torch.randn(3 * d, d, generator=g) * 0.1, "attn.qkv.bias": torch.zeros(3 * d),
        "attn.proj.weight": torch.randn(d, d, generator=g) * 0.1, "attn.proj.bias": torch.zeros(d),
        "mlp.gate.gate.weight": torch.randn(E, d, generator=g), "mlp.gate.gate.bias": torch.zeros(E),
        "mlp.experts.htoh4.weight": torch.randn(E, 4 * d, d, generator=g) * 0.1,
        "mlp.experts.htoh4.bias": torch.zeros(E, 4 * d),
        "mlp.experts.h4toh.weight": torch.randn(E, d, 4 * d, generator=g) * 0.1,
        "mlp.experts.h4toh.bias": torch.zeros(E, d),
        "dense_gate.head.1.weight": torch.randn(1, d, generator=g), "dense_gate.head.1.bias": torch.zeros(1),
        "dense_gate.threshold": torch.tensor(0.9), "dense_gate.disable": True,
        "moe_gate.head.1.weight": torch.randn(1, d, generator=g), "moe_gate.head.1.bias": torch.zeros(1),
        "moe_gate.threshold": torch.tensor(0.9), "moe_gate.disable": True,
    }
    x = torch.randn(2, 9, d, generator=g)
    y = mo.block_forward(x, p, heads, k, residual_moe=True)
    n1 = F.layer_norm(x, (d,), eps=1e-6)
    a = mo.attention(n1, p["attn.qkv.weight"], p["attn.qkv.bias"], p["attn.proj.weight"], p["attn.proj.bias"], heads) + n1
    n2 = F.layer_norm(a, (d,), eps=1e-6)
    m = mo.moe_forward(n2, p["mlp.gate.gate.weight"], p["mlp.gate.gate.bias"], p["mlp.experts.htoh4.weight"],
                       p["mlp.experts.htoh4.bias"], p["mlp.experts.h4toh.weight"], p["mlp.experts.h4toh.bias"], k).out
    assert torch.allclose(y, m + n2, atol=1e-6)
    # with an active gate and threshold 0, every token is skipped: MoE sees all-zero rows (SURVEY 'zero-token skew')
    p["moe_gate.disable"] = False
    p["moe_gate.threshold"] = torch.tensor(0.0)
    y2 = mo.block_forward(x, p, heads, k, residual_moe=True)
    z = mo.moe_forward(torch.zeros_like(n2), p["mlp.gate.gate.weight"], p["mlp.gate.gate.bias"],
                       p["mlp.experts.htoh4.weight"], p["mlp.experts.htoh4.bias"], p["mlp.experts.h4toh.weight"],
                       p["mlp.experts.h4toh.bias"], k).out
    assert torch.allclose(y2, z + n2, atol=1e-6)


def test_fp64_router_agrees_with_fp32_linear_except_near_ties():
    g = torch.Generator().manual_seed(11)
    x, wg = torch.randn(20000, 192, generator=g), torch.randn(8, 192, generator=g) * 0.02
    idx, _, logits = mo.naive_gate(x, wg, None, 1)
    ref = F.linear(x, wg)
    assert torch.allclose(logits, ref, atol=1e-5)
    assert (idx[:, 0] == ref.argmax(-1)).float().mean() > 0.9995


# ---- the reference's own Gate / forward_residule_moe (tests/golden/make_golden_resmoe.py runs models/resMoE.py:32-85,126-145) ----
F32_EPS = float(np.finfo(np.float32).eps)


def _explained_flips(ours_skip, ref_mask, prob_f32, thr):
    """Tokens whose skip decision differs from the reference's.  The oracle (and the HIP gate) decide on the float64 logit
    against logit(thr); the reference compares sigmoid_f32(z_f32) > thr.  Every differing token must sit within float32
    rounding of the threshold: |sigmoid_f32(z) - thr| <= 4 ulp of thr (dot-product rounding moves z by ~1e-7 relative, the
    sigmoid by at most a quarter of that).  Returns the list of (token, prob - thr)."""
    ref_skip = np.rint(ref_mask[..., 0]).astype(bool).reshape(-1)
    differ = np.nonzero(ours_skip.reshape(-1) != ref_skip)[0]
    gaps = [(int(t), float(prob_f32.reshape(-1)[t]) - float(np.float32(thr))) for t in differ]
    for t, gap in gaps:
        assert abs(gap) <= 4 * F32_EPS * max(float(thr), 0.25), (t, gap)
    return gaps


def test_skip_gate_against_the_reference_gate_fixture(golden_dir):
    """oracle.skip_gate vs outputs of the reference's own ``Gate`` (models/resMoE.py:59-85) in eval (threshold), train-hard
    (_threshold) and disabled mode.  Two stated deviations, now measured against the real thing: (1) decision on the f64 logit
    -- differs only for tokens within f32 rounding of the threshold (each listed and bounded); (2) masks exactly 0 / 1 -- the
    reference's ``(p > thr).float() + (1 - p).detach() - (1 - p)`` is within one f32 ulp of that."""
    g = _load(golden_dir, "ref_gate_tiny.npz")
    x, w, b = torch.from_numpy(g["x"]), torch.from_numpy(g["w"]), torch.from_numpy(g["b"])
    n_flips = 0
    for mode, thr in (("eval", g["thr_eval"]), ("train_hard", g["thr_train"])):
        ref = g[f"{mode}_mask"]
        m = mo.skip_gate(x, w, b, float(thr)).numpy()
        assert np.abs(ref - np.rint(ref)).max() <= F32_EPS, "the reference's hard masks are 0 / 1 to one ulp"
        flips = _explained_flips(m[..., 0] > 0.5, ref, g["prob_f32"], thr)
        n_flips += len(flips)
        same = np.ones(m.shape[:2], dtype=bool).reshape(-1)
        same[[t for t, _ in flips]] = False
        assert np.array_equal(m.reshape(-1, 2)[same], np.rint(ref).reshape(-1, 2)[same])
        # counters: the reference sums its masks (resMoE.py:83); up to the flips that is the number of skipped tokens
        assert abs(float(g[f"{mode}_skipped"]) - float(m[..., 0].sum())) <= len(flips) + 1e-3
        assert int(g[f"{mode}_total"]) == x.shape[0] * x.shape[1]
        print(f"{mode}: {int(m[..., 0].sum())} skipped, {len(flips)} decisions differ from the reference: {flips}")
    assert np.array_equal(mo.skip_gate(x, w, b, 0.5, disable=True).numpy(), g["disabled_mask"])
    near = np.concatenate([g["near_rows_eval"], g["near_rows_train"]])
    assert len(near) == 34 and n_flips <= len(near), "only engineered near-threshold rows may flip"


def _resblock_params(g):
    """The fixture's Holder parameters under the names oracle.block_forward reads; the dense ``Mlp`` becomes the E = 1 expert."""
    p = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("p.")}
    d = p["norm1.weight"].shape[0]
    p["mlp.gate.gate.weight"], p["mlp.gate.gate.bias"] = torch.zeros(1, d), torch.zeros(1)
    p["mlp.experts.htoh4.weight"], p["mlp.experts.htoh4.bias"] = p.pop("mlp.fc1.weight")[None], p.pop("mlp.fc1.bias")[None]
    p["mlp.experts.h4toh.weight"], p["mlp.experts.h4toh.bias"] = p.pop("mlp.fc2.weight")[None], p.pop("mlp.fc2.bias")[None]
    return p


def test_block_forward_residual_against_the_reference_fixture(golden_dir):
    """oracle.block_forward(residual_moe=True) vs the output of the reference's own ``forward_residule_moe``
    (models/resMoE.py:126-145) on a block of reference modules (LayerNorm, layers.Attention, layers.Mlp = E = 1 MoE, two Gates
    skipping 40-50 % of the tokens): same skip decisions on every token, same activations to f32 rounding."""
    g = _load(golden_dir, "ref_resblock_tiny.npz")
    p = _resblock_params(g)
    x = torch.from_numpy(g["x"])
    y = mo.block_forward(x, p, int(g["num_heads"]), 1, residual_moe=True)
    ref = torch.from_numpy(g["eval_y"])
    # the decisions the oracle took, recomputed the way block_forward does
    d = x.shape[-1]
    xn = F.layer_norm(x, (d,), p["norm1.weight"], p["norm1.bias"], 1e-6)
    m1 = mo.skip_gate(xn, p["dense_gate.head.1.weight"], p["dense_gate.head.1.bias"], float(p["dense_gate.threshold"]))
    assert np.array_equal(m1.numpy(), np.rint(g["eval_dense_mask"])), "dense gate: every decision equals the reference's"
    assert 0.3 <= float(m1[..., 0].mean()) <= 0.7
    assert np.array_equal(np.rint(g["eval_moe_mask"])[..., 0] + np.rint(g["eval_moe_mask"])[..., 1], np.ones(x.shape[:2]))
    assert torch.allclose(y, ref, rtol=0, atol=2e-5), float((y - ref).abs().max())


def test_forced_routing_is_the_identity_on_the_oracles_own_decisions_and_traces_flips():
    """``forced_idx`` (the bench's same-routing comparison): forcing the oracle's own decisions changes nothing; forcing another
    expert for some tokens changes exactly those tokens' rows, and the trace reports them with the logit gap they gave up."""
    g = torch.Generator().manual_seed(4)
    T, d, h, E = 300, 32, 64, 4
    x = torch.randn(T, d, generator=g)
    wg, bg = torch.randn(E, d, generator=g) * 0.3, torch.zeros(E)
    w1, b1 = torch.randn(E, h, d, generator=g) * 0.1, torch.zeros(E, h)
    w2, b2 = torch.randn(E, d, h, generator=g) * 0.1, torch.zeros(E, d)
    own = mo.moe_forward(x, wg, bg, w1, b1, w2, b2, 1)
    tr = []
    same = mo.moe_forward(x, wg, bg, w1, b1, w2, b2, 1, forced_idx=own.idx, trace=tr)
    assert torch.equal(same.out, own.out) and tr == [{"tokens": T, "flips": 0, "max_margin": 0.0}]
    forced = own.idx.clone()
    forced[::50] = (forced[::50] + 1) % E
    tr = []
    other = mo.moe_forward(x, wg, bg, w1, b1, w2, b2, 1, forced_idx=forced, trace=tr)
    changed = (other.out != own.out).any(1)
    assert changed[::50].all() and int(changed.sum()) == len(range(0, T, 50))
    assert tr[0]["flips"] == len(range(0, T, 50)) and tr[0]["max_margin"] > 0
    logits = mo.router_logits(x, wg, bg)
    gap = (logits.max(1).values - logits.gather(1, forced).squeeze(1))[::50].max()
    assert abs(tr[0]["max_margin"] - float(gap)) < 1e-6


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_committed_fixtures_regenerate_bit_identically(tmp_path):
    """The .npz files under tests/golden/ that pin the oracle are OUTPUTS OF THE REFERENCE'S OWN CODE (models/layers.py Mlp / LayerNorm /
    Attention; models/resMoE.py Gate / forward_residule_moe) as the committed scripts produce them: re-run both scripts into a scratch
    directory and require every array of every committed fixture back bit for bit (same keys, dtypes, shapes, bytes).  A fixture edited
    by hand, or a script that drifted from the data it is said to have made, fails here."""
    import subprocess
    import sys
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    env = dict(os.environ, SLIMMOE_GOLDEN_OUT=str(tmp_path))
    for script in ("make_golden.py", "make_golden_resmoe.py"):
        r = subprocess.run([sys.executable, os.path.join(golden, script)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (script, r.stderr[-800:])
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert made == sorted(f for f in os.listdir(golden) if f.endswith(".npz")), made
    for f in made:
        new, old = np.load(os.path.join(tmp_path, f)), np.load(os.path.join(golden, f))
        assert sorted(new.files) == sorted(old.files), f
        for key in old.files:
            a, b = old[key], new[key]
            assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), (f, key)
