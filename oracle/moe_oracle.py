"""CPU oracle for the Switch-MoE ViT hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped path (``slim-switch-moe-vit_amd``) never routes through it and
raises if the HIP extension is missing.

What it restates
----------------
The reference (d0-rb/slim-switch-moe-vit) holds no MoE arithmetic of its own:
``models/resMoE.py:6,15-29`` subclasses FastMoE's ``FMoETransformerMLP``
(third-party ``fmoe``, un-vendored, un-pinned, CUDA-only, absent from the
reference tree and from this image).  The operator semantics below follow
SURVEY.md Appendix B (the published FastMoE algorithm: NaiveGate / SwitchGate,
count_by_gate, assign_pos, MOEScatter, per-expert FMoELinear pair, MOEGather,
bmm-combine) anchored on the reference's own call sites:

* ``models/resMoE.py:25-29``   activation = GELU -> Dropout, naive gate, top_k
* ``models/resmoe_flop_hook.py:7-8``  ``mlp.gate.gate`` is ``nn.Linear(d, E)`` + softmax
* ``models/layers.py:391-414``  dense ``Mlp`` = what one expert computes (E=1 known answer)
* ``models/resMoE.py:32-85``   token-skip ``Gate``
* ``models/resMoE.py:126-145`` ``forward_residule_moe`` block wrapper

PARITY STATUS, in two parts.
(1) PINNED by outputs of the reference's own code -- everything whose arithmetic lives
IN the reference tree, run in the build container (generators committed beside the
data in ``tests/golden/``; ``tests/test_oracle.py::test_committed_fixtures_regenerate_
bit_identically`` re-runs them and compares every byte):

* the per-expert FFN, LayerNorm and attention against ``models/layers.py``
  (``make_golden.py`` -> ``ref_mlp_tiny / ref_layernorm_tiny / ref_attention_tiny.npz``);
* the token-skip ``Gate`` (eval / train-hard / train-soft / disabled, with
  gradients, rows placed on the threshold) and ``forward_residule_moe`` (eval and
  train-hard with gradients) against ``models/resMoE.py:32-85, 126-145`` itself
  (``make_golden_resmoe.py`` compiles those two definitions out of the file's
  syntax tree -- they use nothing of fmoe / timm -- ->
  ``ref_gate_tiny.npz``, ``ref_resblock_tiny.npz``).  Measured deviation of
  ``skip_gate`` below from the reference: of 34 rows engineered within 1e-4 of the
  threshold in logit space, 3 (eval) / 5 (train) decide differently, each with
  ``|sigmoid_f32(z) - thr| <= 1 ulp``; no other token differs.

(2) **parity unpinned** for FastMoE's part of the operator ONLY -- gate top-k order and
tie-break, slot order, capacity, aux loss: the reference has no tests, fixtures or
golden vectors, and FastMoE (third-party, absent from the tree and from this image)
cannot run here.  That part is a restatement of the published algorithm (SURVEY.md
Appendix B) with the determinism decisions below.

Determinism decisions (normative; upstream leaves these to atomic races)
-----------------------------------------------------------------------
* router logits accumulate in float64 (exact products of float32 inputs) and
  are rounded once to float32, so routing does not depend on summation order;
* top-k tie-break: lowest expert id; the k picks are ordered by descending logit;
* slot order inside an expert bucket: ascending flat index ``t*k + j``;
* capacity: a flat entry is kept iff its rank among same-expert entries (in
  ascending flat index) is ``< cap``; dropped entries get ``idx = -1``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

GATE_NAIVE = 0
GATE_SWITCH = 1


# --------------------------------------------------------------------------- router
def router_logits(x: torch.Tensor, wg: torch.Tensor, bg: Optional[torch.Tensor]) -> torch.Tensor:
    """``logits = x @ Wg^T + bg`` (NaiveGate's ``nn.Linear(d, E)``; resmoe_flop_hook.py:7).

    float64 accumulation, one rounding to float32 (see module docstring)."""
    acc = x.double() @ wg.double().t()
    if bg is not None:
        acc = acc + bg.double()
    return acc.float()


def topk_lowest_id(logits: torch.Tensor, k: int):
    """top-k by repeated first-argmax: ties -> lowest expert id, order = descending logit."""
    work = logits.clone()
    vals, idxs = [], []
    for _ in range(k):
        i = torch.argmax(work, dim=-1)  # first maximal index on CPU
        v = work.gather(-1, i[:, None])[:, 0]
        vals.append(v)
        idxs.append(i)
        work.scatter_(-1, i[:, None], float("-inf"))
    return torch.stack(vals, -1), torch.stack(idxs, -1)


def naive_gate(x, wg, bg, k):
    """FastMoE NaiveGate: topk(logits, k) then softmax over the k kept logits only
    (k = 1 => score == 1.0 exactly).  SURVEY.md Appendix B, A3."""
    logits = router_logits(x, wg, bg)
    val, idx = topk_lowest_id(logits, k)
    score = torch.softmax(val, dim=-1)
    return idx.to(torch.int64), score, logits


def switch_gate(x, wg, bg, noise: Optional[torch.Tensor] = None):
    """FastMoE SwitchGate (k = 1): p = softmax(float32(logits [+ noise])) over all E,
    (score, idx) = top-1.  ``noise`` is the additive train-time jitter, passed in so
    oracle and device see the same numbers.  SURVEY.md A9."""
    logits = router_logits(x, wg, bg)
    if noise is not None:
        logits = logits + noise.float()
    p = torch.softmax(logits, dim=-1)
    idx = torch.argmax(logits, dim=-1, keepdim=True)  # first max -> lowest id
    score = p.gather(-1, idx)
    return idx.to(torch.int64), score, p


def switch_capacity(cf: float, T_local: int, k: int, E: int) -> int:
    """Switch-Transformer capacity per (source rank, expert): ceil(cf*T_local*k/E)."""
    return int(math.ceil(cf * T_local * k / E))


def switch_aux_loss(idx_pruned: torch.Tensor, probs: torch.Tensor, E: int) -> torch.Tensor:
    """aux = E * sum_e frac_e * prob_e ; frac_e = share of kept tokens on e,
    prob_e = sum_t p[t,e] / kept.  SURVEY.md A9."""
    flat = idx_pruned.reshape(-1)
    kept = (flat >= 0).sum().clamp(min=1).to(probs.dtype)
    frac = torch.bincount(flat[flat >= 0], minlength=E).to(probs.dtype) / kept
    prob = probs.sum(0) / kept
    return E * (frac * prob).sum()


# --------------------------------------------------------------------------- plan
@dataclass
class Plan:
    counts: np.ndarray      # int32 [E]   kept entries per expert
    offsets: np.ndarray     # int32 [E+1] exclusive prefix
    pos: np.ndarray         # int64 [n]   slot -> flat index (t*k+j); tail beyond kept = -1
    inv_pos: np.ndarray     # int64 [n]   flat index -> slot, -1 if dropped
    idx_pruned: np.ndarray  # int64 [n]   expert id or -1


def dispatch_plan(idx, E: int, capacity: int = -1) -> Plan:
    """count_by_gate + assign_pos (+ prune_gate_by_capacity), deterministic.
    Integer work: plain numpy.  SURVEY.md A4/A9, N1/N2/N9."""
    flat = np.asarray(idx, dtype=np.int64).reshape(-1).copy()
    n = flat.size
    if capacity is not None and capacity >= 0:
        seen = np.zeros(E, dtype=np.int64)
        for i in range(n):  # ascending flat index
            e = flat[i]
            if e < 0:
                continue
            if seen[e] >= capacity:
                flat[i] = -1
            else:
                seen[e] += 1
    valid = flat >= 0
    counts = np.bincount(flat[valid], minlength=E).astype(np.int32)
    offsets = np.zeros(E + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(counts)
    order = np.argsort(np.where(valid, flat, E), kind="stable")  # stable: ascending flat idx inside a bucket
    kept = int(offsets[E])
    pos = np.full(n, -1, dtype=np.int64)
    pos[:kept] = order[:kept]
    inv = np.full(n, -1, dtype=np.int64)
    inv[pos[:kept]] = np.arange(kept, dtype=np.int64)
    return Plan(counts, offsets, pos, inv, flat)


# --------------------------------------------------------------------------- experts
def gelu_erf(h: torch.Tensor) -> torch.Tensor:
    """nn.GELU() default = exact erf form (resMoE.py:23 act_layer=th.nn.GELU)."""
    return F.gelu(h)


def expert_ffn(R, offsets, w1, b1, w2, b2, act=gelu_erf, dtype=torch.float32):
    """per expert e on its contiguous slice: Y = act(R W1[e]^T + b1[e]) W2[e]^T + b2[e]
    (FMoELinear pair; dense equivalent models/layers.py:408-414).  w1 [E,h,d], w2 [E,d,h]."""
    E = w1.shape[0]
    Y = torch.zeros(R.shape[0], w2.shape[1], dtype=dtype)
    H = torch.zeros(R.shape[0], w1.shape[1], dtype=dtype)
    for e in range(E):
        lo, hi = int(offsets[e]), int(offsets[e + 1])
        if hi == lo:
            continue
        h = R[lo:hi].to(dtype) @ w1[e].to(dtype).t()
        if b1 is not None:
            h = h + b1[e].to(dtype)
        a = act(h)
        H[lo:hi] = a
        y = a @ w2[e].to(dtype).t()
        if b2 is not None:
            y = y + b2[e].to(dtype)
        Y[lo:hi] = y
    return Y, H


# --------------------------------------------------------------------------- whole operator
@dataclass
class MoEOut:
    out: torch.Tensor
    idx: torch.Tensor
    score: torch.Tensor
    plan: Plan
    y_sorted: torch.Tensor
    h_sorted: torch.Tensor
    logits_or_probs: torch.Tensor
    aux_loss: Optional[torch.Tensor] = None


def moe_forward(x, wg, bg, w1, b1, w2, b2, k: int, gate: int = GATE_NAIVE,
                capacity: int = -1, noise=None, dtype=torch.float32, forced_idx=None, trace=None) -> MoEOut:
    """FMoETransformerMLP.forward on one rank (world_size = 1).  SURVEY.md Appendix B.

    ``forced_idx`` (int64 [T, k], NaiveGate only; a checking aid, not part of the operator): route as THESE decisions say instead
    of the oracle's own top-k -- the score is still the softmax of the oracle's logits at the forced experts.  With the routing of a
    reduced-precision implementation forced, what is left of the difference between the two is arithmetic alone; ``trace`` (a list)
    receives per call {"tokens", "flips": entries routed differently from the oracle's own choice, "max_margin": the largest gap
    (own best logit - logit of the forced expert) among them}: a flip is explained by precision iff its margin is tiny."""
    shape = x.shape
    d = shape[-1]
    X = x.reshape(-1, d).float()
    T = X.shape[0]
    E = wg.shape[0]
    if gate == GATE_NAIVE:
        idx, score, aux_in = naive_gate(X, wg, bg, k)
        if forced_idx is not None:
            logits = router_logits(X, wg, bg)
            f = forced_idx.reshape(T, k).to(torch.int64)
            if trace is not None:
                own, forced = logits.gather(1, idx), logits.gather(1, f)
                diff = (torch.sort(idx, 1).values != torch.sort(f, 1).values).any(1)
                margin = (own.max(1).values - forced.min(1).values)[diff]
                trace.append({"tokens": T, "flips": int(diff.sum()), "max_margin": float(margin.max()) if margin.numel() else 0.0})
            idx, score = f, torch.softmax(logits.gather(1, f), dim=-1)
    else:
        assert k == 1 and forced_idx is None
        idx, score, aux_in = switch_gate(X, wg, bg, noise)
    plan = dispatch_plan(idx.numpy(), E, capacity)
    kept = int(plan.offsets[E])
    pos = torch.from_numpy(plan.pos[:kept])
    S = X[pos // k]                                           # MOEScatter (N3)
    Y, H = expert_ffn(S, plan.offsets, w1, b1, w2, b2, dtype=dtype)
    Z = torch.zeros(T * k, d, dtype=dtype)                    # MOEGather (N6): dropped rows stay 0
    Z[pos] = Y
    out = torch.bmm(score.to(dtype).view(T, 1, k), Z.view(T, k, d)).reshape(T, d)   # N7
    aux = None
    if gate == GATE_SWITCH:
        aux = switch_aux_loss(torch.from_numpy(plan.idx_pruned), aux_in, E)
    return MoEOut(out.reshape(shape).to(dtype), idx, score, plan, Y, H, aux_in, aux)


# --------------------------------------------------------------------------- expert-parallel restatement
def ep_reference_forward(xs, wg, bg, w1, b1, w2, b2, k, gate=GATE_NAIVE, capacity=-1):
    """World of W ranks simulated in one process.  ``xs`` = list of per-rank [T_w, d];
    experts are partitioned contiguously (global id = w*E_local + e); every rank
    routes its own tokens; capacity (if any) applies per source rank.  Because each
    token's expert output depends only on (token, expert), the result equals the
    single-rank operator applied to each rank's tokens with all E experts."""
    return [moe_forward(x, wg, bg, w1, b1, w2, b2, k, gate, capacity) for x in xs]


# --------------------------------------------------------------------------- token-skip gate + block
def skip_logit_threshold(threshold: float) -> float:
    """logit(threshold) in float64: ``sigmoid(z) > thr  <=>  z > log(thr / (1 - thr))``; +inf for thr >= 1 (nothing is
    skipped: a sigmoid never exceeds 1), -inf for thr <= 0 (everything is)."""
    t = float(threshold)
    if t >= 1.0:
        return math.inf
    if t <= 0.0:
        return -math.inf
    return math.log(t / (1.0 - t))


def skip_gate(x, w, b, threshold: float, disable: bool = False):
    """Gate.forward, eval / hard branch (resMoE.py:59-85): mask[...,0]=skip, mask[...,1]=keep.
    In value (the straight-through terms cancel): skip = (sigmoid(z) > thr), keep = 1 - skip, z = head.1(x).
    Determinism decision (as for the router): the comparison is made on the float64-accumulated logit against
    logit(thr), so it does not depend on summation order; the reference compares the float32 sigmoid, which differs
    only for tokens within float32 rounding of the threshold."""
    B, N, _ = x.shape
    if disable:
        m = torch.zeros(B, N, 2)
        m[:, :, 1] = 1
        return m
    z = x.double() @ w.double().t() + (b.double() if b is not None else 0.0)   # [B,N,1]
    skip = (z > skip_logit_threshold(threshold)).float()
    return torch.cat([skip, 1.0 - skip], dim=-1)


def attention(x, qkv_w, qkv_b, proj_w, proj_b, num_heads):
    """vision_transformer.py:248-280."""
    B, N, C = x.shape
    qkv = F.linear(x, qkv_w, qkv_b).reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * ((C // num_heads) ** -0.5)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, proj_w, proj_b)


def block_forward(x, p: dict, num_heads: int, k: int, residual_moe: bool, eps: float = 1e-6, forced_idx=None, trace=None):
    """One ViT block with the MoE MLP.  ``residual_moe`` selects forward_residule_moe
    (resMoE.py:126-145: residual taken from the *normed* activations, token-skip gates)
    versus the stock Block.forward (vision_transformer.py:319-322)."""
    d = x.shape[-1]

    def moe(t):
        return moe_forward(t, p["mlp.gate.gate.weight"], p["mlp.gate.gate.bias"],
                           p["mlp.experts.htoh4.weight"], p["mlp.experts.htoh4.bias"],
                           p["mlp.experts.h4toh.weight"], p["mlp.experts.h4toh.bias"], k, forced_idx=forced_idx,
                           trace=trace).out

    def attn(t):
        return attention(t, p["attn.qkv.weight"], p["attn.qkv.bias"], p["attn.proj.weight"],
                         p["attn.proj.bias"], num_heads)

    if not residual_moe:
        x = x + attn(F.layer_norm(x, (d,), p["norm1.weight"], p["norm1.bias"], eps))
        x = x + moe(F.layer_norm(x, (d,), p["norm2.weight"], p["norm2.bias"], eps))
        return x
    x = F.layer_norm(x, (d,), p["norm1.weight"], p["norm1.bias"], eps)
    m = skip_gate(x, p["dense_gate.head.1.weight"], p["dense_gate.head.1.bias"],
                  float(p["dense_gate.threshold"]), p.get("dense_gate.disable", False))
    skip_tk, tk = x * m[:, :, 0:1], x * m[:, :, 1:2]
    x = attn(tk) + tk + skip_tk
    x = F.layer_norm(x, (d,), p["norm2.weight"], p["norm2.bias"], eps)
    m = skip_gate(x, p["moe_gate.head.1.weight"], p["moe_gate.head.1.bias"],
                  float(p["moe_gate.threshold"]), p.get("moe_gate.disable", False))
    skip_tk, tk = x * m[:, :, 0:1], x * m[:, :, 1:2]
    x = moe(tk) + tk + skip_tk
    return x


def vit_forward(images, sd: dict, depth: int, num_heads: int, k: int, residual_moe: bool,
                patch: int = 16, eps: float = 1e-6, forced=None, trace=None):
    """VisionTransformer.forward (vision_transformer.py:818-848), non-distilled, eval mode,
    from a flat state-dict ``sd`` with the reference's key layout.
    ``forced`` (a list of per-block routing decisions) / ``trace``: see moe_forward's ``forced_idx``."""
    x = F.conv2d(images, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd["cls_token"].expand(x.shape[0], -1, -1), x), dim=1)
    x = x + sd["pos_embed"]
    for i in range(depth):
        pre = f"blocks.{i}."
        p = {key[len(pre):]: v for key, v in sd.items() if key.startswith(pre)}
        x = block_forward(x, p, num_heads, k, residual_moe, eps, forced_idx=None if forced is None else forced[i], trace=trace)
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), sd["norm.weight"], sd["norm.bias"], eps)
    return F.linear(x[:, 0], sd["head.weight"], sd["head.bias"])


# --------------------------------------------------------------------------- differentiable restatement (cfg 5)
def moe_forward_diff(x, wg, bg, w1, b1, w2, b2, k: int, gate: int = GATE_NAIVE, capacity: int = -1, noise=None,
                     dtype=torch.float64):
    """Same operator written with differentiable torch ops (routing decided under no_grad exactly as
    moe_forward does) so that torch.autograd yields reference gradients for x, the router and the experts
    (SURVEY.md Appendix B 'backward').  Returns (out, aux_loss or None, plan)."""
    T, d = x.shape
    E = wg.shape[0]
    with torch.no_grad():
        if gate == GATE_NAIVE:
            idx, _, _ = naive_gate(x.detach().float(), wg.detach().float(), None if bg is None else bg.detach().float(), k)
        else:
            idx, _, _ = switch_gate(x.detach().float(), wg.detach().float(), None if bg is None else bg.detach().float(), noise)
        plan = dispatch_plan(idx.numpy(), E, capacity)
    kept = int(plan.offsets[E])
    pos = torch.from_numpy(plan.pos[:kept])
    logits = x.to(dtype) @ wg.to(dtype).t() + (bg.to(dtype) if bg is not None else 0)
    aux = None
    if gate == GATE_NAIVE:
        score = torch.softmax(logits.gather(1, idx), dim=-1)
    else:
        if noise is not None:
            logits = logits + noise.to(dtype)
        p = torch.softmax(logits, dim=-1)
        score = p.gather(1, idx)
        aux = switch_aux_loss(torch.from_numpy(plan.idx_pruned), p, E)
    S = x.to(dtype)[pos // k]
    ys = []
    for e in range(E):
        lo, hi = int(plan.offsets[e]), int(plan.offsets[e + 1])
        hid = F.gelu(S[lo:hi] @ w1[e].to(dtype).t() + b1[e].to(dtype))
        ys.append(hid @ w2[e].to(dtype).t() + b2[e].to(dtype))
    Y = torch.cat(ys, 0) if ys else S.new_zeros(0, d)
    Z = torch.zeros(T * k, d, dtype=dtype).index_copy(0, pos, Y)
    out = (score.to(dtype).unsqueeze(-1) * Z.view(T, k, d)).sum(1)
    return out, aux, plan
