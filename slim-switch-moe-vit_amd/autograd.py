"""Training side of the MoE operator (BASELINE cfg 5: capacity + token dropping + aux loss, fwd + bwd).

Forward keeps what backward needs (expert-sorted inputs S, pre-activations H, activations A, expert outputs
Y when the gate score carries gradient); backward is the adjoint chain of SURVEY.md Appendix B:

    dY  = score * dout[pos // k]                      smoe_scatter_rows(scale=score)
    dsc = <dout[t], Y[inv_pos]>                       smoe_rowdot
    dH  = (dY W2) * gelu'(H)                          smoe_grouped_gemm(W2^T shadow, SMOE_EPI_GELU_GRAD)
    dW2 = dY_e^T A_e, db2 = colsum(dY_e)              smoe_transpose_pad x2 + smoe_grouped_wgrad, smoe_group_colsum
    dW1 = dH_e^T S_e, db1 = colsum(dH_e)              same
    dS  = dH W1                                       smoe_grouped_gemm(W1^T shadow)
    dx  = sum_j dS[inv_pos[t k + j]]                  smoe_gather_combine(score = 1)

The router's own gradient (through the gate score and the aux loss) is a skinny [T, E] computation; it runs as
ordinary differentiable torch ops on logits recomputed from x, with the ROUTING (idx) taken from the HIP router.
Expert-parallel training (world_size > 1) is not built yet.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


def switch_aux_loss(idx_pruned: torch.Tensor, probs: torch.Tensor, E: int) -> torch.Tensor:
    """aux = E * sum_e frac_e * prob_e: frac_e = share of kept tokens routed to e,
    prob_e = sum_t p[t,e] / kept.  Small [E]-sized reduction; differentiable w.r.t. ``probs``."""
    flat = idx_pruned.reshape(-1)
    keep = flat >= 0
    kept = keep.sum().clamp(min=1).to(probs.dtype)
    frac = torch.bincount(torch.where(keep, flat, torch.zeros_like(flat)), weights=keep.to(probs.dtype),
                          minlength=E)[:E] / kept
    prob = probs.sum(0) / kept
    return E * (frac * prob).sum()


class _ExpertFFN(torch.autograd.Function):
    """x [T,d], score [T,k] -> out [T,d] through the grouped expert FFN; plan tensors are constants."""

    @staticmethod
    def forward(ctx, x, score, w1, b1, w2, b2, mod, offsets, pos, inv_pos, drop_mask):
        k, d = mod.top_k, mod.d_model
        T = x.shape[0]
        cd = mod.compute_dtype or _default_cd()
        ex = mod.experts
        w1c, w2c = ex.htoh4.weight_as(cd), ex.h4toh.weight_as(cd)
        S = ops.scatter_rows(x, pos, k, cd, zero_fill=True)
        Hp = ops.grouped_gemm(S, w1c, b1.detach().float() if b1 is not None else None, offsets, ops.EPI_NONE, cd,
                              variant=mod.gemm_variant)
        A = ops.gelu(Hp)
        if drop_mask is not None:
            A = A * drop_mask
        Y = ops.grouped_gemm(A, w2c, b2.detach().float() if b2 is not None else None, offsets, ops.EPI_NONE, cd,
                             variant=mod.gemm_variant)
        out = ops.gather_combine(Y, inv_pos, score.detach().float().contiguous(), T, k, x.dtype)
        ctx.mod, ctx.cd = mod, cd
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        ctx.save_for_backward(S, Hp, A, Y, score.detach(), offsets, pos, inv_pos, drop_mask if drop_mask is not None else torch.empty(0))
        return out

    @staticmethod
    def backward(ctx, dout):
        S, Hp, A, Y, score, offsets, pos, inv_pos, drop_mask = ctx.saved_tensors
        mod, cd = ctx.mod, ctx.cd
        k, d, h, E = mod.top_k, mod.d_model, mod.d_hidden, offsets.numel() - 1
        T = dout.shape[0]
        n = pos.numel()
        dout = dout.contiguous()
        ex = mod.experts
        sc = score.float().contiguous().reshape(-1)
        dY = ops.scatter_rows(dout, pos, k, cd, zero_fill=True, scale=sc)               # [n, d]
        dscore = ops.rowdot(dout, Y, inv_pos, k).view(T, k) if ctx.needs_input_grad[1] else None
        w2t = ex.h4toh.weight_as(cd).transpose(1, 2).contiguous()                        # [E, h, d]  (N = h, K = d)
        if drop_mask.numel():
            dA = ops.grouped_gemm(dY, w2t, None, offsets, ops.EPI_NONE, cd, variant=mod.gemm_variant)
            dA = dA * drop_mask
            # gelu'(H) unfused on this (rare) path
            hp = Hp.float()
            cdf = 0.5 * (1 + torch.erf(hp * 0.7071067811865476))
            dH = (dA.float() * (cdf + hp * torch.exp(-0.5 * hp * hp) * 0.3989422804014327)).to(cd)
        else:
            dH = ops.grouped_gemm(dY, w2t, None, offsets, ops.EPI_GELU_GRAD, cd, variant=mod.gemm_variant, residual=Hp)
        offp = ops.pad_offsets(offsets)
        Lp = ops.padded_len(n, E)
        dW2 = ops.grouped_wgrad(ops.transpose_pad(dY, offsets, offp, Lp), ops.transpose_pad(A, offsets, offp, Lp), offp)
        dW1 = ops.grouped_wgrad(ops.transpose_pad(dH, offsets, offp, Lp), ops.transpose_pad(S, offsets, offp, Lp), offp)
        db2 = ops.group_colsum(dY, offsets) if ctx.has_b2 else None
        db1 = ops.group_colsum(dH, offsets) if ctx.has_b1 else None
        dx = None
        if ctx.needs_input_grad[0]:
            w1t = ex.htoh4.weight_as(cd).transpose(1, 2).contiguous()                    # [E, d, h]  (N = d, K = h)
            dS = ops.grouped_gemm(dH, w1t, None, offsets, ops.EPI_NONE, torch.float32, variant=mod.gemm_variant)
            ones = torch.ones(T * k, dtype=torch.float32, device=dout.device)
            dx = ops.gather_combine(dS, inv_pos, ones, T, k, dout.dtype)
        return dx, dscore, dW1, db1, dW2, db2, None, None, None, None, None


def _default_cd():
    from .fmoe import default_compute_dtype
    return default_compute_dtype()


def moe_forward_train(mod, inp: torch.Tensor) -> torch.Tensor:
    """FMoETransformerMLP.forward with autograd (single rank)."""
    from .fmoe import SwitchGate

    if mod.world_size > 1 or getattr(mod, "force_ep", False):
        raise NotImplementedError("expert-parallel training is not built yet; use world_size=1 for backward")
    if mod._generic_act is not None or not mod._fused_gelu:
        raise NotImplementedError("training path supports the reference's GELU(+Dropout) activation only")
    shape = inp.shape
    d, k = mod.d_model, mod.top_k
    x = inp.reshape(-1, d)
    if not x.is_contiguous():
        x = x.contiguous()
    T = x.shape[0]
    g = mod.gate
    is_switch = isinstance(g, SwitchGate)
    noise = g.make_noise(T, x.device) if is_switch else None
    gw = g.gate.weight.detach().float().contiguous()
    gb = g.gate.bias.detach().float() if g.gate.bias is not None else None
    with torch.no_grad():
        idx, score_c, _, _ = ops.router_topk(x.detach(), gw, gb, k, g.kind, noise)
        cap = g.capacity(T)
        counts, offsets, pos, inv_pos, pruned = ops.dispatch_plan(idx, g.tot_expert, cap)
    mod.last_plan = (idx, score_c, counts, offsets, pos, inv_pos)
    # differentiable gate score (tiny [T,E] work) -- the routing itself stays the HIP router's
    if is_switch or k > 1:
        logits = F.linear(x.float(), g.gate.weight.float(), g.gate.bias.float() if g.gate.bias is not None else None)
        if is_switch:
            if noise is not None:
                logits = logits + noise
            probs = torch.softmax(logits, dim=-1)
            score = probs.gather(1, idx)
            g.set_loss(switch_aux_loss(pruned if pruned is not None else idx, probs, g.tot_expert))
        else:
            score = torch.softmax(logits.gather(1, idx), dim=-1)
    else:
        score = score_c  # top-1 naive gate: softmax over one logit == 1, no gradient (SURVEY.md 'DDP + top-1')
    drop_mask = None
    if mod._drop_p > 0 and mod.training:
        cd = mod.compute_dtype or _default_cd()
        keep = 1.0 - mod._drop_p
        drop_mask = (torch.rand(pos.numel(), mod.d_hidden, device=x.device) < keep).to(cd) / keep
    ex = mod.experts
    out = _ExpertFFN.apply(x, score, ex.htoh4.weight, ex.htoh4.bias, ex.h4toh.weight, ex.h4toh.bias, mod, offsets, pos,
                           inv_pos, drop_mask)
    return out.reshape(shape)
