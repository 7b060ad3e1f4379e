"""Training side of the MoE operator (BASELINE cfg 5: capacity + token dropping + aux loss, fwd + bwd).

The differentiable forward is a chain of four autograd Functions, each the adjoint pair SURVEY.md Appendix B
('backward') names; single-rank and expert-parallel training share them:

    _Scatter   S = x[pos // k]                      <->  dx = sum_j dS[inv_pos[t k + j]]        (smoe_scatter_rows / smoe_gather_combine)
    _AllToAll  rows exchanged by (send, recv) splits <->  the same exchange with the splits swapped (RCCL; expert parallel only)
    _GroupFFN  Y = gelu(R W1^T + b1) W2^T + b2       <->  dH = (dY W2) gelu'(H)   smoe_grouped_gemm(W^T shadows, SMOE_EPI_GELU_GRAD)
               per row group (group -> expert map)        dW2 = dY_e^T A_e, dW1 = dH_e^T R_e   smoe_transpose_pad + smoe_grouped_wgrad
                                                          db = column sums                      smoe_group_colsum
                                                          dR = dH W1
    _Combine   out[t] = sum_j score[t,j] Y[inv_pos]  <->  dY = score * dout[pos // k] (smoe_scatter_rows(scale)),
                                                          dscore = <dout[t], Y[inv_pos]>        (smoe_rowdot)

The router's own gradient (through the gate score and the aux loss) is a skinny [T, E] computation; it runs as
ordinary differentiable torch ops on logits recomputed from x, with the ROUTING (idx) taken from the HIP router.
"""
from __future__ import annotations

from typing import List

import torch

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


def _default_cd():
    from .fmoe import default_compute_dtype
    return default_compute_dtype()


class _Scatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, inv_pos, k, cd):
        ctx.k, ctx.T, ctx.dtype = k, x.shape[0], x.dtype
        ctx.save_for_backward(inv_pos)
        return ops.scatter_rows(x, pos, k, cd, zero_fill=True)

    @staticmethod
    def backward(ctx, dS):
        (inv_pos,) = ctx.saved_tensors
        dx = ops.gather_combine(dS.contiguous(), inv_pos, ops.ones_f32(ctx.T * ctx.k, dS.device), ctx.T, ctx.k, ctx.dtype)
        return dx, None, None, None, None


class _GateScatter(torch.autograd.Function):
    """The two readers of the routed rows x as ONE autograd node: ``(logits, S)`` = (the gate's logits the HIP router already computed,
    the scattered rows).  x feeds the gate linear and the scatter; as two nodes autograd adds their input gradients with a kernel
    of its own (a [T, d] read-modify-write per layer).  Here the scatter's adjoint (smoe_gather_combine) takes the gate's
    dx = dl Wg (smoe_gate_dgrad) as its residual operand -- the add rides in the gather's store."""

    @staticmethod
    def forward(ctx, x, w, b, logits, pos, inv_pos, k, cd):
        ctx.k, ctx.has_b = k, b is not None
        ctx.save_for_backward(x, w, inv_pos)
        ctx.set_materialize_grads(False)
        return logits.view_as(logits), ops.scatter_rows(x, pos, k, cd, zero_fill=True)

    @staticmethod
    def backward(ctx, dl, dS):
        x, w, inv_pos = ctx.saved_tensors
        T, k = x.shape[0], ctx.k
        dgate = dw = db = None
        if dl is not None:
            dl = dl.float().contiguous()
            if ctx.needs_input_grad[0]:
                dgate = ops.gate_dgrad(dl, w.detach().float().contiguous(), x.dtype)
            if ctx.needs_input_grad[1]:
                want_b = ctx.has_b and ctx.needs_input_grad[2]
                if dl.shape[1] <= 16:
                    got = ops.gate_wgrad(dl, x.contiguous(), want_bias=want_b)
                    dw, db = got if want_b else (got, None)
                else:
                    dw = dl.t() @ x.float()
                    db = dl.sum(0) if want_b else None
                dw = dw.to(w.dtype)
            elif ctx.has_b and ctx.needs_input_grad[2]:
                db = dl.sum(0)
        dx = None
        if ctx.needs_input_grad[0]:
            if dS is not None:
                dx = ops.gather_combine(dS.contiguous(), inv_pos, ops.ones_f32(T * k, x.device), T, k, x.dtype, residual=dgate)
            else:
                dx = dgate
        return dx, dw, db, None, None, None, None, None


class _Combine(torch.autograd.Function):
    """out[t] = sum_j score[t, j] y[inv_pos[t k + j]] (+ residual[t]: the block's `x + mlp(...)` add fused into the store; its
    gradient is the output's)."""

    @staticmethod
    def forward(ctx, y, score, pos, inv_pos, k, out_dtype, residual=None):
        T = score.shape[0]
        ctx.k, ctx.ydtype, ctx.has_res = k, y.dtype, residual is not None
        ctx.save_for_backward(y, score, pos, inv_pos)
        return ops.gather_combine(y, inv_pos, score.detach().float().contiguous(), T, k, out_dtype,
                                  residual=residual.detach() if residual is not None else None)

    @staticmethod
    def backward(ctx, dout):
        y, score, pos, inv_pos = ctx.saved_tensors
        dout = dout.contiguous()
        T, k = score.shape[0], ctx.k
        dy = ops.scatter_rows(dout, pos, k, ctx.ydtype, zero_fill=True, scale=score.float().contiguous().reshape(-1))
        dscore = ops.rowdot(dout, y, inv_pos, k).view(T, k) if ctx.needs_input_grad[1] else None
        return dy, dscore, None, None, None, None, (dout if ctx.has_res else None)


class _AllToAll(torch.autograd.Function):
    """all-to-all-v of whole rows; backward is the same exchange with the split lists swapped."""

    @staticmethod
    def forward(ctx, rows, send_rows: List[int], recv_rows: List[int], group):
        from .ep import all_to_all_rows, inline_possible
        ctx.send_rows, ctx.recv_rows, ctx.group, ctx.n_in = send_rows, recv_rows, group, rows.shape[0]
        # (nothing runs beside an exchange of the training path: on the compute stream itself unless SLIMMOE_EP_INLINE=0)
        out, _ = all_to_all_rows(rows.contiguous(), send_rows, recv_rows, group, inline=inline_possible(None, training=True))
        return out

    @staticmethod
    def backward(ctx, dout):
        from .ep import all_to_all_rows, inline_possible
        back, _ = all_to_all_rows(dout.contiguous(), ctx.recv_rows, ctx.send_rows, ctx.group, inline=inline_possible(None, training=True))
        if back.shape[0] < ctx.n_in:  # rows past the kept slots were never sent
            pad = torch.zeros((ctx.n_in - back.shape[0], back.shape[1]), dtype=back.dtype, device=back.device)
            back = torch.cat([back, pad], 0)
        return back, None, None, None


class _GroupFFN(torch.autograd.Function):
    """rows [n, d] in G groups (group g uses expert group_expert[g], or g) -> Y [n, d]."""

    @staticmethod
    def forward(ctx, rows, w1, b1, w2, b2, mod, offsets, group_expert, drop_mask, zero_groups=0, group_end=None, rows_hint=None):
        """``group_end`` (i32 [G]; ``offsets`` then holds the G starts): separate row ranges [offsets[g], group_end[g]) -- the slots of
        the static expert exchange.  Rows outside the ranges (padding, header rows) are neither read nor written by any kernel of
        the forward or the backward: Y / drows are garbage there, and nobody downstream reads them."""
        cd = rows.dtype
        ex = mod.experts
        w1c, w2c = ex.htoh4.weight_as(cd), ex.h4toh.weight_as(cd)
        Hp, A = ops.grouped_gemm_gelu_keep(rows, w1c, b1.detach().float() if b1 is not None else None, offsets,
                                           group_expert=group_expert, variant=mod.gemm_variant, group_end=group_end)
        if drop_mask is not None:
            A = A * drop_mask
        Y = ops.grouped_gemm(A, w2c, b2.detach().float() if b2 is not None else None, offsets, ops.EPI_NONE, cd,
                             variant=mod.gemm_variant, group_expert=group_expert, group_end=group_end, rows_hint=rows_hint)
        ctx.mod = mod
        ctx.zero_groups = zero_groups
        ctx.rows_hint = rows_hint
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        ctx.has_map, ctx.has_drop, ctx.has_end = group_expert is not None, drop_mask is not None, group_end is not None
        if group_end is not None and zero_groups:
            raise RuntimeError("_GroupFFN: zero-row groups and separate row ranges do not combine")
        ctx.save_for_backward(rows, Hp, A, offsets,
                              group_expert if group_expert is not None else torch.empty(0, device=rows.device),
                              drop_mask if drop_mask is not None else torch.empty(0, device=rows.device),
                              group_end if group_end is not None else torch.empty(0, device=rows.device))
        return Y

    @staticmethod
    def backward(ctx, dY):
        rows, Hp, A, offsets, gmap, drop_mask, gend = ctx.saved_tensors
        mod = ctx.mod
        cd = rows.dtype
        gexp = gmap if ctx.has_map else None
        gend = gend if ctx.has_end else None
        hint = ctx.rows_hint
        ex = mod.experts
        E_local = ex.htoh4.weight.shape[0]
        G = offsets.numel() - 1 if gend is None else gend.numel()
        n = rows.shape[0]
        dY = dY.contiguous()
        # column sums (bias gradients) right behind the kernel that produced their operand, while it still sits in the 256-MB
        # Infinity Cache: dY's here (the combine's adjoint just wrote it), dH's directly after the GEMM below -- not behind the two
        # weight-gradient GEMMs, which stream 600 MB through the cache first
        need_cs2 = ctx.has_b2 or bool(ctx.zero_groups)
        need_cs1 = ctx.has_b1
        cs2 = ops.group_colsum(dY, offsets, gend) if need_cs2 else None                  # [G, d]
        w2t = ex.h4toh.weight_t_as(cd)                                                   # [E, h, d]  (N = h, K = d)
        # dH = (dY W2) * gelu'(H) [* dropout mask]: gelu' rides in the dgrad GEMM's epilogue; the (elementwise, commuting)
        # dropout mask of the rare drop > 0 training configuration is one multiply behind it
        dH = ops.grouped_gemm(dY, w2t, None, offsets, ops.EPI_GELU_GRAD, cd, variant=mod.gemm_variant,
                              group_expert=gexp, residual=Hp, group_end=gend, rows_hint=hint)
        if ctx.has_drop:
            dH = dH * drop_mask
        cs1 = ops.group_colsum(dH, offsets, gend) if need_cs1 else None                  # [G, h]
        drows = None
        if ctx.needs_input_grad[0]:                                                      # (dH's other reader, same reason)
            w1t = ex.htoh4.weight_t_as(cd)                                               # [E, d, h]  (N = d, K = h)
            drows = ops.grouped_gemm(dH, w1t, None, offsets, ops.EPI_NONE, cd, variant=mod.gemm_variant,
                                     group_expert=gexp, group_end=gend, rows_hint=hint)
        # dW2[e] = dY_e^T A_e, dW1[e] = dH_e^T R_e straight from the token-major tensors (transposing LDS reads;
        # smoe_transpose_pad + smoe_grouped_wgrad is the older two-step form, kept in ops for A/B tests)
        Z = ctx.zero_groups
        if Z:
            # The last Z row groups hold ALL-ZERO input rows (tokens the skip gate masked: models/resMoE.py:141 `x * mask`), every one
            # routed by the gate bias alone to the same expert gmap[E + j] -- up to half of the batch in ONE group, and a weight-gradient
            # tile walks its whole group (443 us against 129 us per launch at ViT-B with 41 % skipped).  Their weight gradients need no
            # GEMM: dW1 gets nothing (zero input rows), and the rows of A = gelu(b1[e]) are identical, so dW2 gets the rank-1 term
            # colsum(dY_g) (x) A_row; the bias gradients are column sums as for every group.  The GEMMs run over the first E groups only.
            E = G - Z
            offs_e = offsets[: E + 1]
            S = ops.expert_wgrad_splits(E, dH.shape[1], rows.shape[1], dH.shape[0], dH.device)
            offs_s = ops.split_offsets(offs_e.contiguous(), S) if S > 1 else None
            dW1 = ops.grouped_wgrad_rows_split(dH, rows, offs_e, S, offs_s)
            dW2 = ops.grouped_wgrad_rows_split(dY, A, offs_e, S, offs_s)
            # rank-1 terms into dW2 (in place) and the experts' bias gradients, one launch (smoe_zero_group_fold)
            db2, db1 = ops.zero_group_fold(cs2, cs1, A, offsets, gmap, E, dW2, want_b2=ctx.has_b2, want_b1=ctx.has_b1)
            G = E
        elif gend is not None:
            # separate row ranges (static expert exchange): one product per (source rank, local expert) slot, nothing read behind its end
            # (cut into pieces at small widths exactly as the compact layout is: ops.expert_wgrad_splits / split_ranges)
            S = ops.expert_wgrad_splits(G, dH.shape[1], rows.shape[1], ctx.rows_hint or dH.shape[0], dH.device)
            pieces = ops.split_ranges(offsets, gend, S) if S > 1 else None
            dW1 = ops.grouped_wgrad_rows_split(dH, rows, offsets, S, pieces, group_end=gend)
            dW2 = ops.grouped_wgrad_rows_split(dY, A, offsets, S, pieces, group_end=gend)
            db2, db1 = (cs2 if ctx.has_b2 else None), cs1
        else:
            # (few, long groups at small widths -- DeiT-Tiny: 24 tiles -- are cut into pieces: ops.expert_wgrad_splits)
            S = ops.expert_wgrad_splits(G, dH.shape[1], rows.shape[1], dH.shape[0], dH.device)
            offs_s = ops.split_offsets(offsets, S) if S > 1 else None
            dW1 = ops.grouped_wgrad_rows_split(dH, rows, offsets, S, offs_s)
            dW2 = ops.grouped_wgrad_rows_split(dY, A, offsets, S, offs_s)
            db2, db1 = (cs2 if ctx.has_b2 else None), cs1
        if G != E_local:  # rank-major groups (source rank, local expert): fold the source ranks
            dW2, dW1 = dW2.view(-1, E_local, *dW2.shape[1:]).sum(0), dW1.view(-1, E_local, *dW1.shape[1:]).sum(0)
            db2 = db2.view(-1, E_local, db2.shape[1]).sum(0) if db2 is not None else None
            db1 = db1.view(-1, E_local, db1.shape[1]).sum(0) if db1 is not None else None
        return drows, dW1, db1, dW2, db2, None, None, None, None, None, None, None


class _GateLogits(torch.autograd.Function):
    """logits = x Wg^T + bg.  Forward value: the logits the HIP router already computed (f32 accumulate, f64 for the
    tokens it had to re-evaluate) -- no second projection; backward: dx = dl Wg (smoe_gate_dgrad: K = E is too thin for a GEMM), dWg = dl^T x as a
    streaming reduction (smoe_gate_wgrad; the library's skinny-output GEMM needs 4x the HBM time), dbg = column sums."""

    @staticmethod
    def forward(ctx, x, w, b, logits):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return logits.view_as(logits)

    @staticmethod
    def backward(ctx, dl):
        x, w = ctx.saved_tensors
        dl = dl.float().contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            if x.is_cuda and x.shape[1] % 4 == 0 and x.dtype in ops._DT and w.dtype == torch.float32 and w.is_contiguous():
                dx = ops.gate_dgrad(dl, w.detach(), x.dtype)       # [T, E] x [E, d]: a streaming kernel, not a GEMM
            else:
                dx = (dl @ w.float()).to(x.dtype)
        dw = db = None
        want_b = ctx.has_b and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if x.is_cuda and dl.shape[1] <= 16 and x.shape[1] % 4 == 0 and x.dtype in (torch.float32, torch.float16, torch.bfloat16):
                got = ops.gate_wgrad(dl, x.contiguous(), want_bias=want_b)     # the bias gradient = the same pass' "ones column"
                dw, db = got if want_b else (got, None)
                dw = dw.to(w.dtype)
            else:
                dw = (dl.t() @ x.float()).to(w.dtype)
        if want_b and db is None:
            db = dl.sum(0)
        return dx, dw, db, None


class _SwitchScoreAux(torch.autograd.Function):
    """SwitchGate's differentiable outputs from what the HIP router already computed: score = softmax(logits + noise)[idx] and
    the load-balance loss aux = E sum_e frac_e prob_e (fmoe.gates.SwitchGate).  Forward: the router's probabilities
    and score, an [E]-sized reduction for aux.  Backward: ONE pass over [T, E] (smoe_switch_gate_bwd) instead of the ~25 small
    autograd kernels of softmax / gather / bincount / where."""

    @staticmethod
    def forward(ctx, logits, probs, score, idx, counts, E):
        # aux and cbase[e] = E frac_e / kept = d aux / d p[t, e] (the same for every t; dropped tokens count nowhere): smoe_switch_aux
        aux, cbase = ops.switch_aux(probs, counts)
        ctx.save_for_backward(probs, idx, cbase)
        ctx.set_materialize_grads(False)     # no aux loss in the objective: daux is None, not a zero tensor
        return score.view_as(score), aux

    @staticmethod
    def backward(ctx, dscore, daux):
        probs, idx, cbase = ctx.saved_tensors
        if dscore is None and daux is None:
            return (None,) * 6
        ds = dscore.reshape(-1).to(torch.float32).contiguous() if dscore is not None else None
        scale = daux.reshape(1).to(torch.float32) if daux is not None else None      # multiplied onto cbase inside the kernel
        return (ops.switch_gate_bwd(probs, idx.reshape(-1), ds, cbase if daux is not None else None, scale),
                None, None, None, None, None)


def _zero_row_routing(mod):
    """(int64 [k] (device): the experts an all-zero row is routed to (top-k of the gate bias; ties -> lowest id), from the HIP
    router itself; the group -> expert map i32 [E + k]; the zero groups' ids int64 [1, k]), cached per version of the gate parameters."""
    from ._cache import StreamCache, param_version
    g = mod.gate.gate
    cache = mod.__dict__.get("_zero_route")
    if cache is None:
        cache = mod.__dict__["_zero_route"] = StreamCache()
        mod.register_load_state_dict_post_hook(lambda m, _k: m.__dict__["_zero_route"].invalidate())
    ver = (param_version(g.weight), param_version(g.bias) if g.bias is not None else None)

    def make():
        z = torch.zeros((1, mod.d_model), dtype=torch.float32, device=g.weight.device)
        idx0, _, _, _ = ops.router_topk(z, g.weight.detach().float().contiguous(), g.bias.detach().float() if g.bias is not None else None,
                                        mod.top_k, ops.GATE_NAIVE)
        return idx0.reshape(-1).clone()
    dev = g.weight.device
    E, k = mod.gate.tot_expert, mod.top_k
    idx0 = cache.get(("idx0", str(dev)), ver, make)
    g_map = cache.get(("gmap", str(dev)), ver, lambda: torch.cat((torch.arange(E, device=dev), idx0)).to(torch.int32))
    zero_ids = cache.get(("zid", str(dev), E, k), 0, lambda: torch.arange(E, E + k, device=dev).reshape(1, k))
    return idx0, g_map, zero_ids


def _route_train(mod, x, zero_rows=None, scatter_cd=None, slots=None):
    """HIP routing + the differentiable gate score; returns (score, counts, offsets, pos, inv_pos, group map | None, zero groups, S).
    ``scatter_cd``: also scatter the rows (compute dtype) -- S, else None; when the gate's logits carry a gradient the scatter and
    the gate linear are one autograd node (_GateScatter: one gradient for x, no add kernel).
    ``zero_rows`` (bool [T], NaiveGate only): tokens whose row is all zero (masked by the token-skip gate).  They get row groups of
    their own -- group E + j for their j-th choice -- that use the expert the gate bias sends every zero row to (the group ->
    expert map): same numbers as dispatching them with everybody else, but the experts' own groups stay balanced and the zero
    groups' weight gradients are rank-1 (_GroupFFN.backward).
    ``slots`` (dict with "tab": ep._SlotTable, "agreed": rows the table was agreed for): the plan in the slot layout of the static
    expert exchange (smoe_dispatch_plan_slots: pos has tab.rows entries, S is the send buffer); "counts" / "raw" come back in it."""
    from .fmoe import SwitchGate

    g = mod.gate
    k = mod.top_k
    T = x.shape[0]
    is_switch = isinstance(g, SwitchGate)
    noise = g.make_noise(T, x.device) if is_switch else None
    gw = g.gate.weight.detach().float().contiguous()
    gb = g.gate.bias.detach().float() if g.gate.bias is not None else None
    need_grad = is_switch or k > 1
    g_map, zero_groups = None, 0
    with torch.no_grad():
        idx, score_c, logits_r, probs_r = ops.router_topk(x.detach(), gw, gb, k, g.kind, noise, want_logits=need_grad,
                                                          want_probs=is_switch)
        cap = g.capacity(T)
        E = g.tot_expert
        if slots is not None:
            tab, agreed = slots["tab"], slots["agreed"]
            if T <= agreed:
                counts, offsets, _gend, pos, inv_pos, pruned, raw = ops.dispatch_plan_slots(idx, E, tab.base_dev, tab.rows, cap)
            else:   # over the agreed size: the buffers keep their shape (the peers must not hang); reported by the overflow watch
                counts, offsets, _gend, pos, inv_h, _pr, raw = ops.dispatch_plan_slots(idx[:agreed].contiguous(), E, tab.base_dev,
                                                                                      tab.rows, cap)
                inv_pos = torch.full((T * k,), -1, dtype=torch.int64, device=x.device)
                inv_pos[: agreed * k].copy_(inv_h)
            slots["counts"], slots["raw"] = counts, raw
        elif zero_rows is not None and not is_switch and cap < 0 and E + k <= 63:
            _idx0, g_map, zero_ids = _zero_row_routing(mod)                                # (cached per gate-parameter version)
            idx_plan = torch.where(zero_rows.reshape(T, 1), zero_ids, idx)
            counts, offsets, pos, inv_pos, pruned = ops.dispatch_plan(idx_plan, E + k, cap)
            zero_groups = k
        else:
            counts, offsets, pos, inv_pos, pruned = ops.dispatch_plan(idx, E, cap)
    mod.last_plan = (idx, score_c, counts, offsets, pos, inv_pos)
    S = None
    if need_grad:  # tiny [T,E] work -- the routing itself stays the HIP router's
        fused = (scatter_cd is not None and x.is_cuda and x.shape[1] % 4 == 0 and x.dtype in ops._DT
                 and g.gate.weight.dtype == torch.float32)
        if fused:
            logits, S = _GateScatter.apply(x, g.gate.weight, g.gate.bias, logits_r, pos, inv_pos, k, scatter_cd)
        else:
            logits = _GateLogits.apply(x, g.gate.weight, g.gate.bias, logits_r)
        if is_switch:   # probs_r = softmax(logits + noise) and score_c = probs_r[idx] are the router's own
            score, aux = _SwitchScoreAux.apply(logits, probs_r, score_c, idx, counts, g.tot_expert)
            g.set_loss(aux)
        else:
            score = torch.softmax(logits.gather(1, idx), dim=-1)
    else:
        score = score_c  # top-1 naive gate: softmax over one logit == 1, no gradient (SURVEY.md 'DDP + top-1')
    if S is None and scatter_cd is not None:
        S = _Scatter.apply(x, pos, inv_pos, k, scatter_cd)
    return score, counts, offsets, pos, inv_pos, g_map, zero_groups, S


def moe_forward_train(mod, inp: torch.Tensor, residual: torch.Tensor = None, row_scale: torch.Tensor = None,
                      zero_rows: torch.Tensor = None) -> torch.Tensor:
    """FMoETransformerMLP.forward with autograd: single rank, or expert parallel (one exchange each way); ``residual``
    (inp's shape and dtype) is added in the combine's store; ``row_scale`` (f32 [T], no gradient) multiplies every token's
    combined expert output (stochastic depth) -- folded into the combine weights, so the backward sees it as part of them."""
    if mod._generic_act is not None or not mod._fused_gelu:
        raise NotImplementedError("training path supports the reference's GELU(+Dropout) activation only")
    cd = mod.compute_dtype or _default_cd()
    if cd == torch.float32:
        raise NotImplementedError("training needs a 16-bit compute dtype (the wgrad GEMM takes f16 / bf16 operands)")
    shape = inp.shape
    d, k = mod.d_model, mod.top_k
    x = inp.reshape(-1, d)
    if not x.is_contiguous():
        x = x.contiguous()
    T = x.shape[0]
    ep = mod.world_size > 1 or getattr(mod, "force_ep", False)
    if mod._drop_p > 0 and mod.training:
        zero_rows = None      # (the rank-1 form of the zero groups' gradients needs identical activation rows: no dropout behind GELU)
    slots = None
    if ep:
        # A capacity gate's exchange has a static shape ("cfg 5 (capacity-bounded) can use fixed-size padded buffers ->
        # no host sync"): the slot layout of the no-grad forward (ep._ep_forward_static), here with autograd nodes around it.  Decided
        # from the configuration and the AGREED row count only, so every rank takes the same branch.  A gate without a capacity takes
        # it speculatively when the harness that owns the step asked for it (rows that do not fit a slot are dropped here and the
        # forward is void: engine.train_one_epoch reads the report before the backward and repeats the forward).
        from . import ep as _ep
        _ep.check_static_overflow()    # (deferred, deterministic: every rank reads the same stats matrices at the same call)
        kind = _ep.static_kind(mod, cd)      # "capacity", or "speculative" where the training harness opted in (set_speculative)
        if kind is not None:
            agreed = _ep.static_slot_tokens(mod, T, x.device)
            if _ep.static_plan_fits(mod, agreed):
                if T == 0:
                    raise RuntimeError("expert-parallel training: this rank has no rows -- its backward would never run and its peers' "
                                       "gradient exchange would wait for it forever")
                st = _ep._slot_state(mod, kind, agreed, x.device)
                slots = {"tab": st.table, "agreed": agreed, "state": st}
    score, counts, offsets, pos, inv_pos, zmap, zero_groups, S = _route_train(mod, x, None if ep else zero_rows, scatter_cd=cd,
                                                                              slots=slots)
    ex = mod.experts
    if slots is not None:
        tab = slots["tab"]
        W, E_local = mod.world_size, mod.num_expert
        # the counts, this rank's row count and its routing histogram ride in the header rows (nobody on the host reads them); S is
        # the send buffer: the scatter's own output, its unused slots zero-filled
        ops.ep_pack_headers(S, slots["counts"], slots["raw"], tab.base_dev, T)
        rows = _AllToAll.apply(S, tab.in_splits, tab.out_splits, mod.moe_group)
        starts, ends, stats = ops.ep_unpack_headers(rows, W, tab.lbase_dev, mod.gate.tot_expert)
        _ep._watch_overflow(stats, slots["state"])
        g_offsets, g_map, g_end = starts, _ep._group_expert_ids(W, E_local, x.device), ends
    elif ep:
        from .ep import exchange_counts, segment_table
        lec, gec = exchange_counts([counts], mod.world_size, mod.moe_group)
        send_rows, recv_rows = lec[0].sum(1).tolist(), gec[0].sum(1).tolist()
        rows = _AllToAll.apply(S, send_rows, recv_rows, mod.moe_group)
        offs, gexp = segment_table(gec[0])
        g_offsets = torch.tensor(offs, dtype=torch.int32, device=x.device)
        g_map = torch.tensor(gexp, dtype=torch.int32, device=x.device)
    else:
        rows, g_offsets, g_map = S, offsets, zmap
    drop_mask = None
    if mod._drop_p > 0 and mod.training:
        keep = 1.0 - mod._drop_p
        drop_mask = (torch.rand(rows.shape[0], mod.d_hidden, device=x.device) < keep).to(cd) / keep
    if slots is None:
        g_end = None
    Y = _GroupFFN.apply(rows, ex.htoh4.weight, ex.htoh4.bias, ex.h4toh.weight, ex.h4toh.bias, mod, g_offsets, g_map,
                        drop_mask, zero_groups, g_end, (T * k if slots is not None else None))
    if slots is not None:
        back = _AllToAll.apply(Y, slots["tab"].out_splits, slots["tab"].in_splits, mod.moe_group)
    elif ep:
        back = _AllToAll.apply(Y, recv_rows, send_rows, mod.moe_group)
        if back.shape[0] < pos.numel():  # slots past the kept count carry no row
            back = torch.cat([back, back.new_zeros(pos.numel() - back.shape[0], d)], 0)
    else:
        back = Y
    res = residual.reshape(-1, d) if residual is not None else None
    if row_scale is not None:
        score = score * row_scale.reshape(T, 1).to(score.dtype)
    out = _Combine.apply(back, score, pos, inv_pos, k, x.dtype, res)
    return out.reshape(shape)
