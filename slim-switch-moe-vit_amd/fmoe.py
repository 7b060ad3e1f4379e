"""MI355X-native stand-in for the third-party ``fmoe`` operator the reference imports.

The reference's whole MoE hot path is ``from fmoe import FMoETransformerMLP`` (models/resMoE.py:6),
constructed at models/resMoE.py:27-29 and called at models/resMoE.py:121,143 and
models/vision_transformer.py:321.  This module re-exports that surface -- same constructor arguments,
same sub-module / parameter names (``gate.gate``, ``experts.htoh4``, ``experts.h4toh``: evidenced by
models/resmoe_flop_hook.py:7 and SURVEY.md section 5 checkpoint row) -- on top of hand-written HIP kernels
(router, dispatch plan, token scatter, grouped MFMA GEMM, gather/combine) reached through the C-ABI
in include/slimmoe.h.  There is no CPU implementation here: tensors must be on the GPU.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._cache import StreamCache, param_version, register_shadow

_COMPUTE_DTYPES = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


# How the single-rank inference path ends a MoE half that is followed by another block (models/vision_transformer.py:319-322):
#   "epilogue": GEMM-2's row-mapped f32 epilogue adds score * y into the residual stream, the next block's norm1 is its own pass;
#   "ln":       GEMM-2 stores contiguous 16-bit rows (direct-store epilogue), then combine + residual + the next norm1 in ONE pass
#               (smoe_gather_combine_ln).  Same HBM bytes per layer; the f32 traffic moves off the CU store path.
# A/B: SLIMMOE_TAIL, or flip the module variable between forwards (tools/tail_ab.py alternates in one process).
TAIL_MODE = os.environ.get("SLIMMOE_TAIL", "epilogue")


def _tail_ln_ok(next_norm, x2: torch.Tensor, cd: torch.dtype, k: int) -> bool:
    d = x2.shape[1]
    return (isinstance(next_norm, nn.LayerNorm) and next_norm.elementwise_affine and next_norm.bias is not None
            and x2.dtype == torch.float32 and cd in (torch.float16, torch.bfloat16) and k <= 4 and d % 8 == 0 and d <= 1024
            and next_norm.weight.dtype == torch.float32)


def default_compute_dtype() -> torch.dtype:
    """dtype of the MFMA operands of the expert GEMMs (accumulation is always f32)."""
    return _COMPUTE_DTYPES[os.environ.get("SLIMMOE_COMPUTE_DTYPE", "f16")]


class FMoELinear(nn.Module):
    """E independent linears held as one [E, out, in] weight and one [E, out] bias (fmoe.linear.FMoELinear)."""

    def __init__(self, num_expert: int, in_feat: int, out_feat: int, bias: bool = True, rank: int = 0):
        super().__init__()
        self.num_expert, self.in_feat, self.out_feat, self.rank = num_expert, in_feat, out_feat, rank
        self.weight = nn.Parameter(torch.empty(num_expert, out_feat, in_feat))
        if bias:
            self.bias = nn.Parameter(torch.zeros(num_expert, out_feat))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()
        self._shadow = StreamCache()  # compute dtype -> 16-bit copy, with the event of the cast that made it
        self.register_load_state_dict_post_hook(lambda mod, _keys: mod._shadow.invalidate())

    def reset_parameters(self):
        # upstream: kaiming_uniform_(a=sqrt(5)) on the 3-D weight, zero bias
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def weight_as(self, dtype: torch.dtype) -> torch.Tensor:
        """Weight in the MFMA operand dtype; a cached shadow copy, refreshed when the parameter changes."""
        w = self.weight
        if w.dtype == dtype:
            return w.detach()
        key = (dtype, w.device)
        if w.dtype == torch.float32 and w.is_contiguous():
            register_shadow(w, self._shadow, key)      # optim.AdamW refreshes this image in its own pass
        return self._shadow.get(key, param_version(w), lambda: ops.cast(w.detach().contiguous(), dtype))

    def weight_t_as(self, dtype: torch.dtype) -> torch.Tensor:
        """``weight.transpose(1, 2)`` as a contiguous [E, in, out] tensor in ``dtype`` -- the operand of the dgrad GEMM
        (it contracts over ``out``); cached like ``weight_as``, made from the master weight in one HIP pass."""
        w = self.weight
        if w.is_cuda and w.shape[1] % 64 == 0 and w.shape[2] % 64 == 0:
            # from the 16-bit image when it is current (the optimizer keeps it so in training: half the bytes to read; the same
            # values -- rounding commutes with the transpose), else from the master weight
            def make():
                src = self._shadow.peek((dtype, w.device), param_version(w)) if w.dtype != dtype else None
                return ops.transpose_cast(src if src is not None else w.detach().contiguous(), dtype)
        else:
            make = lambda: w.detach().transpose(1, 2).to(dtype).contiguous()
        return self._shadow.get(("t", dtype, w.device), param_version(w), make)

    def extra_repr(self):
        return f"num_expert={self.num_expert}, in_features={self.in_feat}, out_features={self.out_feat}"


class NaiveGate(nn.Module):
    """``gate = nn.Linear(d_model, num_expert * world_size)``; top-k of the logits, softmax over the kept k."""

    kind = ops.GATE_NAIVE

    def __init__(self, d_model: int, num_expert: int, world_size: int, top_k: int = 2):
        super().__init__()
        self.gate = nn.Linear(d_model, num_expert * world_size)
        self.top_k = top_k
        self.num_expert, self.world_size = num_expert, world_size
        self.tot_expert = num_expert * world_size
        self.loss = None

    def capacity(self, n_tokens: int) -> int:
        return -1

    def set_loss(self, loss):
        self.loss = loss

    def get_loss(self, clear: bool = True):
        loss = self.loss
        if clear:
            self.loss = None
        return loss


class SwitchGate(NaiveGate):
    """Switch-Transformer top-1 gate with capacity and load-balance loss (fmoe.gates.SwitchGate).

    ``capacity_factor`` follows the Switch definition per source rank: cap = ceil(cf * T_local * k / E_total);
    ``capacity_mode="fmoe"`` selects upstream's ceil(cf * T_local) instead (cf = (train, eval) pair)."""

    kind = ops.GATE_SWITCH

    def __init__(self, d_model: int, num_expert: int, world_size: int, top_k: int = 1, switch_eps: float = 0.1,
                 capacity=(1.2, 2.4), capacity_mode: str = "switch"):
        assert top_k == 1, "SwitchGate is top-1"
        super().__init__(d_model, num_expert, world_size, top_k=1)
        self.switch_eps = switch_eps
        self.capacity_factor = capacity
        self.capacity_mode = capacity_mode

    def capacity(self, n_tokens: int) -> int:
        cf = self.capacity_factor
        if cf is None:
            return -1
        if isinstance(cf, (tuple, list)):
            cf = cf[0] if self.training else cf[1]
        if self.capacity_mode == "fmoe":
            return int(math.ceil(cf * n_tokens))
        return int(math.ceil(cf * n_tokens * self.top_k / self.tot_expert))

    def make_noise(self, T: int, device) -> Optional[torch.Tensor]:
        if not self.training or self.switch_eps <= 0:
            return None
        # upstream adds U[0,1) * 2*eps + (1 - eps) to the logits: one generator kernel (uniform_ applies the affine map itself)
        return torch.empty(T, self.tot_expert, device=device).uniform_(1.0 - self.switch_eps, 1.0 + self.switch_eps)


class _Expert(nn.Module):
    """fmoe.transformer._Expert: htoh4 -> activation -> h4toh over expert-sorted rows.

    State-dict layouts accepted on load (main.py:703-724 resume / 893-907 save):
      * ``experts.htoh4.weight [E,h,d]`` ... -- one module holding all local experts (FastMoE < 1.1; what this class saves);
      * ``experts.{e}.htoh4.weight [1,h,d]`` ... -- FastMoE >= 1.1's ``ModuleList`` of single-expert modules: stacked
        along dim 0 here, in expert order (``fastmoe_v11_state_dict`` writes that layout back)."""

    _PARTS = ("htoh4.weight", "htoh4.bias", "h4toh.weight", "h4toh.bias")

    def __init__(self, num_expert: int, d_model: int, d_hidden: int, activation, rank: int = 0):
        super().__init__()
        self.htoh4 = FMoELinear(num_expert, d_model, d_hidden, bias=True, rank=rank)
        self.h4toh = FMoELinear(num_expert, d_hidden, d_model, bias=True, rank=rank)
        self.activation = activation

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        E = self.htoh4.num_expert
        if prefix + "0.htoh4.weight" in state_dict:
            for part in self._PARTS:
                keys = [f"{prefix}{e}.{part}" for e in range(E)]
                have = [k for k in keys if k in state_dict]
                if not have:
                    continue
                if len(have) != E or f"{prefix}{E}.{part}" in state_dict:
                    error_msgs.append(f"{prefix}: per-expert layout holds {len(have)} (+?) entries of {part}, this module "
                                      f"has {E} local experts")
                    continue
                pieces = [state_dict.pop(k) for k in keys]
                if any(t.shape[0] != 1 for t in pieces):
                    error_msgs.append(f"{prefix}*.{part}: expected a leading dimension of 1 per expert")
                    continue
                state_dict[prefix + part] = torch.cat(pieces, dim=0)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


def fastmoe_v11_state_dict(state_dict: dict) -> dict:
    """A copy of ``state_dict`` with every ``...experts.{htoh4,h4toh}.{weight,bias}`` [E, ...] entry split into FastMoE
    >= 1.1's ``...experts.{e}.{...}`` [1, ...] entries (what a checkpoint written by that version holds)."""
    out = {}
    for k, v in state_dict.items():
        hit = next((p for p in _Expert._PARTS if k.endswith("experts." + p)), None)
        if hit is None:
            out[k] = v
            continue
        stem = k[: -len(hit)]
        for e in range(v.shape[0]):
            out[f"{stem}{e}.{hit}"] = v[e:e + 1].clone()
    return out


def _parse_activation(act):
    """-> (fused_gelu: bool, dropout_p: float, generic_module | None)."""
    mods = list(act) if isinstance(act, nn.Sequential) else [act]
    fused, p, rest = False, 0.0, []
    for i, m in enumerate(mods):
        if i == 0 and isinstance(m, nn.GELU) and getattr(m, "approximate", "none") == "none":
            fused = True
        elif fused and isinstance(m, nn.Dropout) and not rest:
            p = 1 - (1 - p) * (1 - m.p)
        elif isinstance(m, nn.Identity):
            continue
        else:
            rest.append(m)
    if rest:
        return False, 0.0, act
    return fused, p, None if fused else act


class FMoETransformerMLP(nn.Module):
    """Drop-in for ``fmoe.FMoETransformerMLP(num_expert, d_model, d_hidden, activation, top_k=...)``.

    forward(x[..., d]) -> same shape:  router -> dispatch plan -> token scatter -> grouped GEMM (+bias,
    +GELU) -> grouped GEMM (+bias) -> gather/combine, all HIP kernels.
    Keyword-only extensions with reference defaults: ``gate`` ("naive" | "switch" | a gate class),
    ``capacity_factor``, ``world_size`` / ``moe_group`` (expert parallel), ``compute_dtype``.
    """

    def __init__(self, num_expert: int = 32, d_model: int = 1024, d_hidden: int = 4096, activation=None,
                 expert_dp_comm: str = "none", expert_rank: int = 0, *, top_k: int = 2, world_size: int = 1,
                 moe_group=None, gate="naive", capacity_factor=None, capacity_mode: str = "switch",
                 compute_dtype: Optional[torch.dtype] = None, gemm_variant: Optional[int] = None):
        super().__init__()
        if activation is None:
            activation = nn.GELU()
        self.num_expert, self.d_model, self.d_hidden = num_expert, d_model, d_hidden
        self.world_size, self.moe_group, self.top_k = world_size, moe_group, top_k
        self.expert_dp_comm = expert_dp_comm
        if isinstance(gate, str):
            if gate == "naive":
                self.gate = NaiveGate(d_model, num_expert, world_size, top_k)
            elif gate == "switch":
                cap = capacity_factor if capacity_factor is not None else (1.2, 2.4)
                self.gate = SwitchGate(d_model, num_expert, world_size, top_k, capacity=cap, capacity_mode=capacity_mode)
            else:
                raise ValueError(f"unknown gate {gate!r}")
        else:
            self.gate = gate(d_model, num_expert, world_size, top_k)
        self.experts = _Expert(num_expert, d_model, d_hidden, activation, rank=expert_rank)
        self.compute_dtype = compute_dtype
        self.gemm_variant = ops.DEFAULT_GEMM_VARIANT if gemm_variant is None else gemm_variant
        self.ep_chunks = 1  # micro-batches of the expert-parallel pipeline (ep.py); > 1 overlaps a2a with GEMMs
        self._fused_gelu, self._drop_p, self._generic_act = _parse_activation(activation)
        self.last_plan = None  # (idx, score, counts, offsets, pos, inv_pos) of the latest forward, for inspection
        self.mark_parallel_comm()

    # -- data-parallel bookkeeping (fmoe.layers.FMoE.mark_parallel_comm) -----------------------------------------------
    def mark_parallel_comm(self, expert_dp_comm: Optional[str] = None):
        """Tag every parameter with the group its gradient is reduced over, as FastMoE does: the router (replicated on
        all ranks) ``"dp"`` under expert parallelism (``"gate"`` upstream when a separate gate group exists), the
        experts ``expert_dp_comm`` (``"none"``: rank-private, never all-reduced)."""
        comm = expert_dp_comm or self.expert_dp_comm
        for p in self.experts.parameters():
            p.dp_comm = comm
        for p in self.gate.parameters():
            p.dp_comm = "dp" if self.world_size > 1 else "none"

    # -- hot path ------------------------------------------------------------------------------------
    def forward_add(self, inp: torch.Tensor, residual: torch.Tensor, row_scale: Optional[torch.Tensor] = None,
                    zero_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``residual + self(inp)`` with the add fused into the combine store (the ``x + mlp(norm2(x))`` of
        models/vision_transformer.py:321); same arithmetic as the unfused form, one HBM pass fewer.  ``row_scale`` (f32, one
        entry per token, no gradient): ``residual + row_scale[t] * self(inp)[t]`` -- stochastic depth's per-sample ``mask / keep``
        (models/vision_transformer.py:308) folded into the combine weights.  ``zero_rows`` (bool, one entry per token; training only):
        the rows of ``inp`` that are all zero because the token-skip gate masked them -- a hint that changes no result, only how the
        row groups are cut (autograd._route_train)."""
        if torch.is_grad_enabled() and (inp.requires_grad or residual.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            if inp.is_cuda and residual.shape == inp.shape and residual.dtype == inp.dtype and residual.is_contiguous():
                from .autograd import moe_forward_train
                # the add rides in the combine, forward and backward
                return moe_forward_train(self, inp, residual=residual, row_scale=row_scale, zero_rows=zero_rows)
        if row_scale is not None:
            return residual + self.forward(inp) * row_scale.reshape(inp.shape[:-1] + (1,)).to(inp.dtype)
        if torch.is_grad_enabled() and (inp.requires_grad or residual.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            return residual + self.forward(inp)
        if residual.shape != inp.shape or residual.dtype != inp.dtype:
            return residual + self.forward(inp)
        return self._forward_infer(inp, residual=residual)

    def forward_norm_add(self, x: torch.Tensor, norm: nn.Module) -> torch.Tensor:
        """``x + self(norm(x))`` -- the whole MoE half of a ViT block (models/vision_transformer.py:321) -- with the
        block glue fused: LayerNorm + router in one pass over x (smoe_ln_router_topk), the token scatter folded into
        GEMM-1's operand DMA (a_gather) and the combine + residual add folded into GEMM-2's store.  Falls back to
        the unfused composition whenever a precondition does not hold; results agree to rounding."""
        from .ep import drain
        return drain(self.forward_norm_add_steps(x, norm))

    def ep_active(self) -> bool:
        return self.world_size > 1 or bool(getattr(self, "force_ep", False))

    def forward_norm_add_steps(self, x: torch.Tensor, norm: nn.Module, next_norm: Optional[nn.Module] = None):
        """Generator form of ``forward_norm_add``: under expert parallelism it yields at the points where this
        micro-batch waits for the host or for an all-to-all (ep.ep_forward_steps) so that the caller can interleave
        another micro-batch; otherwise it never yields.  The result is the generator's return value.  With ``next_norm``
        (the LayerNorm that reads the result next) the expert-parallel combine also produces ``next_norm(result)`` in
        16 bit and the return value is the pair."""
        cd = self.compute_dtype or default_compute_dtype()
        g = self.gate
        ok = (x.is_cuda and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine
              and tuple(norm.normalized_shape) == (self.d_model,) and self._fused_gelu
              and not (self._drop_p > 0 and self.training) and cd in (torch.float16, torch.bfloat16)
              and self.gemm_variant in (4, 9) and self.d_model % 64 == 0
              and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())
                                                    or any(p.requires_grad for p in norm.parameters())))
              and ops.ln_router_supported(self.d_model, g.tot_expert, g.top_k))
        if not ok:
            return self.forward_add(norm(x), x)
        if self.ep_active():
            from .ep import ep_forward_steps
            x2 = x.reshape(-1, self.d_model)
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            out = yield from ep_forward_steps(self, x2, cd, residual=x2, norm=norm, next_norm=next_norm)
            if isinstance(out, tuple):      # (x + mlp(norm(x)), next_norm(of that)): the combine produced both
                return out[0].reshape(x.shape), out[1].reshape(x.shape)
            return out.reshape(x.shape)
        shape = x.shape
        d, k = self.d_model, self.top_k
        x2 = x.reshape(-1, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T = x2.shape[0]
        is_switch = isinstance(g, SwitchGate)
        noise = g.make_noise(T, x2.device) if is_switch else None
        gw = g.gate.weight.detach().float().contiguous()
        gb = g.gate.bias.detach().float() if g.gate.bias is not None else None
        hist = ops.chunk_hist(T, d, g.tot_expert, k, x2.device)   # the router pass also counts for the plan (count_by_gate folded in)
        xn16, _, idx, score, _, probs = ops.ln_router_topk(
            x2, norm.weight.detach().float(), norm.bias.detach().float() if norm.bias is not None else None, norm.eps,
            gw, gb, k, g.kind, noise, xn16_dtype=cd, want_probs=is_switch, hist=hist)
        cap = g.capacity(T)
        counts, offsets, pos, inv_pos, pruned = ops.dispatch_plan(idx, g.tot_expert, cap, hist=hist)
        self.last_plan = (idx, score, counts, offsets, pos, inv_pos)
        if is_switch:
            from .autograd import switch_aux_loss
            g.set_loss(switch_aux_loss(pruned if pruned is not None else idx, probs, g.tot_expert))
        ex = self.experts
        w1, w2 = ex.htoh4.weight_as(cd), ex.h4toh.weight_as(cd)
        b1 = ex.htoh4.bias.detach().float() if ex.htoh4.bias is not None else None
        b2 = ex.h4toh.bias.detach().float() if ex.h4toh.bias is not None else None
        if next_norm is not None and TAIL_MODE == "ln" and _tail_ln_ok(next_norm, x2, cd, k):
            # GEMM-2 stores plain 16-bit rows straight from its accumulators (the direct-store epilogue: no row map, no f32 residual
            # traffic on the CU's store path, which nothing overlaps with MFMAs), and ONE pass at the HBM roof then does the combine,
            # the residual add and the NEXT block's norm1 (smoe_gather_combine_ln) -- the expert-parallel path's return side, taken
            # on a single rank too.  One more 2^-11 rounding of y than the f32 epilogue; same routing, same float bar.
            h = ops.grouped_gemm(xn16, w1, b1, offsets, ops.EPI_GELU, cd, variant=self.gemm_variant, a_gather=pos, a_div=k)
            y = ops.grouped_gemm(h, w2, b2, offsets, ops.EPI_NONE, cd, variant=self.gemm_variant)
            out, xn_next = ops.gather_combine_ln(y, inv_pos, score, T, k, x2, next_norm.weight.detach().float(),
                                                 next_norm.bias.detach().float(), next_norm.eps, torch.float16)
            return out.reshape(shape), xn_next.reshape(shape)
        if k == 1:
            out = x2.clone() if cap >= 0 else torch.empty_like(x2)
            # both expert GEMMs (scatter folded into GEMM-1's operand fetch, combine + residual into GEMM-2's store) as ONE persistent
            # launch; same arithmetic tile for tile as the two launches below, which remain for the shapes it does not cover
            if not (ops.FFN_FUSED and self.gemm_variant == 9 and x2.dtype == torch.float32 and ops.expert_ffn(
                    xn16, w1, b1, w2, b2, offsets, out, a_gather=pos, a_div=k, row_map=pos, row_scale=score.reshape(-1),
                    residual=x2) is not None):
                h = ops.grouped_gemm(xn16, w1, b1, offsets, ops.EPI_GELU, cd, variant=self.gemm_variant, a_gather=pos, a_div=k)
                ops.grouped_gemm(h, w2, b2, offsets, ops.EPI_NONE, x2.dtype, row_map=pos, row_scale=score.reshape(-1),
                                 out=out, variant=self.gemm_variant, residual=x2)
        else:
            h = ops.grouped_gemm(xn16, w1, b1, offsets, ops.EPI_GELU, cd, variant=self.gemm_variant, a_gather=pos, a_div=k)
            y = ops.grouped_gemm(h, w2, b2, offsets, ops.EPI_NONE, cd, variant=self.gemm_variant)
            out = ops.gather_combine(y, inv_pos, score, T, k, x2.dtype, residual=x2)
        return out.reshape(shape)

    # -- residual-MoE block half with the token-skip gate (models/resMoE.py:137-143) ---------------------------------
    def norm_gate_fusable(self, x: torch.Tensor, norm: nn.Module) -> bool:
        """Whether ``forward_norm_gate_add`` can take ``x``: inference, f32 activations, the reference's GELU expert
        MLP on 16-bit MFMA operands, the naive gate with at most 8 experts, single rank."""
        cd = self.compute_dtype or default_compute_dtype()
        g = self.gate
        return (x.is_cuda and x.dtype == torch.float32 and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine
                and tuple(norm.normalized_shape) == (self.d_model,) and self._fused_gelu and not self.training
                and cd in (torch.float16, torch.bfloat16) and self.gemm_variant in (4, 9) and self.d_model % 64 == 0
                and not torch.is_grad_enabled() and type(g) is NaiveGate
                and ops.gate_ln_router_supported(self.d_model, g.tot_expert, g.top_k))

    def zero_row_output(self) -> torch.Tensor:
        """[d] f32: what this operator returns for an all-zero input row -- the NaiveGate routes it by the gate bias,
        the experts turn it into ``W2[e] gelu(b1[e]) + b2[e]`` -- computed once per parameter version (HIP GEMV)."""
        g, ex = self.gate.gate, self.experts
        ps = (g.bias, ex.h4toh.weight, ex.htoh4.bias, ex.h4toh.bias)
        ver = tuple(param_version(p) if p is not None else None for p in ps)
        cache = self.__dict__.get("_zero_row")
        if cache is None:
            cache = self.__dict__["_zero_row"] = StreamCache()
            self.register_load_state_dict_post_hook(lambda m, _k: m.__dict__["_zero_row"].invalidate())

        def make():
            f = lambda p: None if p is None else p.detach().float().contiguous()
            if self.world_size > 1:
                # expert parallel: the experts a zero row is routed to may live on other ranks -- every rank computes the part
                # its own experts contribute and the parts are summed (one small all-reduce per PARAMETER VERSION, entered by
                # every rank on the same forward: parameters change on all ranks together)
                import torch.distributed as dist
                from .ep import all_reduce_sum
                rank = dist.get_rank(self.moe_group)
                part = ops.zero_row_output(f(g.bias), self.top_k, f(ex.h4toh.weight), f(ex.htoh4.bias), f(ex.h4toh.bias),
                                           e_base=rank * self.num_expert, E_total=self.gate.tot_expert)
                return all_reduce_sum(part, self.moe_group)
            return ops.zero_row_output(f(g.bias), self.top_k, f(ex.h4toh.weight), f(ex.htoh4.bias), f(ex.h4toh.bias))
        return cache.get(str(ex.h4toh.weight.device), ver, make)

    def forward_norm_gate_add(self, x: torch.Tensor, norm: nn.Module, skip_gate) -> torch.Tensor:
        """``xn = norm(x); m = skip_gate(xn); return self(xn * m[..., 1:]) + xn * m[..., 1:] + xn * m[..., :1]`` -- the
        MoE half of models/resMoE.py:126-145 -- as: ONE pass over x for LayerNorm + skip gate + router
        (ops.gate_ln_router), the dispatch plan over the tokens that enter the experts, GEMM-1 gathering its rows from
        the 16-bit image, GEMM-2 (k = 1) or the combine (k > 1) adding into the f32 residual image in place.  Tokens the
        gate masks are all-zero rows for the operator: they are not dispatched; their constant output
        (``zero_row_output``) is added to their residual row by the first pass.  ``skip_gate`` is a resmoe.Gate."""
        from .ep import drain
        return drain(self.forward_norm_gate_add_steps(x, norm, skip_gate))

    def forward_norm_gate_add_steps(self, x: torch.Tensor, norm: nn.Module, skip_gate):
        """Generator form of ``forward_norm_gate_add`` (result = return value): under expert parallelism it yields where this
        batch waits for the host or an all-to-all (ep.ep_forward_steps, which takes the routing of the fused pass as given: the
        tokens the skip gate masked -- ``idx_plan = -1`` -- are simply not sent)."""
        assert self.norm_gate_fusable(x, norm), "forward_norm_gate_add: preconditions (norm_gate_fusable) do not hold"
        cd = self.compute_dtype or default_compute_dtype()
        g, d, k = self.gate, self.d_model, self.top_k
        shape = x.shape
        x2 = x.reshape(-1, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        T = x2.shape[0]
        lin = skip_gate.head[1]
        thr = skip_gate.active_threshold()
        if thr is not None:
            skip_gate._total_tokens += T
        zero_out = self.zero_row_output() if thr is not None else None    # (expert parallel: a collective on its first call)
        if T == 0:    # an expert-parallel rank without rows still takes part in the layer's collectives
            from .ep import ep_forward_steps
            empty = dict(idx=torch.empty((0, k), dtype=torch.int64, device=x2.device), idx_plan=None,
                         score=torch.empty((0, k), dtype=torch.float32, device=x2.device),
                         src=torch.empty((0, d), dtype=cd, device=x2.device))
            if self.ep_active():
                yield from ep_forward_steps(self, x2, cd, residual=x2, routed=empty)
            return x2.reshape(shape)
        r = ops.gate_ln_router(
            x2, lin.weight, lin.bias, thr,
            ln=(norm.weight.detach(), norm.bias.detach() if norm.bias is not None else None, norm.eps),
            wg=g.gate.weight.detach().float().contiguous(), bg=g.gate.bias.detach().float() if g.gate.bias is not None else None,
            k=k, xn16_dtype=cd, want_xn32=True, zero_out=zero_out,
            skip_count=skip_gate.skip_counter(x.device) if thr is not None else None,
            hist=(hist := ops.chunk_hist(T, d, g.tot_expert, k, x2.device)))
        idx, score, out = r["idx"], r["score"], r["xn32"]
        if self.ep_active():
            from .ep import ep_forward_steps
            res = yield from ep_forward_steps(self, out, cd, residual=out,
                                              routed=dict(idx=idx, idx_plan=r["idx_plan"], score=score, src=r["xn16"]))
            return res.reshape(shape)
        counts, offsets, pos, inv_pos, _ = ops.dispatch_plan(r["idx_plan"], g.tot_expert, -1, hist=hist)
        self.last_plan = (idx, score, counts, offsets, pos, inv_pos)
        ex = self.experts
        w1, w2 = ex.htoh4.weight_as(cd), ex.h4toh.weight_as(cd)
        b1 = ex.htoh4.bias.detach().float() if ex.htoh4.bias is not None else None
        b2 = ex.h4toh.bias.detach().float() if ex.h4toh.bias is not None else None
        if k == 1:
            if not (ops.FFN_FUSED and self.gemm_variant == 9 and ops.expert_ffn(
                    r["xn16"], w1, b1, w2, b2, offsets, out, a_gather=pos, a_div=k, row_map=pos, row_scale=score.reshape(-1),
                    residual=out) is not None):
                h = ops.grouped_gemm(r["xn16"], w1, b1, offsets, ops.EPI_GELU, cd, variant=self.gemm_variant, a_gather=pos, a_div=k)
                ops.grouped_gemm(h, w2, b2, offsets, ops.EPI_NONE, torch.float32, row_map=pos, row_scale=score.reshape(-1),
                                 out=out, variant=self.gemm_variant, residual=out)
        else:
            h = ops.grouped_gemm(r["xn16"], w1, b1, offsets, ops.EPI_GELU, cd, variant=self.gemm_variant, a_gather=pos, a_div=k)
            y = ops.grouped_gemm(h, w2, b2, offsets, ops.EPI_NONE, cd, variant=self.gemm_variant)
            ops.gather_combine(y, inv_pos, score, T, k, torch.float32, residual=out, out=out)
        return out.reshape(shape)

    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        if not inp.is_cuda:
            raise RuntimeError("FMoETransformerMLP: input must be on the GPU; this build has no CPU path "
                               "(the CPU restatement lives in oracle/ and is test infrastructure only)")
        if torch.is_grad_enabled() and (inp.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .autograd import moe_forward_train
            return moe_forward_train(self, inp)
        return self._forward_infer(inp)

    def _route(self, x: torch.Tensor):
        g = self.gate
        T = x.shape[0]
        noise = g.make_noise(T, x.device) if isinstance(g, SwitchGate) else None
        want_probs = isinstance(g, SwitchGate)
        gw = g.gate.weight.detach()
        gb = g.gate.bias.detach() if g.gate.bias is not None else None
        if gw.dtype != torch.float32:
            gw, gb = gw.float(), (gb.float() if gb is not None else None)
        idx, score, _, probs = ops.router_topk(x, gw.contiguous(), gb, g.top_k, g.kind, noise, want_probs=want_probs)
        cap = g.capacity(T)
        counts, offsets, pos, inv_pos, pruned = ops.dispatch_plan(idx, g.tot_expert, cap)
        return idx, score, probs, counts, offsets, pos, inv_pos, pruned

    def _experts_fwd(self, rows: torch.Tensor, offsets: torch.Tensor, cd: torch.dtype, out=None, row_map=None,
                     row_scale=None, out_dtype=None, group_expert=None, residual=None, group_end=None, rows_hint=None):
        """``rows_hint``: with ``group_end`` (row ranges inside a padded buffer) the number of rows expected to exist -- what the
        GEMMs' tile height is chosen for, and what the profiler counts FLOPs over."""
        ex = self.experts
        w1, w2 = ex.htoh4.weight_as(cd), ex.h4toh.weight_as(cd)
        b1 = ex.htoh4.bias.detach().float() if ex.htoh4.bias is not None else None
        b2 = ex.h4toh.bias.detach().float() if ex.h4toh.bias is not None else None
        if self._fused_gelu:
            h = ops.grouped_gemm(rows, w1, b1, offsets, ops.EPI_GELU, cd, variant=self.gemm_variant,
                                 group_expert=group_expert, group_end=group_end, rows_hint=rows_hint)
            if self._drop_p > 0 and self.training:
                h = torch.nn.functional.dropout(h, self._drop_p, True)
        else:
            h = ops.grouped_gemm(rows, w1, b1, offsets, ops.EPI_NONE, cd, variant=self.gemm_variant,
                                 group_expert=group_expert, group_end=group_end, rows_hint=rows_hint)
            h = self._generic_act(h).to(cd).contiguous()
        return ops.grouped_gemm(h, w2, b2, offsets, ops.EPI_NONE, out_dtype, row_map=row_map, row_scale=row_scale,
                                out=out, variant=self.gemm_variant, group_expert=group_expert, residual=residual,
                                group_end=group_end, rows_hint=rows_hint)

    def _forward_infer(self, inp: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        shape = inp.shape
        d, k = self.d_model, self.top_k
        x = inp.reshape(-1, d)
        if not x.is_contiguous():
            x = x.contiguous()
        res = None
        if residual is not None:
            res = residual.reshape(-1, d)
            if not res.is_contiguous():
                res = res.contiguous()
        T = x.shape[0]
        cd = self.compute_dtype or default_compute_dtype()
        if self.world_size > 1 or getattr(self, "force_ep", False):
            from .ep import ep_forward
            return ep_forward(self, x, cd, residual=res).reshape(shape)
        idx, score, probs, counts, offsets, pos, inv_pos, pruned = self._route(x)
        self.last_plan = (idx, score, counts, offsets, pos, inv_pos)
        if isinstance(self.gate, SwitchGate):
            from .autograd import switch_aux_loss
            self.gate.set_loss(switch_aux_loss(pruned if pruned is not None else idx, probs, self.gate.tot_expert))
        buf = ops.scatter_rows(x, pos, k, cd)
        if k == 1:
            # fused combine: GEMM-2 stores row s to out[pos[s]] * score[pos[s]]; dropped tokens stay 0
            dropping = self.gate.capacity(T) >= 0
            if dropping:  # rows of dropped tokens are never stored by the GEMM: pre-fill them
                out = res.clone() if res is not None else torch.zeros((T, d), dtype=inp.dtype, device=inp.device)
            else:
                out = torch.empty((T, d), dtype=inp.dtype, device=inp.device)
            self._experts_fwd(buf, offsets, cd, out=out, row_map=pos, row_scale=score.reshape(-1),
                              out_dtype=inp.dtype, residual=res)
        else:
            y = self._experts_fwd(buf, offsets, cd, out_dtype=cd)
            out = ops.gather_combine(y, inv_pos, score, T, k, inp.dtype, residual=res)
        return out.reshape(shape)


def ddp_ignore_expert_parameters(model: nn.Module) -> list:
    """Call BEFORE wrapping ``model`` in ``torch.nn.parallel.DistributedDataParallel`` (the reference wraps the whole
    model, main.py:611).  Under expert parallelism every rank holds DIFFERENT experts behind the same parameter names:
    plain DDP would broadcast rank 0's expert slices over everybody's at construction and then average the gradients of
    unrelated experts.  This registers every parameter tagged ``dp_comm == "none"`` on an expert-parallel MoE module
    (``world_size > 1``) in ``model._ddp_params_and_buffers_to_ignore`` -- DDP then neither broadcasts nor reduces them --
    and returns the names.  (FastMoE ships its own ``DistributedGroupedDataParallel`` for the same purpose.)"""
    names = []
    for mod_name, mod in model.named_modules():
        if isinstance(mod, FMoETransformerMLP) and mod.world_size > 1:
            for p_name, p in mod.named_parameters():
                if getattr(p, "dp_comm", None) == "none":
                    names.append(f"{mod_name}.{p_name}" if mod_name else p_name)
    if names:
        have = list(getattr(model, "_ddp_params_and_buffers_to_ignore", []))
        model._ddp_params_and_buffers_to_ignore = have + [n for n in names if n not in have]
    return names
