"""The reference's MoE-block surface (models/resMoE.py), re-exported on the MI355X-native operator.

Same class names, constructor signatures, sub-module / buffer names and factory names as the
reference, so callers written against models/resMoE.py (main.py:520-530 create_model, main.py:623
``"moe_gate"`` / ``"dense_gate"`` param-name matching, main.py:812 ``isinstance(m, Gate)``) work
unchanged.  Deliberately NOT reproduced: the dead ``ResBlock`` (resMoE.py:88-123, wrong positional
argument order; SURVEY.md note A).
"""
from __future__ import annotations

import typing as typ

import torch as th
import torch.nn as nn

from . import ops
from .fmoe import FMoETransformerMLP
from .vit import Block, VisionTransformer, register_model, deit_tiny_patch16_224, _deit

__all__ = [
    "CustomizedMoEMLP", "Gate", "forward_residule_moe", "patch_blocks_with_moe",
    "resmoe_tiny_patch16_224_expert8", "moe_tiny_patch16_224_expert8",
    "moe_tiny_patch16_224_expert4_top1", "moe_base_patch16_224_expert8_top1",
    "resmoe_base_patch16_224_expert8_top1", "moe_base_patch16_224_expert16_top1",
    "moe_large_patch16_384_expert32_top1",
]


class CustomizedMoEMLP(FMoETransformerMLP):
    """The reference's expert MLP (models/resMoE.py:15-29): ``FMoETransformerMLP`` whose activation is
    ``act_layer()`` followed by ``Dropout(drop)``, default (naive) gate, ``moe_top_k`` experts per token.
    Positional arguments in the reference's order; everything after them is this build's keyword-only extension
    (``gate``, ``capacity_factor``, ``world_size``, ``moe_group``, ``compute_dtype``, ``gemm_variant``)."""

    def __init__(self, in_features: int, hidden_features: int, moe_num_experts: int, moe_top_k: int, drop: float,
                 act_layer: typ.Callable = th.nn.GELU, **moe_kwargs):
        super().__init__(moe_num_experts, in_features, hidden_features, nn.Sequential(act_layer(), nn.Dropout(p=drop)),
                         top_k=moe_top_k, **moe_kwargs)


class Gate(nn.Module):
    """Token-skip gate (models/resMoE.py:32-85).  ``forward(x[B,N,d]) -> mask[B,N,2]``: channel 0 marks the tokens
    that bypass the following operator, channel 1 the tokens that enter it.

        p = sigmoid(head(x));   skipped  <=>  p > threshold      (``_threshold`` while training, ``threshold`` in eval)

    Same constructor signature, sub-module (``head.0`` dropout, ``head.1`` linear), buffers (``_threshold``,
    ``threshold``) and attributes (``disable``, ``is_hard``, ``_total_tokens``, ``_skipped_tokens``, ``step``) as the
    reference, so ``isinstance(m, Gate)`` loops (main.py:812) and the ``"moe_gate"`` / ``"dense_gate"`` parameter-name
    matching (main.py:623) keep working.  What is different underneath:

    * inference runs ONE HIP kernel (ops.gate_ln_router: dot product, threshold decision defined on the f64-accurate
      logit, mask, device-side skip counter) -- or none at all inside ``forward_residule_moe``'s fused path, where the
      gate rides on the LayerNorm / router pass;
    * the hard masks are exactly 0 / 1 in value; training keeps the straight-through estimator through ``p``;
    * ``_skipped_tokens`` accumulates on the device and is read back only when somebody looks at it (the reference
      synchronises the stream twice per block with ``.item()``, resMoE.py:83); ``step`` anneals on the device too."""

    def __init__(self, in_dim: int, tau: float, dropout: float = 0.0, target_threshold: float = 0.9,
                 starting_threshold: float = 1.0, is_hard: float = True):
        super().__init__()
        self.head = nn.Sequential(nn.Dropout(p=dropout), nn.Linear(in_dim, 1))
        self.register_buffer("_threshold", th.tensor(starting_threshold))
        self.register_buffer("threshold", th.tensor(target_threshold))
        self.tau = tau  # accepted and unused, as in the reference
        self.is_hard = is_hard
        self.disable = False
        self._total_tokens = 0
        self._skip_host = 0.0   # skipped tokens already read back
        self._skip_dev = None   # int32[1] on the device: skipped tokens since the last read-back

    # -- counters ------------------------------------------------------------------------------------------------
    def skip_counter(self, device) -> th.Tensor:
        """The device-side counter the HIP kernels add to."""
        if self._skip_dev is None or self._skip_dev.device != device:
            self._flush_counter()
            self._skip_dev = th.zeros(1, dtype=th.int32, device=device)
        return self._skip_dev

    def _flush_counter(self):
        if self._skip_dev is not None:
            self._skip_host += float(self._skip_dev.item())
            self._skip_dev.zero_()

    @property
    def _skipped_tokens(self):
        self._flush_counter()
        return self._skip_host

    @_skipped_tokens.setter
    def _skipped_tokens(self, v):
        if self._skip_dev is not None:
            self._skip_dev.zero_()
        self._skip_host = float(v)

    # -- schedule (main.py:812-815 calls step(delta) once per iteration) -------------------------------------------
    def step(self, delta):
        """``_threshold <- max(_threshold - delta, threshold)`` without leaving the device."""
        th.maximum(self._threshold - delta, self.threshold, out=self._threshold)

    def active_threshold(self) -> typ.Optional[th.Tensor]:
        """The buffer the decision compares against right now; None when the gate is disabled (everything passes)."""
        if self.disable:
            return None
        return self._threshold if self.training else self.threshold

    def _hip_ok(self, x: th.Tensor) -> bool:
        lin = self.head[1]
        return (x.is_cuda and x.dtype == th.float32 and x.dim() == 3 and not (th.is_grad_enabled() and (
            x.requires_grad or lin.weight.requires_grad)) and not (self.training and (self.head[0].p > 0 or not self.is_hard))
            and ops.gate_ln_router_supported(x.shape[-1], 0, 1))

    def forward(self, x):
        B, N = x.shape[0], x.shape[1]
        if self.disable:
            return th.stack((x.new_zeros(B, N), x.new_ones(B, N)), dim=-1)
        self._total_tokens += B * N
        if self._hip_ok(x):
            lin = self.head[1]
            xr = x.reshape(B * N, x.shape[-1])
            if not xr.is_contiguous():
                xr = xr.contiguous()
            r = ops.gate_ln_router(xr, lin.weight, lin.bias, self.active_threshold(), want_mask=True,
                                   skip_count=self.skip_counter(x.device))
            return r["mask"].reshape(B, N, 2)
        # differentiable composition (training; CPU tensors; dtypes the kernel does not take)
        p = th.sigmoid(self.head(x))
        if self.training and not self.is_hard:
            bypass, enter = 1 - p, p
        else:
            hard = (p > self.active_threshold()).to(p.dtype)
            st = p.detach() - p                     # 0 in value, d/dp = -1: the straight-through estimator
            bypass, enter = hard - st, (1 - hard) + st
        skipped = bypass.detach().sum()
        if skipped.is_cuda:
            self.skip_counter(x.device).add_(skipped.round().to(th.int32))
        else:
            self._skip_host += float(skipped)
        return th.cat((bypass, enter), dim=-1)


def _gated(x_normed, gate):
    """(rows that enter the operator, rows that bypass it): the normed activations split by the gate's mask."""
    mask = gate(x_normed)
    return x_normed * mask[..., 1:2], x_normed * mask[..., 0:1]


def _residual_block_composed(blk, x):
    """models/resMoE.py:126-145 as a composition of modules (training, f32 / non-autocast inference, CPU): in both
    halves the residual is the NORMED activation, split by the skip gate into the part the operator sees and the part
    that bypasses it."""
    x = blk.norm1(x)
    enter, bypass = _gated(x, blk.dense_gate)
    x = blk.drop_path(blk.attn(enter)) + enter + bypass
    x = blk.norm2(x)
    enter, bypass = _gated(x, blk.moe_gate)
    return blk.drop_path(blk.mlp(enter)) + enter + bypass


class _GateLNFn(th.autograd.Function):
    """One gated half's front end in TRAINING (models/resMoE.py:126-131 / 137-140 with Gate.forward's hard branch, 68-77):

        xn = norm(x);  m = gate(xn);  tk = xn * m[..., 1:]          ->  (xn f32, tk, mask)

    forward: ONE HIP pass (smoe_gate_ln_router: LayerNorm, gate logit, decision on the f64-accurate logit, the 16-bit masked
    operand image for the attention half / the masked f32 image the MoE router reads); backward: ONE pass too
    (smoe_gate_ln_bwd: straight-through estimator d m1 / d p = -1, d m0 / d p = +1 with p, xn recomputed from x; LayerNorm backward;
    the gate's and the LayerNorm's parameter gradients).  The
    caller computes out = f(tk) + xn (the reference's f(tk) + tk + skip_tk in value; the mask gradients are this Function's)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, gate_w, gate_b, eps, thr, skip_count, want16):
        r = ops.gate_ln_router(x, gate_w, gate_b, thr, ln=(ln_w.detach().float(), ln_b.detach().float() if ln_b is not None else None, eps),
                               xn16_dtype=th.float16 if want16 else None, want_xn32=True, want_mask=True, skip_count=skip_count,
                               want_tk32=not want16)
        xn, mask = r["xn32"], r["mask"]
        # the masked operand image: 16-bit rows for the attention half; the f32 image with the skipped rows zeroed -- what the MoE's
        # router and scatter read -- for the MoE half (a second f32 store of the same pass)
        tk = r["xn16"] if want16 else r["tk32"]
        ctx.eps, ctx.gate_on = eps, thr is not None
        ctx.save_for_backward(x, ln_w, gate_w, gate_b if gate_b is not None else th.empty(0, device=x.device), mask,
                              ln_b if ln_b is not None else th.empty(0, device=x.device))
        ctx.has_gb, ctx.has_lb = gate_b is not None, ln_b is not None
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)     # no zero tensor for the mask's (never used) gradient, nor for an unused branch
        return xn, tk, mask

    @staticmethod
    def backward(ctx, d_xn, d_tk, _d_mask):
        x, ln_w, gate_w, gate_b, mask, ln_b = ctx.saved_tensors
        if d_tk is None and d_xn is None:
            return (None,) * 9
        if d_tk is None:
            d_tk = th.zeros_like(x)
        d_tk = d_tk.contiguous()
        if d_xn is not None and not (d_xn.dtype == th.float32 and d_xn.is_contiguous()):
            d_xn = d_xn.float().contiguous()
        # ONE pass over x: the gate's straight-through gradients, the LayerNorm backward and both modules' parameter gradients
        # (smoe_gate_ln_bwd; as three kernels -- smoe_skip_gate_bwd -> smoe_layernorm_bwd -> smoe_gate_wgrad -- the [T, d] gradient
        # of the normed activations made a round trip through HBM and xn was read twice more)
        dx, dg, dbeta, dgw, dgb, _ = ops.gate_ln_bwd(x, d_tk, d_xn, ln_w.detach().float(), ln_b.detach().float() if ctx.has_lb else None,
                                                     ctx.eps, gate_w, gate_b if ctx.has_gb else None, mask, ctx.gate_on)
        if ctx.gate_on:
            dgw = dgw.reshape(gate_w.shape).to(gate_w.dtype)
            dgb = dgb.reshape(gate_b.shape).to(gate_b.dtype) if ctx.has_gb else None
        else:
            dgw = dgb = None
        return dx, dg.to(ln_w.dtype), (dbeta if ctx.has_lb else None), dgw, dgb, None, None, None, None


def _train_ok(blk, x) -> bool:
    """fp16-autocast TRAINING of the residual-MoE block on the library's kernels: hard gates without dropout (the reference's
    construction, resMoE.py:163-170), LayerNorms / attention / MoE operator the training kernels cover, own stochastic depth."""
    from . import dense
    from .vit import Attention, DropPath
    d = x.shape[-1]
    if not (x.is_cuda and x.dim() == 3 and x.dtype == th.float32 and x.is_contiguous() and dense.autocast_half_training(x)):
        return False
    gates_ok = all(isinstance(g, Gate) and (g.is_hard or not g.training) and not (g.training and g.head[0].p > 0)
                   for g in (blk.dense_gate, blk.moe_gate))
    return (gates_ok and isinstance(blk.attn, Attention) and (blk.stochastic_depth_inactive() or isinstance(blk.drop_path, DropPath))
            and all(dense.layer_norm_supported(x, n) for n in (blk.norm1, blk.norm2))
            and getattr(blk.mlp, "forward_add", None) is not None and ops.gate_ln_router_supported(d, 0, 1))


def _gate_ln(x2, norm, gate):
    lin = gate.head[1]
    thr = gate.active_threshold()
    if thr is not None:
        gate._total_tokens += x2.shape[0]
    return norm.weight, norm.bias, lin.weight, lin.bias, norm.eps, thr, (gate.skip_counter(x2.device) if thr is not None else None)


def _residual_block_train(blk, x):
    """models/resMoE.py:126-145 in fp16-autocast training, forward and backward on the library's kernels: per half ONE pass for
    LayerNorm + gate (+ the masked operand image), the operator's own training path (qkv / attention / projection GEMMs with the
    normed residual -- and stochastic depth's per-row factor -- in the projection's store; the MoE operator with the residual in
    its combine), and in the backward the gate's straight-through gradients fused with the gradient of the residual in one pass in
    front of the LayerNorm backward kernel.  Preconditions: _train_ok."""
    B, N, d = x.shape
    x2 = x.reshape(B * N, d)
    _, rows1 = blk._depth_scale(x)
    xn, tk16, _ = _GateLNFn.apply(x2, *_gate_ln(x2, blk.norm1, blk.dense_gate), True)
    a, added = blk.attn(tk16.reshape(B, N, d), residual=xn.reshape(B, N, d), row_scale=rows1)
    if not added:   # (the projection could not take the residual: add it here; the gate gradients are unaffected)
        a = xn.reshape(B, N, d) + (a if rows1 is None else a * rows1.reshape(B, N, 1).to(a.dtype))
    x1 = a.reshape(B * N, d)
    _, rows2 = blk._depth_scale(x)
    xn2, tk32, mask2 = _GateLNFn.apply(x1, *_gate_ln(x1, blk.norm2, blk.moe_gate), False)
    # (the skipped tokens' all-zero rows get row groups of their own in the operator's training path: autograd._route_train)
    skipped = mask2[:, 0] > 0.5 if blk.moe_gate.active_threshold() is not None else None
    return blk.mlp.forward_add(tk32.reshape(B, N, d), xn2.reshape(B, N, d), row_scale=rows2, zero_rows=skipped)


def _fused_ok(blk, x) -> bool:
    from .vit import _autocast_half_inference, Attention
    d = x.shape[-1]
    return (x.is_cuda and x.dim() == 3 and x.dtype == th.float32 and x.is_contiguous() and not blk.training
            and blk.stochastic_depth_inactive() and _autocast_half_inference(x)
            and isinstance(blk.attn, Attention) and isinstance(blk.dense_gate, Gate) and isinstance(blk.moe_gate, Gate)
            and all(isinstance(n, nn.LayerNorm) and n.elementwise_affine and tuple(n.normalized_shape) == (d,)
                    for n in (blk.norm1, blk.norm2))
            and hasattr(blk.mlp, "norm_gate_fusable") and blk.mlp.norm_gate_fusable(x, blk.norm2)
            and ops.gate_ln_router_supported(d, 0, 1))


def _residual_block_fused(blk, x):
    """The same block for fp16-autocast inference in six launches' worth of glue-free work per half:

      attention half   norm1 + dense_gate in ONE pass over x (ops.gate_ln_router): the fp16 operand image of the
                       tokens that enter attention (zero rows for the bypassing ones) and the f32 residual image;
                       qkv GEMM -> attention kernel -> projection GEMM with `+ residual` in its store
      MoE half         norm2 + moe_gate + router in ONE pass (FMoETransformerMLP.forward_norm_gate_add): bypassing tokens
                       are all-zero rows for the experts, so they are not dispatched at all -- what they would receive
                       (the per-layer constant of ops.zero_row_output) is added to their residual row by the same pass;
                       GEMM-1 gathers its rows from the fp16 image, GEMM-2 adds into the residual image in place.
    Preconditions: _fused_ok."""
    B, N, d = x.shape
    g = blk.dense_gate
    lin = g.head[1]
    n1 = blk.norm1
    if not g.disable:
        g._total_tokens += B * N
    r = ops.gate_ln_router(x.reshape(B * N, d), lin.weight, lin.bias, g.active_threshold(),
                           ln=(n1.weight.detach(), n1.bias.detach() if n1.bias is not None else None, n1.eps),
                           xn16_dtype=th.float16, want_xn32=True,
                           skip_count=None if g.disable else g.skip_counter(x.device))
    res = r["xn32"].reshape(B, N, d)
    a, added = blk.attn(r["xn16"].reshape(B, N, d), residual=res)
    x = a if added else a + res
    return blk.mlp.forward_norm_gate_add(x, blk.norm2, blk.moe_gate)


def forward_residule_moe(self, x):
    """Bound to every Block by the ``resmoe_*`` factories in place of ``Block.forward`` (models/resMoE.py:126-145,
    185-186)."""
    if _fused_ok(self, x):
        return _residual_block_fused(self, x)
    if _train_ok(self, x):
        return _residual_block_train(self, x)
    if x.is_cuda and th.is_autocast_enabled():
        from .vit import _warn_fallback
        _warn_fallback("residual-MoE block", "config: composed from torch modules (needs fp16-autocast inference, f32 contiguous "
                       "activations, inactive stochastic depth, the naive gate)", x.shape)
    return _residual_block_composed(self, x)


def patch_blocks_with_moe(model: VisionTransformer, num_experts: int, top_k: int, residual: bool,
                          starting_threshold: float = 1.0, target_threshold: float = 0.9, drop_rate: float = 0.0,
                          mlp_ratio: int = 4, **moe_kwargs):
    """What the reference factories do to every Block (resMoE.py:163-186 / 200-208)."""
    embed_dim = model.embed_dim
    for _, module in model.named_modules():
        if isinstance(module, Block):
            if residual:
                module.dense_gate = Gate(embed_dim, 1.0, starting_threshold=starting_threshold,
                                         target_threshold=target_threshold)
                module.moe_gate = Gate(embed_dim, 1.0, starting_threshold=starting_threshold,
                                       target_threshold=target_threshold)
            module.mlp = CustomizedMoEMLP(embed_dim, embed_dim * mlp_ratio, moe_num_experts=num_experts,
                                          moe_top_k=top_k, drop=drop_rate, **moe_kwargs)
            if residual:
                bound_method = forward_residule_moe.__get__(module, module.__class__)
                setattr(module, "forward", bound_method)
    return model


def _split_moe_kwargs(kwargs):
    keys = ("gate", "capacity_factor", "capacity_mode", "world_size", "moe_group", "compute_dtype", "gemm_variant")
    return {k: kwargs.pop(k) for k in keys if k in kwargs}


@register_model
def resmoe_tiny_patch16_224_expert8(pretrained=False, starting_threshold=1.0, target_threshold=0.9, **kwargs):
    """models/resMoE.py:151-187 (DeiT-Tiny, E=8, top-2, token-skip gates)."""
    moe_kwargs = _split_moe_kwargs(kwargs)
    model = deit_tiny_patch16_224(pretrained=pretrained, **kwargs)
    return patch_blocks_with_moe(model, 8, 2, True, starting_threshold, target_threshold, **moe_kwargs)


@register_model
def moe_tiny_patch16_224_expert8(pretrained=False, **kwargs):
    """models/resMoE.py:190-209 (DeiT-Tiny, E=8, top-2, stock Block.forward)."""
    moe_kwargs = _split_moe_kwargs(kwargs)
    model = deit_tiny_patch16_224(pretrained=pretrained, **kwargs)
    return patch_blocks_with_moe(model, 8, 2, False, **moe_kwargs)


# ---- named variants for the BASELINE.json configs (E / top_k are hard-coded in the reference's factories;
#      SURVEY.md section 5 'config / flags': the build registers one factory per benchmark configuration) ----
@register_model
def moe_tiny_patch16_224_expert4_top1(pretrained=False, **kwargs):
    """BASELINE cfg 1: ViT-Ti/16, E=4 (global), top-1."""
    moe_kwargs = _split_moe_kwargs(kwargs)
    model = deit_tiny_patch16_224(pretrained=pretrained, **kwargs)
    return patch_blocks_with_moe(model, 4 // moe_kwargs.get("world_size", 1), 1, False, **moe_kwargs)


@register_model
def moe_base_patch16_224_expert8_top1(pretrained=False, **kwargs):
    """BASELINE cfg 2: ViT-B/16 (models/model.py:163-183 shell), E=8 (global; E/world_size per rank), top-1."""
    moe_kwargs = _split_moe_kwargs(kwargs)
    model = _deit(768, 12, 12, pretrained=pretrained, **kwargs)
    return patch_blocks_with_moe(model, 8 // moe_kwargs.get("world_size", 1), 1, False, **moe_kwargs)


@register_model
def resmoe_base_patch16_224_expert8_top1(pretrained=False, starting_threshold=1.0, target_threshold=0.9, **kwargs):
    """ViT-B/16, E=8, top-1, with the token-skip gates and residual-on-normed forward."""
    moe_kwargs = _split_moe_kwargs(kwargs)
    model = _deit(768, 12, 12, pretrained=pretrained, **kwargs)
    return patch_blocks_with_moe(model, 8 // moe_kwargs.get("world_size", 1), 1, True, starting_threshold,
                                 target_threshold, **moe_kwargs)


@register_model
def moe_base_patch16_224_expert16_top1(pretrained=False, **kwargs):
    """BASELINE cfg 3: ViT-B/16, E=16, top-1."""
    moe_kwargs = _split_moe_kwargs(kwargs)
    ws = moe_kwargs.get("world_size", 1)
    model = _deit(768, 12, 12, pretrained=pretrained, **kwargs)
    return patch_blocks_with_moe(model, 16 // ws, 1, False, **moe_kwargs)


@register_model
def moe_large_patch16_384_expert32_top1(pretrained=False, **kwargs):
    """BASELINE cfg 4: ViT-L/16 @384 (models/vision_transformer.py:1227-1236 dims), E=32, top-1."""
    moe_kwargs = _split_moe_kwargs(kwargs)
    ws = moe_kwargs.get("world_size", 1)
    kwargs.setdefault("img_size", 384)
    model = _deit(1024, 24, 16, pretrained=pretrained, **kwargs)
    return patch_blocks_with_moe(model, 32 // ws, 1, False, **moe_kwargs)
