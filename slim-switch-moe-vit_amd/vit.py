"""Caller-side ViT shell (timm-free) that the MoE block plugs into.

The reference builds its models from timm pieces that are absent on both boxes:
``PatchEmbed``, ``DropPath``, ``Mlp``, ``trunc_normal_`` and the model registry.  This file restates
just enough of them -- with the reference's attribute names, so ``state_dict()`` keys match
models/vision_transformer.py:642-848 / models/model.py:80-183 -- for the MoE factories in resmoe.py to
patch.  It is the *caller* of the hot path.  Under fp16-autocast inference its dense pieces run on the library's own kernels too
(LayerNorm, qkv / projection / patch-embedding / head on the grouped MFMA GEMM with one row group, the attention kernel);
a shape those kernels do not cover falls back to a torch op and SAYS SO (one ``SlimMoEFallbackWarning`` per shape and
reason): a silent second backend would let a benchmark or a test run on vendor kernels unnoticed.
"""
from __future__ import annotations

import os
from functools import partial
from typing import Callable, Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._cache import StreamCache, param_version, register_shadow

_MODEL_REGISTRY: Dict[str, Callable] = {}
_LN_DIMS = (192, 384, 768, 1024)


def register_model(fn: Callable) -> Callable:
    """timm.models.registry.register_model: name -> factory (later registration wins)."""
    _MODEL_REGISTRY[fn.__name__] = fn
    return fn


def create_model(model_name: str, pretrained: bool = False, **kwargs):
    """timm.create_model(name, pretrained, **kwargs) as used at main.py:520-530 (None-valued kwargs dropped)."""
    if model_name not in _MODEL_REGISTRY:
        raise RuntimeError(f"Unknown model ({model_name}); registered: {sorted(_MODEL_REGISTRY)}")
    kwargs = {k: v for k, v in kwargs.items() if v is not None}
    return _MODEL_REGISTRY[model_name](pretrained=pretrained, **kwargs)


def list_models():
    return sorted(_MODEL_REGISTRY)


def trunc_normal_(t: torch.Tensor, std: float = 0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class DropPath(nn.Module):
    """Per-sample stochastic depth; identity in eval or when p == 0."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)

    def forward(self, x):
        assert tuple(x.shape[-2:]) == self.img_size, f"input {tuple(x.shape[-2:])} != model {self.img_size}"
        # Conv2d with kernel == stride is a per-patch linear map: same result as self.proj(x).flatten(2).transpose(1, 2),
        # written as one GEMM (no MIOpen algorithm search on first use)
        B, C, H, W = x.shape
        ph, pw = self.patch_size
        gh, gw = self.grid_size
        patches = x.reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * ph * pw)
        return F.linear(patches, self.proj.weight.reshape(self.proj.weight.shape[0], -1), self.proj.bias)


class Mlp(nn.Module):
    """Dense FFN each expert generalises (models/layers.py:391-414)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


def _autocast_half_inference(x: torch.Tensor) -> bool:
    return (x.is_cuda and not torch.is_grad_enabled() and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.float16)


class _HalfCache(StreamCache):
    """fp16 shadow of a parameter (what autocast would re-create on every forward), refreshed on change; every entry
    carries the event of the kernel that produced it (_cache.StreamCache), so another stream may consume it."""

    def offsets(self, rows: int, device) -> torch.Tensor:
        return super().get(("offs", rows, str(device)), 0,
                           lambda: torch.tensor([0, rows], dtype=torch.int32, device=device))

    def split_offsets(self, rows: int, parts: int, device) -> torch.Tensor:
        """[0, r_1, ..., rows]: ``rows`` cut into ``parts`` near-equal slices (multiples of 64 rows): the pseudo-groups of a
        single-group weight gradient (dense.LinearFn.backward)."""
        def make():
            step = -(-rows // parts)
            step = -(-step // 64) * 64
            cuts = [min(i * step, rows) for i in range(parts)] + [rows]
            return torch.tensor(cuts, dtype=torch.int32, device=device)
        return super().get(("split", rows, parts, str(device)), 0, make)

    def identity_rows(self, rows: int, device) -> torch.Tensor:
        """int64 [rows] = 0 .. rows-1: the row map under which the GEMM's per-row combine scale applies to plain rows."""
        return super().get(("ident", rows, str(device)), 0, lambda: torch.arange(rows, dtype=torch.int64, device=device))

    def get(self, p: torch.Tensor) -> torch.Tensor:  # noqa: D102
        def make():
            q = p.detach()
            if q.is_cuda and q.dtype == torch.float32 and q.is_contiguous() and q.numel() % 8 == 0:
                from . import ops
                return ops.cast(q, torch.float16)
            return q.half()
        if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous():
            register_shadow(p, self, id(p))            # optim.AdamW refreshes this image in its own pass
        return super().get(id(p), param_version(p), make)

    def get_padded(self, p: torch.Tensor, rows: int, dtype) -> torch.Tensor:
        """``p`` (a weight [N, K] or a bias [N]) in ``dtype`` with zero rows appended up to ``rows`` (a classifier head whose
        class count is not a multiple of 8 runs on the GEMM kernel with a padded N; the caller drops the extra columns)."""
        def make():
            t = p.detach().to(dtype).reshape(p.shape[0], -1)
            out = torch.zeros((rows, t.shape[1]), dtype=dtype, device=p.device)
            out[: t.shape[0]].copy_(t)
            return out if p.dim() > 1 else out.reshape(rows)
        return super().get(("pad", id(p), rows, dtype), param_version(p), make)

    def get_t(self, p: torch.Tensor) -> torch.Tensor:
        """fp16 image of ``p [N, K]`` transposed, ``[1, K, N]``: the operand of the backward's dgrad GEMM (it contracts
        over N), made in one HIP pass from the f32 master (N, K multiples of 64)."""
        from . import ops
        N = p.shape[0]

        def make():   # from the current fp16 image when there is one (half the bytes; same values), else from the master
            src = self.peek(id(p), param_version(p))
            if src is None:
                src = p.detach().float()
            return ops.transpose_cast(src.reshape(1, N, -1).contiguous(), torch.float16)
        return super().get(("t", id(p)), param_version(p), make)

    def get_t_padded(self, p: torch.Tensor, rows: int) -> torch.Tensor:
        """``get_t`` of ``p [N, K]`` with zero rows appended up to ``rows`` (a multiple of 64): ``[1, K, rows]`` fp16 -- the
        dgrad operand of a classifier head whose class count is not a multiple of 64."""
        from . import ops

        def make():
            t = torch.zeros((1, rows, p.shape[1]), dtype=torch.float32, device=p.device)
            t[0, : p.shape[0]].copy_(p.detach().reshape(p.shape[0], -1))
            return ops.transpose_cast(t, torch.float16)
        return super().get(("tpad", id(p), rows), param_version(p), make)

    def get_f32(self, p: torch.Tensor) -> torch.Tensor:
        """f32 view of a (bias) parameter: the parameter itself unless it is stored in another dtype."""
        if p.dtype == torch.float32:
            return p.detach()
        return super().get(("f32", id(p)), param_version(p), lambda: p.detach().float())


def _half_cache(mod: nn.Module) -> "_HalfCache":
    hc = mod.__dict__.get("_half")
    if hc is None:
        hc = mod.__dict__["_half"] = _HalfCache()
        mod.register_load_state_dict_post_hook(lambda m, _keys: m.__dict__["_half"].invalidate())
    return hc


DENSE_GEMM = os.environ.get("SLIMMOE_DENSE_GEMM", "own")   # "own": the grouped MFMA GEMM (one group); "blas": F.linear (A/B)


class SlimMoEFallbackWarning(UserWarning):
    """A dense piece of the eval forward left the hand-written kernels for a torch / vendor kernel."""


_fallbacks_seen = set()


def _warn_fallback(what: str, reason: str, shape) -> None:
    """One warning per (piece, reason, shape): the BASELINE shapes raise none (tests/test_gpu_model.py checks that)."""
    key = (what, reason, tuple(shape))
    if key in _fallbacks_seen:
        return
    _fallbacks_seen.add(key)
    import warnings
    warnings.warn(f"{what}: shape {tuple(shape)} runs on a torch / vendor kernel, not on libslimmoe_hip ({reason})",
                  SlimMoEFallbackWarning, stacklevel=3)


def _linear16_reason(x16: torch.Tensor, weight: torch.Tensor):
    """Why ``_linear16`` cannot take this GEMM (None = it can)."""
    M, K = x16.shape
    N = weight.shape[0]
    if DENSE_GEMM != "own":
        return f"SLIMMOE_DENSE_GEMM={DENSE_GEMM}"
    if K % 64:
        return "K % 64 != 0"
    if not x16.is_contiguous():
        return "rows not contiguous"
    return None


def _linear16(hc: "_HalfCache", x16: torch.Tensor, weight: torch.Tensor, bias, out_dtype=torch.float16, residual=None,
              name: str = "dense_gemm", row_scale=None):
    """``x16 @ weight^T + bias`` (+ residual) for fp16 rows ``x16 [M, K]`` on the hand-written grouped MFMA GEMM with a
    single row group -- the dense projections around the MoE (qkv, attention output, patch embedding, classifier head:
    models/vision_transformer.py:262-266, 276, 819, 847) -- returning ``[M, N]`` in ``out_dtype``.  Same arithmetic as
    the fp16 GEMM autocast would run (f16 operands, f32 accumulate, one rounding).  An N that is not a multiple of 8 (a
    10-class head) is padded with zero weight rows and the extra columns dropped.  ``row_scale`` (f32 [M]): row r of the
    product is multiplied by ``row_scale[r]`` before the residual is added (stochastic depth's per-sample ``mask / keep``,
    models/vision_transformer.py:320: it rides on the GEMM's fused-combine scale under an identity row map).  Returns None, after a
    SlimMoEFallbackWarning, when the shape is not the kernel's (K % 64), so the caller can fall back to ``F.linear``."""
    from . import ops
    M, K = x16.shape
    N = weight.shape[0]
    if M == 0:
        return None
    why = _linear16_reason(x16, weight)
    if why is not None:
        _warn_fallback(name, why, (M, K, N))
        return None
    if N % 8:
        assert residual is None and row_scale is None
        Np = (N + 7) // 8 * 8
        w = hc.get_padded(weight, Np, torch.float16)[None]
        b = hc.get_padded(bias, Np, torch.float32)[None] if bias is not None else None
        variant = 1 if M <= 1024 else ops.DEFAULT_GEMM_VARIANT
        out = ops.grouped_gemm(x16, w, b, hc.offsets(M, x16.device), ops.EPI_NONE, out_dtype, variant=variant, prof_name=name)
        return out[:, :N].contiguous()
    w = hc.get(weight).reshape(N, K)[None]
    b = hc.get_f32(bias)[None] if bias is not None else None
    # a handful of rows (the classifier head sees one row per image): 128 x 128 tiles spread the few output tiles over
    # more CUs than the 320 x 256 tile of the big GEMMs would
    if row_scale is not None:
        out = torch.empty((M, N), dtype=out_dtype, device=x16.device)
        return ops.grouped_gemm(x16, w, b, hc.offsets(M, x16.device), ops.EPI_NONE, out_dtype, residual=residual,
                                row_map=hc.identity_rows(M, x16.device), row_scale=row_scale, out=out,
                                variant=ops.DEFAULT_GEMM_VARIANT, prof_name=name)
    variant = 1 if M <= 1024 else ops.DEFAULT_GEMM_VARIANT
    return ops.grouped_gemm(x16, w, b, hc.offsets(M, x16.device), ops.EPI_NONE, out_dtype, residual=residual,
                            variant=variant, prof_name=name)


class Attention(nn.Module):
    """models/vision_transformer.py:248-280 (softmax(q k^T * scale) v, then proj)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, residual=None, row_scale=None):
        """``attn(x)``; with ``residual`` the pair ``(out, added)``: ``(residual + s * attn(x), True)`` when the add (and the
        optional per-row factor ``s`` = ``row_scale`` f32 [B*N], stochastic depth's ``mask / keep``) was fused into the
        projection GEMM's store, else ``(attn(x), False)`` -- unscaled; the caller scales and adds.

        Under fp16 autocast -- inference or training -- every piece runs on the library's kernels whether or not a residual is
        handed in (a block whose stochastic depth is merely *inactive* calls without one); anything else goes to
        ``_forward`` = ``nn.Linear`` + ``F.scaled_dot_product_attention`` and says so once (SlimMoEFallbackWarning)."""
        B, N, C = x.shape
        tr = self._forward_train(x, residual, row_scale)
        if tr is not None:
            return tr if residual is not None else tr[0]
        if _autocast_half_inference(x) and x.dtype in (torch.float16, torch.float32) and not (
                self.training and (self.attn_drop.p > 0 or self.proj_drop.p > 0)):
            out, added = self._forward_infer16(x, residual, row_scale)
            return (out, added) if residual is not None else out
        if x.is_cuda and torch.is_autocast_enabled():
            _warn_fallback("attention", "config: needs fp16 autocast without attention / projection dropout in training", (B, N, C))
        return (self._forward(x), False) if residual is not None else self._forward(x)

    def _forward_infer16(self, x, residual, row_scale):
        """fp16-autocast inference on the own kernels: the arithmetic autocast runs (fp16 GEMM operands, f32 accumulate, one
        rounding), without re-casting the weights on every call.  -> (out, added)."""
        from . import ops
        B, N, C = x.shape
        hc = _half_cache(self)
        x2 = x.reshape(B * N, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if x2.dtype != torch.float16:
            x2 = ops.cast(x2, torch.float16)      # the cast autocast puts in front of the qkv GEMM
        qkv = _linear16(hc, x2, self.qkv.weight, self.qkv.bias, name="qkv_gemm")
        if qkv is None:
            qkv = F.linear(x2, hc.get(self.qkv.weight), hc.get(self.qkv.bias) if self.qkv.bias is not None else None)
        hd = C // self.num_heads
        if ops.attention_supported(N, hd) and qkv.is_contiguous():
            # hand-written attention on the fused qkv layout [B,N,3,H,hd] (no q/k/v transposes)
            o = ops.attention(qkv, B, N, self.num_heads, hd, self.scale).reshape(B * N, C)
        else:
            _warn_fallback("attention", "kernel covers head_dim 64, N <= 640" if qkv.is_contiguous() else "qkv not contiguous",
                           (N, hd))
            q, k, v = qkv.reshape(B, N, 3, self.num_heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
            o = F.scaled_dot_product_attention(q, k, v, scale=self.scale).transpose(1, 2).reshape(B * N, C)
        if residual is not None:
            if C % 64 == 0 and residual.dtype == torch.float32 and residual.is_contiguous():
                # projection + bias + residual add in one launch of the grouped MFMA GEMM (a single group):
                # residual + proj(o), f32 out -- the same arithmetic as the unfused `x + attn(...)`
                out = _linear16(hc, o, self.proj.weight, self.proj.bias, torch.float32,
                                residual=residual.reshape(B * N, C), row_scale=row_scale, name="attn_proj_gemm")
                if out is not None:
                    return out.reshape(B, N, C), True
            else:
                _warn_fallback("attn_proj_gemm", "residual must be contiguous f32 and C % 64 == 0", (B * N, C, C))
        out = _linear16(hc, o, self.proj.weight, self.proj.bias, name="attn_proj_gemm")
        if out is None:
            out = F.linear(o, hc.get(self.proj.weight), hc.get(self.proj.bias) if self.proj.bias is not None else None)
        return out.reshape(B, N, C), False

    def _forward_train(self, x, residual, row_scale=None):
        """The training step's attention half on the library's kernels, forward AND backward (dense.py): fp16 rows from the
        HIP LayerNorm -> qkv GEMM -> attention -> projection GEMM (+ the f32 residual, and stochastic depth's per-row factor, in
        its store).  None when a precondition does not hold (the caller then takes torch's path, loudly): fp16 autocast with
        gradients, no attention / projection dropout (the reference's defaults, main.py --drop 0.0), shapes the kernels cover."""
        from . import dense, ops
        if not (dense.autocast_half_training(x) and x.dtype in (torch.float16, torch.float32)):
            return None
        B, N, C = x.shape
        if self.training and (self.attn_drop.p > 0 or self.proj_drop.p > 0):
            _warn_fallback("attention (training)", "config: attention / projection dropout > 0", (B, N, C))
            return None
        hd = C // self.num_heads
        if x.dtype == torch.float32:
            x = dense.cast16(x)                      # the cast autocast puts in front of the qkv GEMM (backward: cast back)
        x2 = x.reshape(B * N, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        if not (dense.linear_supported(x2, self.qkv.weight) and dense.linear_supported(x2, self.proj.weight)
                and ops.attention_supported(N, hd) and ops.attention_bwd_supported(N, hd) and DENSE_GEMM == "own"):
            _warn_fallback("attention (training)", "shape outside the backward kernels' reach (N <= 256, head dim 64, C % 64 == 0)",
                           (B, N, C, self.num_heads))
            return None
        hc = _half_cache(self)
        qkv = dense.LinearFn.apply(x2, self.qkv.weight, self.qkv.bias, None, hc, torch.float16, "qkv_gemm")
        o = dense.AttentionFn.apply(qkv, B, N, self.num_heads, hd, self.scale).reshape(B * N, C)
        if residual is not None and residual.dtype == torch.float32 and residual.is_contiguous():
            out = dense.LinearFn.apply(o, self.proj.weight, self.proj.bias, residual.reshape(B * N, C), hc, torch.float32,
                                       "attn_proj_gemm", row_scale)
            return out.reshape(B, N, C), True
        out = dense.LinearFn.apply(o, self.proj.weight, self.proj.bias, None, hc, torch.float16, "attn_proj_gemm")
        return out.reshape(B, N, C), False

    def _forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4).unbind(0)
        p = self.attn_drop.p if self.training else 0.0
        x = F.scaled_dot_product_attention(q, k, v, dropout_p=p, scale=self.scale)
        return self.proj_drop(self.proj(x.transpose(1, 2).reshape(B, N, C)))


class Block(nn.Module):
    """models/vision_transformer.py:283-322; ``mlp`` is what the MoE factories replace."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, num_tokens=-1):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def _norm1(self, x):
        """norm1 for the attention half; under fp16-autocast inference the LayerNorm writes fp16 directly (the cast
        autocast would insert in front of the qkv GEMM), through the same HIP LayerNorm the MoE half uses."""
        n = self.norm1
        if (isinstance(n, nn.LayerNorm) and n.elementwise_affine and x.shape[-1] in _LN_DIMS and x.is_contiguous()
                and x.dtype == torch.float32 and _autocast_half_inference(x)):
            from . import ops
            return ops.layernorm(x, n.weight.detach(), n.bias.detach() if n.bias is not None else None, n.eps,
                                 torch.float16)
        from . import dense
        if dense.autocast_half_training(x) and dense.layer_norm_supported(x, n):
            # training: the same HIP LayerNorm (fp16 rows for the qkv GEMM) with its HIP backward
            return dense.layer_norm(x, n, torch.float16)
        return n(x)

    def forward(self, x):
        from .ep import drain
        return drain(self.forward_steps(x))

    def stochastic_depth_inactive(self) -> bool:
        """``drop_path`` is the identity right now: the module IS ``nn.Identity`` (rate 0: models/vision_transformer.py:308), or
        it is a ``DropPath`` in eval mode or with probability 0 (timm's DropPath returns its input then).  The fast paths below
        are gated on THIS, not on the module's type: the reference's default is ``--drop-path 0.1`` (main.py:74-79), and such a
        model evaluates exactly like a rate-0 one."""
        dp = self.drop_path
        if isinstance(dp, nn.Identity):
            return True
        return isinstance(dp, DropPath) and (dp.drop_prob == 0.0 or not dp.training)

    def _depth_scale(self, x):
        """One draw of stochastic depth for a [B, N, d] branch: (per-sample factor f32 [B] = bernoulli(keep) / keep, the same per
        row f32 [B * N]); (None, None) when inactive.  None as well for a foreign ``drop_path`` module (the caller then
        applies the module itself)."""
        dp = self.drop_path
        if self.stochastic_depth_inactive() or not isinstance(dp, DropPath):
            return None, None
        keep = 1.0 - dp.drop_prob
        mask = torch.empty(x.shape[0], dtype=torch.float32, device=x.device).bernoulli_(keep)
        if x.is_cuda:      # the division and the expansion to rows in one launch (the same IEEE division as div_)
            from . import ops
            return ops.depth_scale_rows(mask, keep, x.shape[1])
        f = mask.div_(keep)
        return f, f.repeat_interleave(x.shape[1])

    def forward_steps(self, x, xn1=None, next_norm=None):
        """The block as a generator (result = return value): it yields only inside an expert-parallel MoE ``mlp``, at
        the points where this micro-batch waits for the host or an all-to-all (ep.ep_forward_steps).

        ``xn1``: ``norm1(x)`` already computed by the previous block's combine (see ``next_norm``).  ``next_norm``: the NEXT
        block's ``norm1``; when this block's expert-parallel combine can produce that LayerNorm in the same pass
        (smoe_gather_combine_ln) the result is ``(x, norm1_next(x))`` instead of ``x``.

        Stochastic depth (``x + drop_path(f(norm(x)))``, models/vision_transformer.py:319-322): inactive = the plain fused paths;
        active (training) = the per-sample ``mask / keep`` factor travels as a per-row scale into the stores that add the
        residual (projection GEMM / MoE combine) -- the add stays fused, forward and backward."""
        from . import dense
        inactive = self.stochastic_depth_inactive()
        own_dp = inactive or isinstance(self.drop_path, DropPath)
        train = dense.autocast_half_training(x)
        xres = x
        if xn1 is not None:
            xin = xn1
        elif train and x.requires_grad and dense.layer_norm_supported(x, self.norm1) and own_dp:
            # training: LayerNorm and the bypass from one Function, so that its backward kernel also adds the bypass' gradient
            xin, xres = dense.layer_norm_res(x, self.norm1, torch.float16)
        else:
            xin = self._norm1(x)
        if own_dp and x.is_contiguous() and isinstance(self.attn, Attention):
            f_b, f_rows = self._depth_scale(x)
            a, added = self.attn(xin, residual=xres, row_scale=f_rows)
            if added:
                x = a
            else:
                x = xres + (a if f_b is None else a * f_b.view(-1, 1, 1).to(a.dtype))
        else:
            if x.is_cuda and torch.is_autocast_enabled():
                _warn_fallback("block (attention half)", "config: foreign drop_path / attention module or non-contiguous input",
                               x.shape)
            x = x + self.drop_path(self.attn(xin))
        if train and dense.layer_norm_supported(x, self.norm2) and getattr(self.mlp, "forward_add", None) is not None and own_dp:
            # training: HIP LayerNorm (f32 rows: the router routes on them) forward and backward, then the MoE operator's
            # training path; the residual add (and the stochastic-depth factor) rides in the operator's combine
            f_b, f_rows = self._depth_scale(x)
            if x.requires_grad:
                xn, xres2 = dense.layer_norm_res(x, self.norm2, torch.float32)
                return self.mlp.forward_add(xn, xres2, row_scale=f_rows)
            xn = dense.layer_norm(x, self.norm2, torch.float32)
            return self.mlp.forward_add(xn, x, row_scale=f_rows)
        if inactive:
            steps = getattr(self.mlp, "forward_norm_add_steps", None)
            if steps is not None:
                # x + mlp(norm2(x)): LN + router, scatter, combine + add all fused
                fuse_next = (next_norm is not None and _autocast_half_inference(x) and isinstance(next_norm, nn.LayerNorm)
                             and next_norm.elementwise_affine and x.dtype == torch.float32 and x.shape[-1] in _LN_DIMS)
                return (yield from steps(x, self.norm2, next_norm=next_norm if fuse_next else None))
            fused = getattr(self.mlp, "forward_add", None)
            if fused is not None:
                return fused(self.norm2(x), x)  # x + mlp(norm2(x)), add fused into the MoE combine store
        elif x.is_cuda and torch.is_autocast_enabled():
            _warn_fallback("block (MLP half)", "config: stochastic depth active outside the fp16-autocast training path", x.shape)
        x = x + self.drop_path(self.mlp(self.norm2(x)))
        return x


def _init_vit_weights(m: nn.Module):
    # non-jax branch of models/vision_transformer.py:851-885
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.LayerNorm):
        nn.init.zeros_(m.bias)
        nn.init.ones_(m.weight)


class VisionTransformer(nn.Module):
    """Non-distilled ViT (models/vision_transformer.py:642-848); unknown kwargs are swallowed as there."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
                 embed_layer=PatchEmbed, norm_layer=None, act_layer=None, **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.dist_token = None
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, act_layer=act_layer,
                  num_tokens=num_patches + self.num_tokens)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.head_dist = None
        trunc_normal_(self.pos_embed, std=0.02)
        trunc_normal_(self.cls_token, std=0.02)
        self.apply(_init_vit_weights)

    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token"}

    def _first_norm1(self):
        """Block 0's norm1 if the embedding pass may produce it (a stock ``Block`` whose forward was not replaced: it takes
        ``xn1``; a LayerNorm with affine parameters over the embedding width), else None."""
        blk = self.blocks[0] if len(self.blocks) else None
        if blk is None or not hasattr(blk, "forward_steps") or "forward" in blk.__dict__:
            return None
        n = blk.norm1
        ok = (isinstance(n, nn.LayerNorm) and n.elementwise_affine and tuple(n.normalized_shape) == (self.embed_dim,)
              and n.weight.dtype == torch.float32)
        return n if ok else None

    def _embed(self, x, want_xn1: bool = False):
        """``pos_drop(cat(cls_token, patch_embed(x)) + pos_embed)`` (models/vision_transformer.py:818-824).  Under
        fp16-autocast inference the same arithmetic in three launches fewer: patch gather and the fp16 cast autocast
        puts in front of the projection in one copy, cached fp16 weights, and the class-token concat folded into the
        position-embedding add (fp16 tokens + f32 embedding -> f32 stream, as the mixed-dtype cat / add promote)."""
        pe = self.patch_embed
        if (type(pe) is PatchEmbed and _autocast_half_inference(x) and x.dtype == torch.float32 and not self.training
                and tuple(x.shape[-2:]) == pe.img_size):
            from . import ops
            hc = _half_cache(self)
            B, C = x.shape[0], x.shape[1]
            (ph, pw), (gh, gw) = pe.patch_size, pe.grid_size
            dm = pe.proj.weight.shape[0]
            own = pw % 4 == 0 and dm in _LN_DIMS and x.is_contiguous() and self.pos_embed.dtype == torch.float32
            if own:   # patch gather + the cast autocast puts in front of the projection: one HIP pass (reads the image once)
                p2 = ops.patchify_cast(x, ph, pw, torch.float16)
            else:
                _warn_fallback("patch embedding", "patch width % 4 != 0 or an embedding width outside {192, 384, 768, 1024}", x.shape)
                p16 = torch.empty((B, gh, gw, C, ph, pw), dtype=torch.float16, device=x.device)
                p16.copy_(x.reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5))
                p2 = p16.reshape(B * gh * gw, C * ph * pw)
            w = hc.get(pe.proj.weight).reshape(pe.proj.weight.shape[0], -1)
            tok = _linear16(hc, p2, pe.proj.weight, pe.proj.bias, name="patch_embed_gemm")
            if tok is None:
                tok = F.linear(p2, w, hc.get(pe.proj.bias) if pe.proj.bias is not None else None)
            if own and tok.is_contiguous():
                # class-token row, position embedding and -- when the first block takes it -- that block's norm1, in one pass
                n1 = self._first_norm1()
                ln = (n1.weight.detach(), n1.bias.detach() if n1.bias is not None else None, n1.eps) if (want_xn1 and n1 is not None) else None
                out, xn1 = ops.embed_ln(tok, self.cls_token, self.pos_embed, B, gh * gw, ln=ln)
                return (out, xn1) if want_xn1 else out
            tok = tok.reshape(B, gh * gw, -1)
            out = torch.empty((B, gh * gw + 1, tok.shape[-1]), dtype=torch.float32, device=x.device)
            torch.add(tok, self.pos_embed[:, 1:], out=out[:, 1:])
            out[:, 0] = self.cls_token[0, 0] + self.pos_embed[0, 0]
            return (out, None) if want_xn1 else out
        from . import dense
        if (type(pe) is PatchEmbed and dense.autocast_half_training(x) and x.dtype == torch.float32
                and tuple(x.shape[-2:]) == pe.img_size and DENSE_GEMM == "own"):
            # training: the per-patch projection on the own GEMM, forward and backward (the images need no gradient)
            B, C = x.shape[0], x.shape[1]
            (ph, pw), (gh, gw) = pe.patch_size, pe.grid_size
            p16 = torch.empty((B, gh, gw, C, ph, pw), dtype=torch.float16, device=x.device)
            p16.copy_(x.detach().reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5))
            p2 = p16.reshape(B * gh * gw, C * ph * pw)
            if dense.linear_supported(p2, pe.proj.weight):
                tok = dense.LinearFn.apply(p2, pe.proj.weight, pe.proj.bias, None, _half_cache(self), torch.float16,
                                           "patch_embed_gemm").reshape(B, gh * gw, -1)
                x = torch.cat((self.cls_token.expand(B, -1, -1), tok.float()), dim=1)
                x = self.pos_drop(x + self.pos_embed)
                return (x, None) if want_xn1 else x
        x = pe(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        x = self.pos_drop(x + self.pos_embed)
        return (x, None) if want_xn1 else x

    def forward_features(self, x):
        n = self._ep_pipeline_depth(x)
        if self._ep_blocks():
            # the static exchange's buffers are agreed and sized per micro-batch (ep.static_slot_tokens): the modules must know how
            # the local batch is cut -- shared configuration (depth, gates, ep_micro_batches), so every rank says the same
            for blk in self.blocks:
                m = getattr(blk, "mlp", None)
                if hasattr(m, "ep_active"):
                    m.ep_rows_div = n
                    m.ep_rows_unit = int(self.pos_embed.shape[1])
        if n > 1:
            return self._forward_features_pipelined(x, n)
        if (int(self.compute_streams) == 2 and x.is_cuda and not torch.is_grad_enabled() and not self.training
                and x.shape[0] >= 2):
            return self._forward_features_two_streams(x)
        x, xn1 = self._embed(x, want_xn1=True)
        if x.is_cuda and not torch.is_grad_enabled():   # (generator form: every block is handed the next block's norm1)
            from .ep import drain
            x = drain(self._blocks_steps(x, xn1))
        else:
            x = self.blocks(x)
        return self.pre_logits(self._final_norm_cls(x))

    def _final_norm_cls(self, x):
        """``self.norm(x)[:, 0]`` (models/vision_transformer.py:826-830).  LayerNorm is per token, so normalising the
        class token alone gives the identical row and skips a pass over the other 196 tokens of every image."""
        if isinstance(self.norm, nn.LayerNorm):
            from . import dense
            if dense.autocast_half_training(x):
                cls = x[:, 0].contiguous()
                if dense.layer_norm_supported(cls, self.norm):
                    return dense.layer_norm(cls, self.norm, torch.float32)
            if (_autocast_half_inference(x) and x.dtype == torch.float32 and self.norm.elementwise_affine
                    and x.shape[-1] in _LN_DIMS):
                from . import ops
                n = self.norm
                if x.is_contiguous() and n.weight.dtype == torch.float32:   # the class-token rows read in place, (P + 1) * d apart
                    return ops.layernorm_rows(x, x.shape[1] * x.shape[2], x.shape[0], x.shape[2], n.weight.detach(),
                                              n.bias.detach() if n.bias is not None else None, n.eps)
                return ops.layernorm(x[:, 0].contiguous(), n.weight.detach(), n.bias.detach() if n.bias is not None else None,
                                     n.eps, torch.float32)
            return self.norm(x[:, 0])
        return self.norm(x)[:, 0]

    # -- optional: the two halves of the batch on two compute streams -----------------------------------------------
    compute_streams = 1  # opt-in (2): one half's partly filled last round of workgroups runs beside the other half's
                         # next kernel (+2.5-3.4 % images/s measured in round 1); eval / no-grad only

    def _forward_features_two_streams(self, x):
        """Block by block, half 0 on side stream 0 and half 1 on side stream 1 (images are independent in eval mode, so
        each half computes exactly what it computes alone).  Cross-stream hand-offs: the side streams wait for the
        caller's stream before touching ``x`` and the caller's stream waits for both before the concat; tensors that
        cross streams are registered with the allocator (record_stream); derived tensors shared by both halves (16-bit
        weight shadows, constant tables) carry their producer's event (_cache.StreamCache)."""
        cur = torch.cuda.current_stream(x.device)
        side = self.__dict__.get("_side_streams")
        if side is None or side[0].device != x.device:
            side = self.__dict__["_side_streams"] = [torch.cuda.Stream(x.device), torch.cuda.Stream(x.device)]
        fork = torch.cuda.Event()
        fork.record(cur)
        halves = list(x.chunk(2, dim=0))
        for st, xb in zip(side, halves):
            st.wait_event(fork)
            xb.record_stream(st)
        for i, st in enumerate(side):
            with torch.cuda.stream(st):
                halves[i] = self._embed(halves[i])
        for blk in self.blocks:
            for i, st in enumerate(side):
                with torch.cuda.stream(st):
                    halves[i] = blk(halves[i])
        outs = []
        for i, st in enumerate(side):
            with torch.cuda.stream(st):
                o = self.pre_logits(self._final_norm_cls(halves[i]))
            done = torch.cuda.Event()
            done.record(st)
            cur.wait_event(done)
            o.record_stream(cur)
            outs.append(o)
        return torch.cat(outs, dim=0)

    # -- expert-parallel inference: software pipeline over micro-batches ------------------------------
    ep_micro_batches = 2

    def _ep_pipeline_depth(self, x) -> int:
        """Micro-batches to interleave.  > 1 only for inference with expert-parallel MoE blocks whose gates never
        drop (capacity is defined over the whole local batch, so a dropping gate must see it whole)."""
        n = int(self.ep_micro_batches)
        if n <= 1 or torch.is_grad_enabled() or self.training or not x.is_cuda or x.shape[0] < n:
            return 1
        ep = False
        for blk in self.blocks:
            m = getattr(blk, "mlp", None)
            if not hasattr(blk, "forward_steps") or "forward" in blk.__dict__:  # e.g. resmoe.forward_residule_moe
                return 1
            if hasattr(m, "ep_active") and m.ep_active():
                if m.gate.capacity(1 << 20) >= 0:
                    return 1
                ep = True
        return n if ep else 1

    def _blocks_steps(self, x, xn0=None):
        """``self.blocks(x)`` as a generator, handing every block the next block's ``norm1`` so that an expert-parallel
        combine can produce it on the way out (one pass over the residual stream less per layer)."""
        blocks = list(self.blocks)
        xn = xn0                 # norm1(x) of block 0 when the embedding pass produced it (_embed)
        for i, blk in enumerate(blocks):
            if not hasattr(blk, "forward_steps") or "forward" in blk.__dict__:   # e.g. resmoe.forward_residule_moe
                x, xn = blk(x), None
                continue
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            nn1 = nxt.norm1 if (nxt is not None and hasattr(nxt, "forward_steps") and "forward" not in nxt.__dict__) else None
            r = yield from blk.forward_steps(x, xn1=xn, next_norm=nn1)
            x, xn = r if isinstance(r, tuple) else (r, None)
        return x

    def _ep_blocks(self) -> bool:
        return any(hasattr(getattr(b, "mlp", None), "ep_active") and b.mlp.ep_active() for b in self.blocks)

    def _features_steps(self, x):
        x, xn1 = self._embed(x, want_xn1=True)
        x = yield from self._blocks_steps(x, xn1)
        return self.pre_logits(self._final_norm_cls(x))

    def _forward_features_pipelined(self, x, n: int):
        """Round-robin over n micro-batches of the local batch; each runs until its next wait point (count read-back,
        dispatch / return all-to-all), so the compute stream always holds the other micro-batch's kernels while one
        waits, and RCCL's stream moves one micro-batch's rows under the other's GEMMs.  The schedule depends only
        on n and the depth, never on the routing: every rank issues its collectives in the same order.  Images are
        independent in eval mode, so the result equals the un-pipelined forward row for row."""
        from collections import deque
        gens = deque((i, self._features_steps(xb)) for i, xb in enumerate(x.chunk(n, dim=0)))
        outs = [None] * len(gens)
        while gens:
            i, g = gens.popleft()
            try:
                next(g)
                gens.append((i, g))
            except StopIteration as stop:
                outs[i] = stop.value
        return torch.cat(outs, dim=0)

    def forward(self, x):
        f = self.forward_features(x)
        if isinstance(self.head, nn.Linear) and _autocast_half_inference(f):
            hc = _half_cache(self)  # what autocast computes, minus the per-call weight casts
            if f.dtype == torch.float32 and f.is_contiguous():
                from . import ops
                f16 = ops.cast(f, torch.float16)
            else:
                f16 = f.to(torch.float16)
            out = _linear16(hc, f16.reshape(-1, f16.shape[-1]), self.head.weight, self.head.bias, name="head_gemm") if f16.dim() == 2 else None
            if out is not None:
                return out
            return F.linear(f16, hc.get(self.head.weight),
                            hc.get(self.head.bias) if self.head.bias is not None else None)
        from . import dense
        if isinstance(self.head, nn.Linear) and f.dim() == 2 and dense.autocast_half_training(f) and DENSE_GEMM == "own":
            f16 = dense.cast16(f) if f.dtype == torch.float32 else f.contiguous()
            if dense.linear_supported(f16, self.head.weight):
                return dense.LinearFn.apply(f16, self.head.weight, self.head.bias, None, _half_cache(self), torch.float16,
                                            "head_gemm")
            _warn_fallback("head_gemm (training)", "K % 64 != 0", (f.shape[0], f.shape[1], self.head.weight.shape[0]))
        elif isinstance(self.head, nn.Linear) and f.is_cuda and torch.is_autocast_enabled():
            _warn_fallback("head_gemm", "config: needs fp16 autocast and a [B, d] feature matrix", tuple(f.shape))
        return self.head(f)


def _deit(embed_dim, default_depth, num_heads, pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained weights need network access (torch.hub); not available offline")
    depth = kwargs.pop("depth", default_depth)  # tests build shallow copies of the big configurations
    return VisionTransformer(patch_size=16, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


@register_model
def deit_tiny_patch16_224(pretrained=False, **kwargs):
    """models/model.py:80-100."""
    return _deit(192, 12, 3, pretrained, **kwargs)


@register_model
def deit_base_patch16_224(pretrained=False, **kwargs):
    """models/model.py:163-183."""
    return _deit(768, 12, 12, pretrained, **kwargs)
