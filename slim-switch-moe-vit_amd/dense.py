"""Training-side autograd Functions for the dense half of the ViT block on the library's own kernels.

The reference trains the whole model forward + backward under fp16 autocast (engine.py:52-74); the block is
models/vision_transformer.py:283-322: LayerNorm -> qkv -> attention -> proj (+ residual) -> LayerNorm -> MoE (+ residual).
Under ``torch.autocast`` torch would run these as ``native_layer_norm`` (f32), casts, hipBLASLt GEMMs and an aotriton
attention, forward and backward.  Here:

    LayerNormFn   forward = smoe_layernorm (f32 rows -> f16 or f32, the cast autocast inserts fused);
                  backward = smoe_layernorm_bwd (statistics recomputed from x; dgamma / dbeta deterministic), optionally with
                  the gradient of the residual connection that bypasses the norm added in the same pass
    LinearFn      forward = smoe_grouped_gemm with one row group (f16 operands, f32 accumulate -- what autocast's GEMM
                  computes), optional f32 residual added in the store;
                  backward: dX = dY W through the same GEMM on the transposed weight image (smoe_transpose_cast),
                  dW = dY^T X by smoe_grouped_wgrad_rows with one group, db = smoe_group_colsum
    AttentionFn   forward = smoe_attention_fwd on the fused [B, N, 3, H, 64] qkv layout;
                  backward = smoe_attention_bwd (scores recomputed per (image, head); dqkv in the same fused layout)

Shapes the kernels do not cover are reported by ``supported_*`` so that the caller keeps torch's path for them (loudly:
vit.SlimMoEFallbackWarning)."""
from __future__ import annotations

import torch

import os

from . import ops

TRAIN_BACKEND = os.environ.get("SLIMMOE_DENSE_TRAIN", "own")   # "own": the kernels below; "torch": autocast's own ops (A/B, tests)


def autocast_half_training(x: torch.Tensor) -> bool:
    return (TRAIN_BACKEND == "own" and x.is_cuda and torch.is_grad_enabled() and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.float16)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        ctx.eps = eps
        ctx.save_for_backward(x, weight)
        return ops.layernorm(x, weight.detach().float() if weight is not None else None,
                             bias.detach().float() if bias is not None else None, eps, out_dtype)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw, db = ops.layernorm_bwd(x, dy.contiguous(), weight.detach().float() if weight is not None else None, ctx.eps)
        return dx, dw.to(weight.dtype) if weight is not None else None, db if ctx.needs_input_grad[2] else None, None, None


class LayerNormResFn(torch.autograd.Function):
    """``(norm(x), x)``: the pre-norm block reads x twice -- through the LayerNorm and over the residual connection that
    bypasses it (models/vision_transformer.py:320-321) -- and autograd would add the two gradients with a kernel of its own
    (a 155-MB read-modify-write per half block).  Returning the bypass from the same Function lets the LayerNorm backward add
    the residual's gradient while it writes dx (smoe_layernorm_bwd's ``dres``)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        ctx.eps = eps
        ctx.save_for_backward(x, weight)
        y = ops.layernorm(x, weight.detach().float() if weight is not None else None,
                          bias.detach().float() if bias is not None else None, eps, out_dtype)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, weight = ctx.saved_tensors
        if dy is None:      # only the bypass was used
            return dres, None, None, None, None
        if dres is not None and not (dres.dtype == torch.float32 and dres.is_contiguous()):
            dres = dres.float().contiguous()
        dx, dw, db = ops.layernorm_bwd(x, dy.contiguous(), weight.detach().float() if weight is not None else None, ctx.eps,
                                       dres=dres)
        return dx, dw.to(weight.dtype) if weight is not None else None, db if ctx.needs_input_grad[2] else None, None, None


def layer_norm_res(x: torch.Tensor, norm: torch.nn.LayerNorm, out_dtype: torch.dtype):
    """``(norm(x), x)`` with the two gradients of x summed inside the LayerNorm backward kernel."""
    return LayerNormResFn.apply(x, norm.weight, norm.bias, norm.eps, out_dtype)


def layer_norm(x: torch.Tensor, norm: torch.nn.LayerNorm, out_dtype: torch.dtype) -> torch.Tensor:
    """``norm(x)`` for contiguous f32 rows on the HIP LayerNorm with its HIP backward; output in ``out_dtype``."""
    return LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps, out_dtype)


def layer_norm_supported(x: torch.Tensor, norm) -> bool:
    return (isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine and x.is_cuda and x.dtype == torch.float32
            and x.is_contiguous() and x.shape[-1] in ops.LN_DIMS and tuple(norm.normalized_shape) == (x.shape[-1],))


class _Cast16Fn(torch.autograd.Function):
    """f32 -> f16 rows (the cast autocast inserts in front of a GEMM); backward: the f16 gradient back in f32."""

    @staticmethod
    def forward(ctx, x):
        return ops.cast(x.contiguous(), torch.float16)

    @staticmethod
    def backward(ctx, dy):
        return ops.cast(dy.contiguous(), torch.float32)


def cast16(x: torch.Tensor) -> torch.Tensor:
    return _Cast16Fn.apply(x)


def _pad64(n: int) -> int:
    return -(-n // 64) * 64


class LinearFn(torch.autograd.Function):
    """``x16 [M, K] @ weight[N, K]^T + bias`` (+ residual, f32) with f16 operands; ``cache`` is the module's _HalfCache.

    ``row_scale`` (f32 [M], no gradient; needs ``residual``): ``residual + row_scale[r] * (x16 W^T + bias)[r]`` -- stochastic
    depth's per-sample ``mask / keep`` (timm DropPath, models/vision_transformer.py:308,320) riding on the GEMM's combine scale;
    the backward scales dY by it in the pass that casts dY to 16 bit.

    N % 64 != 0 (the 1000-class head, models/vision_transformer.py:847): the backward's two GEMMs contract over N / write N
    rows, so they run on zero-padded images (weight rows / dY columns up to the next multiple of 64) and the padding is dropped
    from dW; the forward pads N to a multiple of 8 as the inference path does."""

    @staticmethod
    def forward(ctx, x16, weight, bias, residual, cache, out_dtype, name, row_scale=None):
        from .vit import _linear16
        if row_scale is not None and residual is None:
            raise RuntimeError(f"{name}: row_scale rides on the fused residual store; pass the residual")
        out = _linear16(cache, x16, weight, bias, out_dtype, residual=residual, name=name, row_scale=row_scale)
        if out is None:
            raise RuntimeError(f"{name}: shape {tuple(x16.shape)} x {tuple(weight.shape)} is outside the GEMM kernel's reach "
                               "(check linear_supported first)")
        ctx.cache, ctx.has_bias, ctx.has_res, ctx.name = cache, bias is not None, residual is not None, name
        ctx.save_for_backward(x16, weight, row_scale if row_scale is not None else torch.empty(0, device=x16.device))
        ctx.has_scale = row_scale is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        x16, weight, row_scale = ctx.saved_tensors
        N, K = weight.shape[0], x16.shape[1]
        M = x16.shape[0]
        dy = dy.contiguous()
        dres = dy if ctx.has_res else None                   # the residual's gradient is the output's
        Np = _pad64(N)
        if ctx.has_scale:
            # d(branch) = row_scale * dY, cast to 16 bit in the same pass (smoe_scatter_rows under the identity map)
            dy16 = ops.scatter_rows(dy, ctx.cache.identity_rows(M, dy.device), 1, torch.float16, scale=row_scale)
        elif dy.dtype == torch.float16:
            dy16 = dy
        else:
            dy16 = ops.cast(dy, torch.float16)
        offs = ctx.cache.offsets(M, x16.device)
        if Np != N:                                          # zero columns up to a multiple of 64 (a handful of rows: the head)
            pad = torch.zeros((M, Np), dtype=torch.float16, device=dy16.device)
            pad[:, :N].copy_(dy16)
            dy16 = pad
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.group_colsum(dy16, offs)[0][:N]
        dx = None
        if ctx.needs_input_grad[0]:
            # [1, K, Np] f16: the dgrad GEMM contracts over N
            wt = ctx.cache.get_t(weight) if Np == N else ctx.cache.get_t_padded(weight, Np)
            dx = ops.grouped_gemm(dy16, wt, None, offs, ops.EPI_NONE, torch.float16, variant=ops.DEFAULT_GEMM_VARIANT,
                                  prof_name=ctx.name + "_dgrad")
        dw = None
        if ctx.needs_input_grad[1]:
            # dW = dY^T X contracts over the M rows: with ONE row group the wgrad kernel's grid is only (N / 256) x (K / 256)
            # tiles (27 for the qkv projection: a tenth of the chip, 700 us at ViT-B).  The rows are cut into S pseudo-groups
            # so that S x tiles fill the CUs once; the S partial [N, K] products are summed in group order (deterministic).
            S = _wgrad_splits(M, Np, K, x16.device)
            if S > 1:
                offs_s = ctx.cache.split_offsets(M, S, x16.device)
                dw = ops.grouped_wgrad_rows(dy16, x16, offs_s).sum(0)
            else:
                dw = ops.grouped_wgrad_rows(dy16, x16, offs)[0]
            dw = dw[:N].reshape(weight.shape).to(weight.dtype)
        return dx, dw, db, dres, None, None, None, None


def _wgrad_splits(M: int, N: int, K: int, device) -> int:
    """Row slices for a single-group weight gradient: enough (<= 16, >= 2,048 rows each) for one round of workgroups."""
    tiles = -(-N // 256) * -(-K // 256)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    if tiles <= 4:      # DeiT-Tiny's widths (3 tiles for qkv, 1 for the projection): more, shorter slices -- 36 workgroups are not a round
        return max(1, min(32, cus // max(tiles, 1), M // 512))
    return max(1, min(16, cus // max(tiles, 1), M // 2048))


def linear_supported(x16: torch.Tensor, weight: torch.Tensor) -> bool:
    """Forward and backward GEMMs take these shapes: K a multiple of 64 (the forward's contraction length and the dgrad's
    output width); any N >= 1 (the backward pads it to a multiple of 64, the forward to a multiple of 8)."""
    M, K = x16.shape
    N = weight.shape[0]
    return (x16.is_cuda and x16.dtype == torch.float16 and x16.is_contiguous() and M > 0 and K % 64 == 0 and N >= 1
            and weight.dim() >= 2 and weight.numel() == N * K)


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T scale) v on the fused qkv layout [B, N, 3, H, 64] (16-bit), forward and backward on the HIP kernels."""

    @staticmethod
    def forward(ctx, qkv, B, N, H, hd, scale):
        ctx.dims = (B, N, H, hd, scale)
        o, lse = ops.attention(qkv, B, N, H, hd, scale, want_lse=True)
        ctx.save_for_backward(qkv, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        B, N, H, hd, scale = ctx.dims
        dqkv = ops.attention_bwd(qkv, o, do.contiguous(), lse, B, N, H, hd, scale)
        return dqkv, None, None, None, None, None
