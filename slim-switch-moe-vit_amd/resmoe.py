"""The reference's MoE-block surface (models/resMoE.py), re-exported on the MI355X-native operator.

Same class names, constructor signatures, sub-module / buffer names and factory names as the
reference, so callers written against models/resMoE.py (main.py:520-530 create_model, main.py:623
``"moe_gate"`` / ``"dense_gate"`` param-name matching, main.py:812 ``isinstance(m, Gate)``) work
unchanged.  Deliberately NOT reproduced: the dead ``ResBlock`` (resMoE.py:88-123, wrong positional
argument order; SURVEY.md note A).
"""
from __future__ import annotations

import math
import typing as typ

import torch as th
import torch.nn as nn

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
    """models/resMoE.py:15-29 -- activation = Sequential(act_layer(), Dropout(drop)), naive gate."""

    def __init__(
        self,
        in_features: int,
        hidden_features: int,
        moe_num_experts: int,
        moe_top_k: int,
        drop: float,
        act_layer: typ.Callable = th.nn.GELU,
        **moe_kwargs,
    ):
        activation = nn.Sequential(*[act_layer(), nn.Dropout(p=drop)])
        # use naive-gate
        super().__init__(moe_num_experts, in_features, hidden_features, activation, top_k=moe_top_k, **moe_kwargs)


class Gate(nn.Module):
    """Token-skip gate, models/resMoE.py:32-85.  ``forward(x[B,N,d]) -> mask[B,N,2]`` (skip, keep).

    Difference from the reference that does not change results: the skipped-token counter is
    accumulated on the device and only synchronised when ``_skipped_tokens`` is read (the reference
    calls ``.item()`` inside forward, resMoE.py:83, which stalls the stream twice per block)."""

    def __init__(
        self,
        in_dim: int,
        tau: float,
        dropout: float = 0.0,
        target_threshold: float = 0.9,
        starting_threshold: float = 1.0,
        is_hard: float = True,
    ):
        super().__init__()
        self.head = nn.Sequential(nn.Dropout(p=dropout), nn.Linear(in_dim, 1))
        self.register_buffer("_threshold", th.tensor(starting_threshold))
        self.register_buffer("threshold", th.tensor(target_threshold))

        self._total_tokens = 0
        self._skipped_acc = None
        self._skipped_host = 0.0

        self.is_hard = is_hard
        self.disable = False

    @property
    def _skipped_tokens(self):
        if self._skipped_acc is not None:
            self._skipped_host += float(self._skipped_acc.item())
            self._skipped_acc = None
        return self._skipped_host

    @_skipped_tokens.setter
    def _skipped_tokens(self, v):
        self._skipped_acc = None
        self._skipped_host = float(v)

    def step(self, delta: th.Tensor):
        thresh = self._threshold - delta
        self._threshold.data.copy_(max(thresh, self.threshold))

    def forward(self, x):
        if self.disable:
            ret = th.zeros((x.size(0), x.size(1), 2), device=x.device)
            ret[:, :, 1] = 1
            return ret

        out = self.head(x)  # (B x Token x 1)

        threshold = self._threshold if self.training else self.threshold
        prob = th.sigmoid(out)
        _prob = 1 - prob

        if self.training and not self.is_hard:
            skip_tk = _prob
            tk = prob
        else:
            skip_tk = (prob > threshold).float() + _prob.detach() - _prob
            tk = (prob <= threshold).float() + prob.detach() - prob

        ret = th.cat([skip_tk, tk], dim=-1)

        self._total_tokens += math.prod(out.shape[0:2])
        s = skip_tk.sum().detach()
        self._skipped_acc = s if self._skipped_acc is None else self._skipped_acc + s
        return ret


def forward_residule_moe(self, x):
    """models/resMoE.py:126-145: residual is taken from the *normed* activations; token-skip gates
    zero the rows that bypass attention / the MoE."""
    x = self.norm1(x)

    mask = self.dense_gate(x)

    skip_tk = x * mask[:, :, 0].unsqueeze(dim=-1)
    tk = x * mask[:, :, 1].unsqueeze(dim=-1)

    x = self.drop_path(self.attn(tk)) + tk + skip_tk
    x = self.norm2(x)

    mask = self.moe_gate(x)

    skip_tk = x * mask[:, :, 0].unsqueeze(dim=-1)
    tk = x * mask[:, :, 1].unsqueeze(dim=-1)

    x = self.drop_path(self.mlp(tk)) + tk + skip_tk

    return x


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
