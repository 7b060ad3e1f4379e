"""Python face of the C-ABI (include/slimmoe.h): torch tensors in, torch tensors out.

Torch is plumbing here (device memory, streams); every function below launches hand-written HIP
kernels from libslimmoe_hip.so on torch's current stream and raises if that library is missing.
Shapes / dtypes / contiguity are validated here, before the C call (SURVEY.md 8b error convention).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._cache import StreamCache

F32, F16, BF16 = 0, 1, 2
GATE_NAIVE, GATE_SWITCH = 0, 1
EPI_NONE, EPI_GELU, EPI_GELU_GRAD = 0, 1, 2

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}
# grouped GEMM variant the modules use: 9 = the persistent kernel (one workgroup per CU walking tiles; auto tile height and
# schedule), bit-identical to 4 = one workgroup per tile (kept for A/B: SLIMMOE_GEMM_VARIANT=4)
import os as _os
DEFAULT_GEMM_VARIANT = int(_os.environ.get("SLIMMOE_GEMM_VARIANT", "9"))


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"unsupported dtype {dt}; expected float32/float16/bfloat16") from None


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the current stream's handle without building a Stream object


def _stream(t: torch.Tensor):
    """The HIP stream torch would launch on for ``t``'s device right now (every launch function takes it).  ~250 calls per forward:
    torch.cuda.current_stream() costs ~4.5 us each (a Python Stream object per call), the raw getter a tenth of that."""
    idx = t.device.index
    _lib.init_device(idx)  # first touch of a device: all launch attributes set before any launch
    if _raw_stream is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(t.device).cuda_stream


# ---- optional per-launch timing (bench.py's roofline leg): HIP events on the stream the kernels run on ----
_PROFILE = None  # list of (name, meta, start_event, end_event) while enabled


_PROFILE_ONLY = None  # optional set of launch names to time (None = all)


def profile_begin(only=None):
    """Start recording per-launch HIP events.  ``only``: restrict to these launch names (each timed launch costs the
    host two event records -- with the expert-parallel pipeline the host has no slack for timing every launch)."""
    global _PROFILE, _PROFILE_ONLY
    _PROFILE = []
    _PROFILE_ONLY = set(only) if only is not None else None


def profile_end():
    """-> list of (name, meta, milliseconds); call after a device synchronize."""
    global _PROFILE
    rec, _PROFILE = _PROFILE or [], None
    return [(n, m, s.elapsed_time(e)) for n, m, s, e in rec]


class _timed:
    def __init__(self, name, meta, ref):
        self.on = _PROFILE is not None and (_PROFILE_ONLY is None or name in _PROFILE_ONLY)
        if self.on:
            self.name, self.meta, self.ref = name, meta, ref

    def __enter__(self):
        if self.on:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record(torch.cuda.current_stream(self.ref.device))

    def __exit__(self, *a):
        if self.on:
            self.e.record(torch.cuda.current_stream(self.ref.device))
            _PROFILE.append((self.name, self.meta, self.s, self.e))


def _chk(t: torch.Tensor, name: str, dtype=None, ndim=None, align: int = 16):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: must live on the GPU (got {t.device}); the MoE hot path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError(f"{name}: expected {ndim} dims, got {tuple(t.shape)}")
    if t.data_ptr() % align != 0 and t.numel() > 0:
        raise RuntimeError(f"{name}: data pointer must be {align}-byte aligned")


_router_ws = {}   # (device index, stream) -> uint8 workspace whose counter words are zero between calls
WS_KEPT_ZERO = 0x200   # gate_kind flag (router entry points) / with_ln bit 1 (smoe_gate_ln_router): "the counter words are zero"


def _router_workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    """The router's redo workspace (counter words + token list), one per device and stream, allocated ZEROED once: every
    redo pass clears the counter words on its way out (csrc/router16_kernel.h), so no call has to launch a clearing kernel
    in front of the router -- the caller says so with WS_KEPT_ZERO.  Keyed by the stream: two compute streams never share
    counters; a captured graph replays on the tensor it captured."""
    key = (dev.index, _raw_stream(dev.index) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream)
    t = _router_ws.get(key)
    if t is None or t.numel() < nbytes:
        t = _router_ws[key] = torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev)
    return t


def router_topk(x: torch.Tensor, wg: torch.Tensor, bg: Optional[torch.Tensor], k: int, gate_kind: int = GATE_NAIVE,
                noise: Optional[torch.Tensor] = None, want_logits: bool = False, want_probs: bool = False,
                force_f64: bool = False):
    """(idx int64 [T,k], score f32 [T,k], logits f32 [T,E] | None, probs f32 [T,E] | None)."""
    _chk(x, "x", ndim=2)
    _chk(wg, "wg", torch.float32, 2)
    T, d = x.shape
    E = wg.shape[0]
    if wg.shape[1] != d:
        raise RuntimeError(f"wg: expected [E,{d}], got {tuple(wg.shape)}")
    if bg is not None:
        _chk(bg, "bg", torch.float32, 1)
        if bg.shape[0] != E:
            raise RuntimeError("bg: expected [E]")
    if noise is not None:
        _chk(noise, "noise", torch.float32, 2)
        if tuple(noise.shape) != (T, E):
            raise RuntimeError("noise: expected [T,E]")
    idx = torch.empty((T, k), dtype=torch.int64, device=x.device)
    score = torch.empty((T, k), dtype=torch.float32, device=x.device)
    logits = torch.empty((T, E), dtype=torch.float32, device=x.device) if want_logits else None
    probs = torch.empty((T, E), dtype=torch.float32, device=x.device) if want_probs else None
    lib = _lib.load()
    ws_bytes = lib.smoe_router_workspace_bytes(T)
    # (the all-f64 test mode runs its f32 pass without a redo pass behind it: it gets a scratch workspace and the clearing launch)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if force_f64 else _router_workspace(x.device, ws_bytes)
    with _timed("router", {"bytes": T * d * x.element_size()}, x):
        rc = lib.smoe_router_topk(_ptr(x), dtype_code(x.dtype), _ptr(wg), _ptr(bg), _ptr(noise), T, d, E, k,
                                  gate_kind | (0x100 if force_f64 else WS_KEPT_ZERO), _ptr(idx), _ptr(score), _ptr(logits),
                                  _ptr(probs), _ptr(ws), ws_bytes, _stream(x))
    _lib.check(rc, "smoe_router_topk")
    return idx, score, logits, probs


LN_DIMS = (192, 384, 768, 1024)


def layernorm(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], eps: float,
              out_dtype: torch.dtype) -> torch.Tensor:
    """LayerNorm over the last dim (d in LN_DIMS) with the output cast fused."""
    _chk(x, "x")
    d = x.shape[-1]
    T = x.numel() // d
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    with _timed("layernorm", {"bytes": T * d * (x.element_size() + out.element_size())}, x):
        rc = _lib.load().smoe_layernorm(_ptr(x), dtype_code(x.dtype), _ptr(weight), _ptr(bias), float(eps), T, d,
                                        _ptr(out), dtype_code(out_dtype), _stream(x))
    _lib.check(rc, "smoe_layernorm")
    return out


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, weight: Optional[torch.Tensor], eps: float,
                  dres: Optional[torch.Tensor] = None):
    """Backward of LayerNorm over the last dim from x alone: (dx f32 [same shape] (+ dres), dweight f32 [d], dbias f32 [d])."""
    _chk(x, "x", torch.float32)
    _chk(dy, "dy")
    d = x.shape[-1]
    T = x.numel() // d
    if dy.numel() != x.numel():
        raise RuntimeError("layernorm_bwd: dy must have x's shape")
    if dres is not None:
        _chk(dres, "dres", torch.float32)
    lib = _lib.load()
    dx = torch.empty_like(x)
    gb = torch.empty(2 * d, dtype=torch.float32, device=x.device)
    ws_bytes = lib.smoe_layernorm_bwd_workspace_bytes(T, d)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    with _timed("layernorm_bwd", {"bytes": T * d * (8 + dy.element_size())}, x):
        rc = lib.smoe_layernorm_bwd(_ptr(x), _ptr(dy), dtype_code(dy.dtype), _ptr(weight), _ptr(dres), float(eps), T, d,
                                    _ptr(dx), _ptr(gb), _ptr(ws), ws_bytes, _stream(x))
    _lib.check(rc, "smoe_layernorm_bwd")
    return dx, gb[:d], gb[d:]


def depth_scale_rows(mask: torch.Tensor, keep: float, N: int):
    """(factor f32 [B] = mask / keep, rows f32 [B * N] = the sample's factor on each of its N rows): stochastic depth's division and
    expansion in one launch (smoe_depth_scale_rows)."""
    _chk(mask, "mask", torch.float32, 1, align=4)
    B = mask.shape[0]
    factor = torch.empty(B, dtype=torch.float32, device=mask.device)
    rows = torch.empty(B * int(N), dtype=torch.float32, device=mask.device)
    rc = _lib.load().smoe_depth_scale_rows(_ptr(mask), float(keep), B, int(N), _ptr(factor), _ptr(rows), _stream(mask))
    _lib.check(rc, "smoe_depth_scale_rows")
    return factor, rows


def patchify_cast(images: torch.Tensor, ph: int, pw: int, out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """images f32 [B, C, H, W] -> patch rows [B * gh * gw, C * ph * pw] in 16 bit (smoe_patchify_cast)."""
    _chk(images, "images", torch.float32, 4)
    B, C, H, W = images.shape
    out = torch.empty((B * (H // ph) * (W // pw), C * ph * pw), dtype=out_dtype, device=images.device)
    rc = _lib.load().smoe_patchify_cast(_ptr(images), B, C, H, W, ph, pw, _ptr(out), dtype_code(out_dtype), _stream(images))
    _lib.check(rc, "smoe_patchify_cast")
    return out


def embed_ln(tokens: torch.Tensor, cls_token: torch.Tensor, pos_embed: torch.Tensor, B: int, P: int,
             ln: Optional[tuple] = None, xn_dtype: torch.dtype = torch.float16):
    """(x32 [B, P + 1, d], xn | None): the f32 stream cat(cls, tokens) + pos_embed and, with ``ln`` = (weight, bias, eps), its
    LayerNorm in 16 bit from the same pass (smoe_embed_ln)."""
    _chk(tokens, "tokens", ndim=2)
    d = tokens.shape[1]
    cls = cls_token.detach().reshape(-1)
    pos = pos_embed.detach().reshape(-1, d)
    _chk(cls, "cls_token", torch.float32, 1)
    _chk(pos, "pos_embed", torch.float32, 2)
    if cls.numel() != d or pos.shape[0] != P + 1 or tokens.shape[0] != B * P:
        raise RuntimeError("embed_ln: shapes disagree")
    x32 = torch.empty((B, P + 1, d), dtype=torch.float32, device=tokens.device)
    xn = torch.empty((B, P + 1, d), dtype=xn_dtype, device=tokens.device) if ln is not None else None
    lg, lb, eps = ln if ln is not None else (None, None, 0.0)
    with _timed("embed_ln", {"bytes": B * (P + 1) * d * (2 + 4 + (2 if ln is not None else 0))}, tokens):
        rc = _lib.load().smoe_embed_ln(_ptr(tokens), dtype_code(tokens.dtype), _ptr(cls), _ptr(pos), _ptr(lg), _ptr(lb), float(eps), B, P, d,
                                       _ptr(x32), _ptr(xn), dtype_code(xn_dtype), _stream(tokens))
    _lib.check(rc, "smoe_embed_ln")
    return x32, xn


def layernorm_rows(x: torch.Tensor, row_stride: int, T: int, d: int, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                   eps: float) -> torch.Tensor:
    """LayerNorm of the T rows x.data_ptr() + t * row_stride (f32, d elements each) -> f32 [T, d] (smoe_layernorm_rows)."""
    if not (x.is_cuda and x.dtype == torch.float32):
        raise RuntimeError("layernorm_rows: f32 GPU tensor expected")
    out = torch.empty((T, d), dtype=torch.float32, device=x.device)
    rc = _lib.load().smoe_layernorm_rows(_ptr(x), int(row_stride), _ptr(weight), _ptr(bias), float(eps), T, d, _ptr(out), _stream(x))
    _lib.check(rc, "smoe_layernorm_rows")
    return out


def attention_supported(N: int, head_dim: int) -> bool:
    return bool(_lib.load().smoe_attention_supported(N, head_dim))


def attention(qkv: torch.Tensor, B: int, N: int, H: int, head_dim: int, scale: float, want_lse: bool = False):
    """qkv [B*N, 3*H*head_dim] (or [B,N,3,H,hd]) 16-bit -> softmax(q k^T * scale) v as [B, N, H*head_dim]; with
    ``want_lse`` also the [B, H, N] f32 log2-normalisers the backward needs (returns (out, lse))."""
    _chk(qkv, "qkv")
    if qkv.dtype not in (torch.float16, torch.bfloat16) or qkv.numel() != B * N * 3 * H * head_dim:
        raise RuntimeError("qkv: expected a 16-bit [B,N,3,H,hd] tensor")
    out = torch.empty((B, N, H * head_dim), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device) if want_lse else None
    with _timed("attention", {"flops": 4.0 * B * H * N * N * head_dim}, qkv):
        rc = _lib.load().smoe_attention_fwd(_ptr(qkv), _ptr(out), dtype_code(qkv.dtype), B, N, H, head_dim, float(scale),
                                            _ptr(lse), _stream(qkv))
    _lib.check(rc, "smoe_attention_fwd")
    return (out, lse) if want_lse else out


def attention_bwd_supported(N: int, head_dim: int) -> bool:
    return bool(_lib.load().smoe_attention_bwd_supported(N, head_dim))


def attention_bwd(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, B: int, N: int, H: int,
                  head_dim: int, scale: float) -> torch.Tensor:
    """dqkv (qkv's shape and dtype) of softmax(q k^T scale) v from the forward's ``out`` and ``lse``."""
    _chk(qkv, "qkv")
    _chk(out, "out", qkv.dtype)
    _chk(dout, "dout", qkv.dtype)
    _chk(lse, "lse", torch.float32)
    if out.numel() != B * N * H * head_dim or dout.numel() != out.numel() or lse.numel() != B * H * N:
        raise RuntimeError("attention_bwd: shapes do not match [B,N,H*hd] / [B,H,N]")
    dqkv = torch.empty_like(qkv)
    with _timed("attention_bwd", {"flops": 10.0 * B * H * N * N * head_dim}, qkv):
        rc = _lib.load().smoe_attention_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), dtype_code(qkv.dtype), B, N,
                                            H, head_dim, float(scale), _stream(qkv))
    _lib.check(rc, "smoe_attention_bwd")
    return dqkv


def ln_router_supported(d: int, E: int, k: int) -> bool:
    return bool(_lib.load().smoe_ln_router_supported(d, E, k))


# SLIMMOE_ROUTER_HIST=1: the fused router pass also counts for the dispatch plan (count_by_gate folded in: chunk_hist +
# smoe_dispatch_plan_hist, one plan launch instead of two).  Bit-identical plans, and OFF by default: measured inside the model
# (two alternating bench runs on one box) the LayerNorm + router pass costs 72.3-73.1 us with it against 68.9-69.2 us without (it must
# walk contiguous 64-token chunks: 788 workgroups of 4 groups instead of 631 of 5) while the plan only drops from 15.1-15.4 to
# 14.8 us -- the counting launch was nearly free next to the assign launch, which now reads a 16 x longer table.
ROUTER_HIST = _os.environ.get("SLIMMOE_ROUTER_HIST", "0") == "1"


def chunk_hist(T: int, d: int, E: int, k: int, device) -> Optional[torch.Tensor]:
    """An (uninitialised) chunk-histogram table for the fused router passes to fill -- i32 [ceil(T / tok), E] with ``.tok`` tokens
    per row -- or None when this shape's router writes none (dispatch_plan then counts by itself)."""
    if not ROUTER_HIST or T <= 0:
        return None
    tok = _lib.load().smoe_router_chunk_hist_tokens(d, E, k)
    if tok <= 0:
        return None
    t = torch.empty((-(-T // tok), E), dtype=torch.int32, device=device)
    t.tok = tok
    return t


def ln_router_topk(x: torch.Tensor, ln_weight: Optional[torch.Tensor], ln_bias: Optional[torch.Tensor], eps: float,
                   wg: torch.Tensor, bg: Optional[torch.Tensor], k: int, gate_kind: int = GATE_NAIVE,
                   noise: Optional[torch.Tensor] = None, xn16_dtype: Optional[torch.dtype] = torch.float16,
                   want_xn32: bool = False, want_probs: bool = False, want_logits: bool = False, force_f64: bool = False,
                   hist: Optional[torch.Tensor] = None):
    """LayerNorm + router in one pass: (xn16 | None, xn32 | None, idx, score, logits | None, probs | None).  ``hist``
    (chunk_hist(T, d, E, k, device)): filled with the routing's per-chunk expert counts for dispatch_plan(hist=...)."""
    _chk(x, "x", ndim=2)
    _chk(wg, "wg", torch.float32, 2)
    T, d = x.shape
    E = wg.shape[0]
    for t, nm in ((ln_weight, "ln_weight"), (ln_bias, "ln_bias"), (bg, "bg")):
        if t is not None:
            _chk(t, nm, torch.float32, 1)
    dev = x.device
    xn16 = torch.empty((T, d), dtype=xn16_dtype, device=dev) if xn16_dtype is not None else None
    xn32 = torch.empty((T, d), dtype=torch.float32, device=dev) if want_xn32 else None
    idx = torch.empty((T, k), dtype=torch.int64, device=dev)
    score = torch.empty((T, k), dtype=torch.float32, device=dev)
    logits = torch.empty((T, E), dtype=torch.float32, device=dev) if want_logits else None
    probs = torch.empty((T, E), dtype=torch.float32, device=dev) if want_probs else None
    lib = _lib.load()
    ws_bytes = lib.smoe_router_workspace_bytes(T)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if force_f64 else _router_workspace(dev, ws_bytes)
    nbytes = T * d * (x.element_size() + (2 if xn16 is not None else 0) + (4 if want_xn32 else 0))
    with _timed("ln_router", {"bytes": nbytes}, x):
        rc = lib.smoe_ln_router_topk(_ptr(x), dtype_code(x.dtype), _ptr(ln_weight), _ptr(ln_bias), float(eps), _ptr(xn16),
                                     dtype_code(xn16_dtype) if xn16_dtype is not None else F16, _ptr(xn32), _ptr(wg),
                                     _ptr(bg), _ptr(noise), T, d, E, k, gate_kind | (0x100 if force_f64 else WS_KEPT_ZERO),
                                     _ptr(idx), _ptr(score), _ptr(logits), _ptr(probs), _ptr(None if force_f64 else hist), _ptr(ws),
                                     ws_bytes, _stream(x))
    _lib.check(rc, "smoe_ln_router_topk")
    return xn16, xn32, idx, score, logits, probs


def gate_ln_router_supported(d: int, E: int, k: int) -> bool:
    return bool(_lib.load().smoe_gate_ln_router_supported(d, E, k))


def gate_ln_router(x: torch.Tensor, gate_w: torch.Tensor, gate_b: Optional[torch.Tensor], threshold: Optional[torch.Tensor],
                   ln: Optional[tuple] = None, wg: Optional[torch.Tensor] = None, bg: Optional[torch.Tensor] = None,
                   k: int = 1, xn16_dtype: Optional[torch.dtype] = None, want_xn32: bool = False,
                   zero_out: Optional[torch.Tensor] = None, want_mask: bool = False,
                   skip_count: Optional[torch.Tensor] = None, xn32_out: Optional[torch.Tensor] = None,
                   hist: Optional[torch.Tensor] = None, want_tk32: bool = False):
    """[LayerNorm +] token-skip gate (+ NaiveGate router) in one pass (smoe_gate_ln_router).  ``ln`` = (weight, bias,
    eps) or None; ``threshold`` = the gate's 0-dim DEVICE buffer (None = gate disabled); ``wg`` None = no router.
    Returns a dict: xn16, xn32, idx, idx_plan, score, mask, tk32 (entries that were not asked for are None); ``tk32`` = the f32 row
    with zeros for the skipped tokens (``want_tk32``)."""
    _chk(x, "x", torch.float32, 2)
    T, d = x.shape
    dev = x.device
    gw = gate_w.detach().reshape(-1)
    _chk(gw, "gate_w", torch.float32, 1, align=4)
    if gw.numel() != d:
        raise RuntimeError(f"gate_w: expected {d} elements")
    gb = gate_b.detach().reshape(-1) if gate_b is not None else None
    thr = threshold.detach().reshape(-1) if threshold is not None else None
    for t, nm in ((gb, "gate_b"), (thr, "threshold")):
        if t is not None:
            _chk(t, nm, torch.float32, 1, align=4)
    E = 0
    if wg is not None:
        _chk(wg, "wg", torch.float32, 2)
        E = wg.shape[0]
        if bg is not None:
            _chk(bg, "bg", torch.float32, 1, align=4)
    lg = lb = None
    eps = 0.0
    if ln is not None:
        lg, lb, eps = ln
        for t, nm in ((lg, "ln_weight"), (lb, "ln_bias")):
            if t is not None:
                _chk(t, nm, torch.float32, 1)
    if zero_out is not None:
        _chk(zero_out, "zero_out", torch.float32, 1)
    xn16 = torch.empty((T, d), dtype=xn16_dtype, device=dev) if xn16_dtype is not None else None
    xn32 = xn32_out if xn32_out is not None else (torch.empty((T, d), dtype=torch.float32, device=dev) if want_xn32 else None)
    if xn32 is not None:
        _chk(xn32, "xn32", torch.float32, 2)
    idx = torch.empty((T, k), dtype=torch.int64, device=dev) if E else None
    idx_plan = torch.empty((T, k), dtype=torch.int64, device=dev) if E else None
    score = torch.empty((T, k), dtype=torch.float32, device=dev) if E else None
    mask = torch.empty((T, 2), dtype=torch.float32, device=dev) if want_mask else None
    tk32 = torch.empty((T, d), dtype=torch.float32, device=dev) if want_tk32 else None
    if skip_count is not None:
        _chk(skip_count, "skip_count", torch.int32, align=4)
    lib = _lib.load()
    ws_bytes = lib.smoe_router_workspace_bytes(T)
    ws = _router_workspace(dev, ws_bytes)
    nbytes = T * d * (4 + (2 if xn16 is not None else 0) + (4 if xn32 is not None else 0) + (4 if tk32 is not None else 0))
    with _timed("gate_ln_router" if E else "gate_ln", {"bytes": nbytes}, x):
        rc = lib.smoe_gate_ln_router(_ptr(x), F32, (1 if ln is not None else 0) | 2, _ptr(lg), _ptr(lb), float(eps), _ptr(gw),
                                     _ptr(gb), _ptr(thr), _ptr(xn16),
                                     dtype_code(xn16_dtype) if xn16_dtype is not None else F16, _ptr(xn32),
                                     _ptr(zero_out), _ptr(wg), _ptr(bg), T, d, E, k, _ptr(idx), _ptr(idx_plan),
                                     _ptr(score), _ptr(mask), _ptr(skip_count), _ptr(hist if E else None), _ptr(tk32), _ptr(ws),
                                     ws_bytes, _stream(x))
    _lib.check(rc, "smoe_gate_ln_router")
    return {"xn16": xn16, "xn32": xn32, "idx": idx, "idx_plan": idx_plan, "score": score, "mask": mask, "tk32": tk32}


def gate_ln_bwd(x: torch.Tensor, g_f: torch.Tensor, g_out: Optional[torch.Tensor], ln_w: Optional[torch.Tensor],
                ln_b: Optional[torch.Tensor], eps: float, gate_w: torch.Tensor, gate_b: Optional[torch.Tensor],
                mask: Optional[torch.Tensor], gate_on: bool = True, want_dz: bool = False):
    """Backward of one gated half's front end (LayerNorm -> token-skip gate) in one pass over its input x (smoe_gate_ln_bwd):
    (dx f32 [T,d], dln_w [d], dln_b [d], dgate_w [d], dgate_b [1], dz [T] | None)."""
    _chk(x, "x", torch.float32, 2)
    _chk(g_f, "g_f", ndim=2)
    T, d = x.shape
    if tuple(g_f.shape) != (T, d):
        raise RuntimeError("gate_ln_bwd: g_f must have x's shape")
    if g_out is not None:
        _chk(g_out, "g_out", torch.float32, 2)
    gw = gate_w.detach().reshape(-1)
    _chk(gw, "gate_w", torch.float32, 1, align=4)
    gb = gate_b.detach().reshape(-1) if gate_b is not None else None
    if gb is not None:
        _chk(gb, "gate_b", torch.float32, 1, align=4)
    if mask is not None:
        _chk(mask, "mask", torch.float32, 2, align=4)
    for nm, t in (("ln_w", ln_w), ("ln_b", ln_b)):
        if t is not None:
            _chk(t, nm, torch.float32, 1)
    lib = _lib.load()
    dx = torch.empty_like(x)
    out = torch.empty(3 * d + 4, dtype=torch.float32, device=x.device)
    dz = torch.empty(T, dtype=torch.float32, device=x.device) if want_dz else None
    ws_bytes = lib.smoe_gate_ln_bwd_workspace_bytes(T, d)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    rc = lib.smoe_gate_ln_bwd(_ptr(x), _ptr(g_f), dtype_code(g_f.dtype), _ptr(g_out), _ptr(ln_w), _ptr(ln_b), float(eps), _ptr(gw),
                              _ptr(gb), _ptr(mask), 1 if gate_on else 0, T, d, _ptr(dx), _ptr(dz), _ptr(out), _ptr(ws), ws_bytes,
                              _stream(x))
    _lib.check(rc, "smoe_gate_ln_bwd")
    return dx, out[:d], out[d:2 * d], out[2 * d:3 * d], out[3 * d:3 * d + 1], dz


def skip_gate_bwd(xn: torch.Tensor, g_f: torch.Tensor, g_out: Optional[torch.Tensor], gate_w: torch.Tensor,
                  gate_b: Optional[torch.Tensor], mask: Optional[torch.Tensor], gate_on: bool = True):
    """Backward of one gated half of the residual-MoE block in training (smoe_skip_gate_bwd): (dxn f32 [T,d], dz f32 [T])."""
    _chk(xn, "xn", torch.float32, 2)
    _chk(g_f, "g_f", ndim=2)
    T, d = xn.shape
    if tuple(g_f.shape) != (T, d):
        raise RuntimeError("skip_gate_bwd: g_f must have xn's shape")
    if g_out is not None:
        _chk(g_out, "g_out", torch.float32, 2)
    gw = gate_w.detach().reshape(-1)
    _chk(gw, "gate_w", torch.float32, 1, align=4)
    gb = gate_b.detach().reshape(-1) if gate_b is not None else None
    if mask is not None:
        _chk(mask, "mask", torch.float32, 2, align=4)
    dxn = torch.empty_like(xn)
    dz = torch.empty(T, dtype=torch.float32, device=xn.device)
    rc = _lib.load().smoe_skip_gate_bwd(_ptr(xn), _ptr(g_f), dtype_code(g_f.dtype), _ptr(g_out), _ptr(gw), _ptr(gb), _ptr(mask),
                                        1 if gate_on else 0, T, d, _ptr(dxn), _ptr(dz), _stream(xn))
    _lib.check(rc, "smoe_skip_gate_bwd")
    return dxn, dz


def zero_row_output(bg: Optional[torch.Tensor], k: int, w2: torch.Tensor, b1: Optional[torch.Tensor],
                    b2: Optional[torch.Tensor], e_base: int = 0, E_total: Optional[int] = None) -> torch.Tensor:
    """out[d] = sum_j score_j (W2[e_j] gelu(b1[e_j]) + b2[e_j]) for the NaiveGate routing of an all-zero row.  Expert parallel
    (``E_total`` global experts in ``bg``, ``w2`` / ``b1`` / ``b2`` = this rank's experts [e_base, e_base + E_local)): this rank's
    partial sum -- the ranks' results add up to the row."""
    _chk(w2, "w2", torch.float32, 3)
    E_local, d, h = w2.shape
    E = E_local if E_total is None else int(E_total)
    for t, nm in ((bg, "bg"), (b1, "b1"), (b2, "b2")):
        if t is not None:
            _chk(t, nm, torch.float32, align=4)
    if bg is not None and bg.numel() != E:
        raise RuntimeError(f"zero_row_output: bg has {bg.numel()} entries, expected {E}")
    out = torch.empty(d, dtype=torch.float32, device=w2.device)
    rc = _lib.load().smoe_zero_row_output(_ptr(bg), E, k, _ptr(w2), _ptr(b1), _ptr(b2), d, h, int(e_base), E_local, _ptr(out),
                                          _stream(w2))
    _lib.check(rc, "smoe_zero_row_output")
    return out


def dispatch_plan(idx: torch.Tensor, E: int, capacity: int = -1, want_pruned: Optional[bool] = None,
                  hist: Optional[torch.Tensor] = None):
    """(counts i32 [E], offsets i32 [E+1], pos i64 [n], inv_pos i64 [n], idx_pruned i64 [n] | None).  ``hist``: the chunk histogram the
    fused router filled for exactly this ``idx`` (chunk_hist / ln_router_topk / gate_ln_router): the plan then skips its counting
    launch (smoe_dispatch_plan_hist)."""
    _chk(idx, "idx", torch.int64, align=8)  # read element-wise: slices of a [T,k] tensor are fine
    flat = idx.reshape(-1)
    n = flat.numel()
    dev = idx.device
    if want_pruned is None:
        want_pruned = capacity >= 0
    lib = _lib.load()
    ws_bytes = lib.smoe_dispatch_plan_workspace_bytes(n, E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    counts = torch.empty(E, dtype=torch.int32, device=dev)
    offsets = torch.empty(E + 1, dtype=torch.int32, device=dev)
    pos = torch.empty(n, dtype=torch.int64, device=dev)
    inv_pos = torch.empty(n, dtype=torch.int64, device=dev)
    pruned = torch.empty(n, dtype=torch.int64, device=dev) if want_pruned else None
    with _timed("plan", {"bytes": n * 24}, idx):
        if hist is not None:
            _chk(hist, "hist", torch.int32, 2)
            k = idx.shape[1] if idx.dim() == 2 else 1      # flat entries per token
            rc = lib.smoe_dispatch_plan_hist(_ptr(flat), n, E, int(capacity), _ptr(hist), int(hist.tok) * k, _ptr(counts), _ptr(offsets),
                                             _ptr(pos), _ptr(inv_pos), _ptr(pruned), _ptr(ws), ws_bytes, _stream(idx))
        else:
            rc = lib.smoe_dispatch_plan(_ptr(flat), n, E, int(capacity), _ptr(counts), _ptr(offsets), _ptr(pos),
                                        _ptr(inv_pos), _ptr(pruned), _ptr(ws), ws_bytes, _stream(idx))
    _lib.check(rc, "smoe_dispatch_plan")
    return counts, offsets, pos, inv_pos, pruned


def dispatch_plan_padded(idx: torch.Tensor, E: int, capacity: int, slot_rows: Optional[int] = None, want_raw: bool = False):
    """The plan in the padded layout of a capacity gate's static exchange: expert e owns the slots [e * slot_rows, (e + 1) *
    slot_rows) (slot_rows >= capacity, default = capacity).  Returns (counts i32 [E], offsets i32 [E+1] (compact prefix),
    group_end i32 [E] = e * slot_rows + counts[e], pos_padded i64 [E * slot_rows] (-1 = unused slot), inv_pos i64 [n] (padded
    slot or -1), idx_pruned i64 [n]); with ``want_raw`` a seventh entry, raw_counts i32 [E] = the entries routed to each expert
    before the capacity clamp."""
    slot_rows = int(capacity) if slot_rows is None else int(slot_rows)
    _chk(idx, "idx", torch.int64, align=8)
    flat = idx.reshape(-1)
    n = flat.numel()
    dev = idx.device
    lib = _lib.load()
    ws_bytes = lib.smoe_dispatch_plan_workspace_bytes(n, E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    counts = torch.empty(E, dtype=torch.int32, device=dev)
    offsets = torch.empty(E + 1, dtype=torch.int32, device=dev)
    group_end = torch.empty(E, dtype=torch.int32, device=dev)
    pos = torch.empty(E * slot_rows, dtype=torch.int64, device=dev)
    inv_pos = torch.empty(n, dtype=torch.int64, device=dev)
    pruned = torch.empty(n, dtype=torch.int64, device=dev)
    raw = torch.empty(E, dtype=torch.int32, device=dev) if want_raw else None
    with _timed("plan", {"bytes": n * 24}, idx):
        rc = lib.smoe_dispatch_plan_padded(_ptr(flat), n, E, int(capacity), slot_rows, _ptr(counts), _ptr(offsets), _ptr(group_end),
                                           _ptr(pos), _ptr(inv_pos), _ptr(pruned), _ptr(raw), _ptr(ws), ws_bytes, _stream(idx))
    _lib.check(rc, "smoe_dispatch_plan_padded")
    if want_raw:
        return counts, offsets, group_end, pos, inv_pos, pruned, raw
    return counts, offsets, group_end, pos, inv_pos, pruned


def dispatch_plan_slots(idx: torch.Tensor, E: int, slot_base: torch.Tensor, n_slots: int, capacity: int = -1, hdr_rows: int = 1):
    """The plan over a slot table (smoe_dispatch_plan_slots): expert e owns the slots [slot_base[e], slot_base[e + 1]) (i32 [E + 1] on
    the device; ``n_slots`` = slot_base[E], known to the host), the last ``hdr_rows`` of them reserved.  Returns (counts i32 [E],
    offsets i32 [E+1], group_end i32 [E], pos_slots i64 [n_slots] (-1 = unused), inv_pos i64 [n], idx_pruned i64 [n],
    raw_counts i32 [E] = the entries routed to each expert before any clamp)."""
    _chk(idx, "idx", torch.int64, align=8)
    _chk(slot_base, "slot_base", torch.int32, 1, align=4)
    if slot_base.numel() != E + 1:
        raise RuntimeError("slot_base: expected E + 1 entries")
    flat = idx.reshape(-1)
    n = flat.numel()
    dev = idx.device
    lib = _lib.load()
    ws_bytes = lib.smoe_dispatch_plan_workspace_bytes(n, E)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    counts = torch.empty(E, dtype=torch.int32, device=dev)
    offsets = torch.empty(E + 1, dtype=torch.int32, device=dev)
    group_end = torch.empty(E, dtype=torch.int32, device=dev)
    pos = torch.empty(int(n_slots), dtype=torch.int64, device=dev)
    inv_pos = torch.empty(n, dtype=torch.int64, device=dev)
    pruned = torch.empty(n, dtype=torch.int64, device=dev)
    raw = torch.empty(E, dtype=torch.int32, device=dev)
    with _timed("plan", {"bytes": n * 24}, idx):
        rc = lib.smoe_dispatch_plan_slots(_ptr(flat), n, E, int(capacity), _ptr(slot_base), int(hdr_rows), _ptr(counts), _ptr(offsets),
                                          _ptr(group_end), _ptr(pos), _ptr(inv_pos), _ptr(pruned), _ptr(raw), _ptr(ws), ws_bytes,
                                          _stream(idx))
    _lib.check(rc, "smoe_dispatch_plan_slots")
    return counts, offsets, group_end, pos, inv_pos, pruned, raw


def ep_pack_headers(send: torch.Tensor, counts: Optional[torch.Tensor], raw_counts: Optional[torch.Tensor], slot_base: torch.Tensor,
                    t_rows: int) -> None:
    """Write the header rows of a static-exchange send buffer (smoe_ep_pack_headers): region g's last row gets int32
    {counts[g], raw_counts[g], t_rows, G, raw_counts[0 .. G)}."""
    _chk(send, "send", ndim=2)
    _chk(slot_base, "slot_base", torch.int32, 1, align=4)
    G = slot_base.numel() - 1
    for t, nm in ((counts, "counts"), (raw_counts, "raw_counts")):
        if t is not None:
            _chk(t, nm, torch.int32, 1, align=4)
            if t.numel() != G:
                raise RuntimeError(f"{nm}: expected {G} entries")
    rc = _lib.load().smoe_ep_pack_headers(_ptr(counts), _ptr(raw_counts), _ptr(slot_base), G, send.shape[1] * send.element_size(),
                                          int(t_rows), _ptr(send), _stream(send))
    _lib.check(rc, "smoe_ep_pack_headers")


def ep_unpack_headers(recv: torch.Tensor, W: int, local_base: torch.Tensor, E_total: int):
    """(starts i32 [G], ends i32 [G], stats i32 [W, 1 + E_total]) from the header rows of a received static-exchange buffer
    (smoe_ep_unpack_headers): group l = (source rank, local expert) holds the rows [starts[l], ends[l]); stats[w] = (source w's
    row count, its pre-clamp counts of all E_total experts)."""
    _chk(recv, "recv", ndim=2)
    _chk(local_base, "local_base", torch.int32, 1, align=4)
    E_local = local_base.numel() - 1
    G = int(W) * E_local
    dev = recv.device
    starts = torch.empty(G, dtype=torch.int32, device=dev)
    ends = torch.empty(G, dtype=torch.int32, device=dev)
    stats = torch.empty((int(W), 1 + int(E_total)), dtype=torch.int32, device=dev)
    rc = _lib.load().smoe_ep_unpack_headers(_ptr(recv), int(W), E_local, _ptr(local_base), recv.shape[1] * recv.element_size(),
                                            int(E_total), _ptr(starts), _ptr(ends), _ptr(stats), _stream(recv))
    _lib.check(rc, "smoe_ep_unpack_headers")
    return starts, ends, stats


def scatter_rows(x: torch.Tensor, pos: torch.Tensor, k: int, out_dtype: torch.dtype,
                 out: Optional[torch.Tensor] = None, zero_fill: bool = False,
                 scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """buf[s] = cast(x[pos[s] // k]) (* scale[pos[s]]) for every slot with pos[s] >= 0 (MOEScatter local part;
    with ``scale`` = the adjoint of the combine); ``zero_fill``: the other slots become zero rows (also of a given ``out``)."""
    _chk(x, "x", ndim=2)
    _chk(pos, "pos", torch.int64, 1, align=8)
    n_slots = pos.numel()
    d = x.shape[1]
    if out is None:
        out = torch.empty((n_slots, d), dtype=out_dtype, device=x.device)
    else:
        _chk(out, "out", out_dtype, 2)
    if scale is not None:
        _chk(scale, "scale", torch.float32, align=4)
        if scale.numel() != x.shape[0] * k:
            raise RuntimeError("scale: expected T*k entries")
    lib = _lib.load()
    # zero_fill: slots no token maps to (pos < 0) are written as zero rows by the same pass, not by a memset of the whole buffer
    fn = lib.smoe_scatter_rows_fill if zero_fill else lib.smoe_scatter_rows
    with _timed("scatter", {"bytes": n_slots * d * (x.element_size() + out.element_size())}, x):
        rc = fn(_ptr(x), dtype_code(x.dtype), _ptr(pos), _ptr(scale), n_slots, k, d, _ptr(out), dtype_code(out.dtype), _stream(x))
    _lib.check(rc, "smoe_scatter_rows")
    return out


def gather_combine(y: torch.Tensor, inv_pos: torch.Tensor, score: torch.Tensor, T: int, k: int,
                   out_dtype: torch.dtype, residual: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[t] = sum_j score[t,j] * y[inv_pos[t*k+j]] (+ residual[t]); dropped entries contribute 0."""
    _chk(y, "y", ndim=2)
    _chk(inv_pos, "inv_pos", torch.int64, align=8)
    _chk(score, "score", torch.float32, align=4)
    d = y.shape[1]
    if inv_pos.numel() != T * k or score.numel() != T * k:
        raise RuntimeError("inv_pos / score: expected T*k entries")
    if out is None:
        out = torch.empty((T, d), dtype=out_dtype, device=y.device)
    else:
        _chk(out, "out", out_dtype, 2)
        if tuple(out.shape) != (T, d):
            raise RuntimeError("out: expected [T,d]")
    if residual is not None:
        _chk(residual, "residual", out_dtype, 2)
    lib = _lib.load()
    with _timed("combine", {"bytes": T * d * (k * y.element_size() + out.element_size())}, y):
        rc = lib.smoe_gather_combine(_ptr(y), dtype_code(y.dtype), _ptr(inv_pos), _ptr(score), T, k, d, _ptr(residual),
                                     _ptr(out), dtype_code(out_dtype), _stream(y))
    _lib.check(rc, "smoe_gather_combine")
    return out


def gather_combine_ln(y: torch.Tensor, inv_pos: torch.Tensor, score: torch.Tensor, T: int, k: int,
                      residual: Optional[torch.Tensor], ln_weight: torch.Tensor, ln_bias: torch.Tensor, eps: float,
                      xn_dtype: torch.dtype = torch.float16, out: Optional[torch.Tensor] = None,
                      xn: Optional[torch.Tensor] = None):
    """(out f32 [T,d], xn [T,d] 16-bit): out[t] = residual[t] + sum_j score[t,j] y[inv_pos[t k + j]]; xn = LayerNorm(out).
    ``out`` / ``xn``: write into these (row slices of a layer's buffers: one call per token chunk)."""
    _chk(y, "y", ndim=2)
    _chk(inv_pos, "inv_pos", torch.int64, align=8)
    _chk(score, "score", torch.float32, align=4)
    _chk(ln_weight, "ln_weight", torch.float32, 1)
    _chk(ln_bias, "ln_bias", torch.float32, 1)
    d = y.shape[1]
    if residual is not None:
        _chk(residual, "residual", torch.float32, 2)
    if out is None:
        out = torch.empty((T, d), dtype=torch.float32, device=y.device)
    else:
        _chk(out, "out", torch.float32, 2)
    if xn is None:
        xn = torch.empty((T, d), dtype=xn_dtype, device=y.device)
    else:
        _chk(xn, "xn", xn_dtype, 2)
    if tuple(out.shape) != (T, d) or tuple(xn.shape) != (T, d):
        raise RuntimeError("gather_combine_ln: out / xn must be [T, d]")
    with _timed("combine_ln", {"bytes": T * d * (k * y.element_size() + 8 + 2)}, y):
        rc = _lib.load().smoe_gather_combine_ln(_ptr(y), dtype_code(y.dtype), _ptr(inv_pos), _ptr(score), T, k, d, _ptr(residual),
                                                _ptr(out), _ptr(ln_weight), _ptr(ln_bias), float(eps), _ptr(xn), dtype_code(xn_dtype),
                                                _stream(y))
    _lib.check(rc, "smoe_gather_combine_ln")
    return out, xn


_reserved_cus = 0


def set_reserved_cus(n: int) -> int:
    """The persistent grouped GEMM leaves ``n`` CUs free from now on (smoe_set_reserved_cus; process-wide): kernels on other streams
    -- RCCL's all-to-all under the expert-parallel micro-batch pipeline -- can then run BESIDE the GEMMs instead of behind them.
    Returns the previous value.  Results do not depend on it."""
    global _reserved_cus
    _lib.check(_lib.load().smoe_set_reserved_cus(int(n)), "smoe_set_reserved_cus")
    prev, _reserved_cus = _reserved_cus, int(n)
    return prev


def _ps_variant(rows: int, G: int, K: int, N: int, device) -> int:
    """The persistent GEMM's tile height / schedule for ``rows`` real rows in G groups -- csrc/gemm.hip's variant-9 rule (expected
    tiles of 256 or 320 rows, cost-weighted rounds of workgroups, ties to the taller tile; deep schedule for K >= 2048) as an
    explicit variant: 10 / 11 = 320- / 256-row tiles, 13 / 12 = the same on the deep schedule."""
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    ntn = -(-N // 256)
    t256 = (-(-rows // 256) + G // 2) * ntn
    t320 = (-(-rows // 320) + G // 2) * ntn
    tall = -(-t320 // cus) * 1.25 <= -(-t256 // cus) * 1.0
    if K >= 2048:
        return 13 if tall else 12
    return 10 if tall else 11


def grouped_gemm(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], offsets: torch.Tensor,
                 epilogue: int = EPI_NONE, out_dtype: Optional[torch.dtype] = None,
                 row_map: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None, variant: int = 0,
                 group_expert: Optional[torch.Tensor] = None, rows_hint: Optional[int] = None,
                 residual: Optional[torch.Tensor] = None, a_gather: Optional[torch.Tensor] = None,
                 a_div: int = 1, prof_name: str = "grouped_gemm", group_end: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = epi(A[r] @ W[e]^T + bias[e]) for r in [offsets[g], offsets[g+1]), e = group_expert[g] (or g).
    W [E,N,K].  ``rows_hint`` = number of rows actually routed (for the profiler's FLOP count only).
    ``group_end`` (i32 [G]): separate row ranges [offsets[g], group_end[g]) -- ``offsets`` then has G entries (the padded
    buffers of a capacity gate's static exchange; persistent kernel only)."""
    _chk(A, "A", ndim=2)
    _chk(W, "W", A.dtype, 3)
    _chk(offsets, "offsets", torch.int32, 1)
    M, K = A.shape
    if a_gather is not None:  # A is the un-permuted token matrix; the GEMM has one row per routed slot
        _chk(a_gather, "a_gather", torch.int64, 1, align=8)
        M = a_gather.numel()
    E, N, K2 = W.shape
    if K2 != K:
        raise RuntimeError(f"W: expected [E,N,{K}], got {tuple(W.shape)}")
    G = offsets.numel() - 1
    if group_end is not None:
        _chk(group_end, "group_end", torch.int32, 1)
        G = offsets.numel()
        if group_end.numel() != G:
            raise RuntimeError("group_end: expected one entry per row group (offsets then holds the G starts)")
    if group_expert is None:
        if G != E:
            raise RuntimeError("offsets: expected E+1 entries")
    else:
        _chk(group_expert, "group_expert", torch.int32, 1)
        if group_expert.numel() != G:
            raise RuntimeError("group_expert: expected one entry per row group")
    if bias is not None:
        _chk(bias, "bias", torch.float32, 2)
        if tuple(bias.shape) != (E, N):
            raise RuntimeError("bias: expected [E,N]")
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else A.dtype
    if row_map is not None:
        _chk(row_map, "row_map", torch.int64, 1, align=8)
        if out is None:
            raise RuntimeError("row_map needs a caller-provided `out`")
        if row_scale is not None:
            _chk(row_scale, "row_scale", torch.float32, align=4)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    else:
        _chk(out, "out", out_dtype, 2)
        if out.shape[1] != N:
            raise RuntimeError("out: expected N columns")
    if residual is not None:
        _chk(residual, "residual", out_dtype, 2)
        if tuple(residual.shape) != tuple(out.shape):
            raise RuntimeError("residual: expected the shape of out")
    lib = _lib.load()
    rows = M if rows_hint is None else rows_hint
    if variant == 9 and group_end is not None and rows_hint is not None:
        # separate row ranges inside a PADDED buffer: the library picks tile height from the rows it is told about (M = the buffer);
        # pick here, by the same cost-weighted rounds, from the rows that are expected to exist
        variant = _ps_variant(int(rows_hint), G, K, N, A.device)
    with _timed(prof_name, {"flops": 2.0 * rows * K * N, "K": K, "N": N, "epilogue": epilogue}, A):
        rc = lib.smoe_grouped_gemm(_ptr(A), _ptr(W), _ptr(bias), _ptr(offsets), _ptr(group_expert), G, E, M, K, N,
                                   dtype_code(A.dtype), epilogue, _ptr(row_map), _ptr(row_scale), _ptr(residual),
                                   _ptr(a_gather), a_div, _ptr(out), out.shape[0], dtype_code(out_dtype), variant,
                                   _ptr(group_end), _stream(A))
    _lib.check(rc, "smoe_grouped_gemm")
    return out


# SLIMMOE_FFN_FUSED=1: one persistent launch for both expert GEMMs of a layer (smoe_expert_ffn).  Bit-identical to the two launches
# and OFF by default: measured 0.530-0.545 ms against 0.50-0.51 ms for the two launches at ViT-B / 256 images -- what the shared
# tile list saves in partly filled rounds (~30 us) goes into the switches between the two GEMM bodies, the write-through stores
# of H and the waits on late m-tiles (profiles/r04_fused_ffn.md).
# Since round 5 the launch is not in the default build: `make -C slim-switch-moe-vit_amd/csrc FFN=-DSMOE_FFN_FUSED` compiles it back in.
FFN_FUSED = _os.environ.get("SLIMMOE_FFN_FUSED", "0") == "1"


def ffn_fused_available() -> bool:
    return _lib.has_symbol("smoe_expert_ffn")
_ffn_ws = {}   # (device index, stream) -> the fused launch's kept-zero workspace


def _ffn_workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    """Ticket / row-counter words of smoe_expert_ffn: allocated ZEROED once per device and stream; every launch leaves them zero
    (the last workgroup to leave clears them), so no clearing launch runs in front of the GEMMs.  Keyed by the stream: launches on
    two streams may overlap and must not share counters."""
    key = (dev.index, _raw_stream(dev.index) if _raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream)
    t = _ffn_ws.get(key)
    if t is None or t.numel() < nbytes:
        t = _ffn_ws[key] = torch.zeros(max(nbytes, 1 << 14), dtype=torch.uint8, device=dev)
    return t


def ffn_workspace_error(dev: torch.device) -> bool:
    """True if any fused FFN launch on this device ever ran out of its wait on a row counter (reads the error words: a sync)."""
    return any(bool(t.view(torch.int32)[17].item()) for (d, _), t in _ffn_ws.items() if d == dev.index)   # FUSED_WS_ERR


def expert_ffn(X: torch.Tensor, W1: torch.Tensor, b1: Optional[torch.Tensor], W2: torch.Tensor, b2: Optional[torch.Tensor],
               offsets: torch.Tensor, out: torch.Tensor, a_gather: Optional[torch.Tensor] = None, a_div: int = 1,
               row_map: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, group_expert: Optional[torch.Tensor] = None,
               H: Optional[torch.Tensor] = None, rows_hint: Optional[int] = None) -> Optional[torch.Tensor]:
    """``out[row_map[r]] = residual[row_map[r]] + row_scale[row_map[r]] * (gelu(X[a_gather[r] / a_div] W1[e]^T + b1[e]) W2[e]^T + b2[e])``
    -- both expert GEMMs of a MoE layer with the top-1 combine -- in ONE persistent launch (smoe_expert_ffn); bit-identical to
    grouped_gemm(EPI_GELU) followed by grouped_gemm(row_map, row_scale, residual).  Returns ``out``, or None when the shape is
    outside the fused launch's reach (the caller then issues the two launches)."""
    _chk(X, "X", ndim=2)
    _chk(W1, "W1", X.dtype, 3)
    _chk(W2, "W2", X.dtype, 3)
    _chk(offsets, "offsets", torch.int32, 1)
    _chk(out, "out", torch.float32, 2)
    E, h, d_in = W1.shape
    E2, d_out, h2 = W2.shape
    if X.shape[1] != d_in or h2 != h or E2 != E or out.shape[1] != d_out:
        raise RuntimeError(f"expert_ffn: shapes disagree: X {tuple(X.shape)}, W1 {tuple(W1.shape)}, W2 {tuple(W2.shape)}, out {tuple(out.shape)}")
    M = X.shape[0]
    if a_gather is not None:
        _chk(a_gather, "a_gather", torch.int64, 1, align=8)
        M = a_gather.numel()
    G = offsets.numel() - 1
    if group_expert is None:
        if G != E:
            raise RuntimeError("offsets: expected E+1 entries")
    else:
        _chk(group_expert, "group_expert", torch.int32, 1)
    for t, nm, n in ((b1, "b1", h), (b2, "b2", d_out)):
        if t is not None:
            _chk(t, nm, torch.float32, 2)
            if tuple(t.shape) != (E, n):
                raise RuntimeError(f"{nm}: expected [E,{n}]")
    if row_map is not None:
        _chk(row_map, "row_map", torch.int64, 1, align=8)
    if row_scale is not None:
        _chk(row_scale, "row_scale", torch.float32, align=4)
    if residual is not None:
        _chk(residual, "residual", torch.float32, 2)
        if tuple(residual.shape) != tuple(out.shape):
            raise RuntimeError("residual: expected the shape of out")
    if X.dtype not in (torch.float16, torch.bfloat16) or M == 0:
        return None
    if not ffn_fused_available():
        raise _lib.SlimMoEError("smoe_expert_ffn is not in this build of libslimmoe_hip.so (it measured slower than the two launches and "
                                "left the default build in round 5): make -C slim-switch-moe-vit_amd/csrc FFN=-DSMOE_FFN_FUSED")
    if H is None:
        H = torch.empty((M, h), dtype=X.dtype, device=X.device)
    else:
        _chk(H, "H", X.dtype, 2)
    lib = _lib.load()
    ws_bytes = lib.smoe_expert_ffn_workspace_bytes(M, G)
    ws = _ffn_workspace(X.device, ws_bytes)
    rows = M if rows_hint is None else rows_hint
    with _timed("expert_ffn", {"flops": 2.0 * rows * h * (d_in + d_out)}, X):
        rc = lib.smoe_expert_ffn(_ptr(X), _ptr(a_gather), a_div, _ptr(W1), _ptr(b1), _ptr(H), _ptr(W2), _ptr(b2), _ptr(offsets),
                                 _ptr(group_expert), G, E, M, d_in, h, d_out, dtype_code(X.dtype), _ptr(row_map), _ptr(row_scale),
                                 _ptr(residual), _ptr(out), out.shape[0], F32, _ptr(ws), ws.numel(), _stream(X))
    if rc == -1:
        return None
    _lib.check(rc, "smoe_expert_ffn")
    return out


def grouped_gemm_gelu_keep(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], offsets: torch.Tensor,
                           group_expert: Optional[torch.Tensor] = None, variant: int = DEFAULT_GEMM_VARIANT,
                           group_end: Optional[torch.Tensor] = None):
    """(H, gelu(H)) with H = A W[e]^T + bias[e] per row group -- the first expert linear of the training forward, which keeps
    both (gelu' needs H, the second linear's weight gradient needs gelu(H)): one epilogue with two stores where the kernel
    has it, else the GEMM followed by the GELU pass."""
    _chk(A, "A", ndim=2)
    _chk(W, "W", A.dtype, 3)
    _chk(offsets, "offsets", torch.int32, 1)
    M, K = A.shape
    E, N, _ = W.shape
    G = offsets.numel() - 1 if group_end is None else group_end.numel()
    if group_end is not None:
        _chk(group_end, "group_end", torch.int32, 1)
    if A.dtype in (torch.float16, torch.bfloat16) and W.shape[2] == K and (group_expert is not None or G == E):
        if bias is not None:
            _chk(bias, "bias", torch.float32, 2)
        if group_expert is not None:
            _chk(group_expert, "group_expert", torch.int32, 1)
        pre = torch.empty((M, N), dtype=A.dtype, device=A.device)
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
        rc = _lib.load().smoe_grouped_gemm_gelu_keep(_ptr(A), _ptr(W), _ptr(bias), _ptr(offsets), _ptr(group_expert), _ptr(group_end),
                                                     G, E, M, K, N, dtype_code(A.dtype), _ptr(pre), _ptr(out), _stream(A))
        if rc == 0:
            return pre, out
        if rc != -1:
            _lib.check(rc, "smoe_grouped_gemm_gelu_keep")
    if group_end is not None:
        raise RuntimeError("grouped_gemm_gelu_keep: separate row ranges (group_end) need the persistent kernel's shapes "
                           "(16-bit operands, K % 64 == 0, <= 63 groups)")
    pre = grouped_gemm(A, W, bias, offsets, EPI_NONE, A.dtype, variant=variant, group_expert=group_expert)
    return pre, gelu(pre)


def cast(src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    _chk(src, "src")
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    lib = _lib.load()
    rc = lib.smoe_cast(_ptr(src), dtype_code(src.dtype), _ptr(dst), dtype_code(dtype), src.numel(), _stream(src))
    _lib.check(rc, "smoe_cast")
    return dst


def transpose_cast(src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """``src [B, R, C] -> [B, C, R]`` in ``dtype`` (R, C multiples of 64): the second, transposed operand image of an expert
    weight that the dgrad GEMMs read, made in one pass from the master weight."""
    _chk(src, "src")
    if src.dim() != 3 or src.shape[1] % 64 or src.shape[2] % 64:
        raise ValueError(f"transpose_cast: [B, R, C] with R, C multiples of 64, got {tuple(src.shape)}")
    B, R, C = src.shape
    dst = torch.empty((B, C, R), dtype=dtype, device=src.device)
    rc = _lib.load().smoe_transpose_cast(_ptr(src), dtype_code(src.dtype), _ptr(dst), dtype_code(dtype), B, R, C, _stream(src))
    _lib.check(rc, "smoe_transpose_cast")
    return dst


# ------------------------------------------------------------------------------------------ backward pieces
def gelu(src: torch.Tensor) -> torch.Tensor:
    _chk(src, "src")
    dst = torch.empty_like(src)
    rc = _lib.load().smoe_gelu(_ptr(src), _ptr(dst), dtype_code(src.dtype), src.numel(), _stream(src))
    _lib.check(rc, "smoe_gelu")
    return dst


def rowdot(dout: torch.Tensor, y: torch.Tensor, inv_pos: torch.Tensor, k: int) -> torch.Tensor:
    """dscore[i] = <dout[i // k], y[inv_pos[i]]> (0 for dropped entries)."""
    _chk(dout, "dout", ndim=2)
    _chk(y, "y", ndim=2)
    _chk(inv_pos, "inv_pos", torch.int64, align=8)
    n = inv_pos.numel()
    out = torch.empty(n, dtype=torch.float32, device=dout.device)
    rc = _lib.load().smoe_rowdot(_ptr(dout), dtype_code(dout.dtype), _ptr(y), dtype_code(y.dtype), _ptr(inv_pos), n, k,
                                 dout.shape[1], _ptr(out), _stream(dout))
    _lib.check(rc, "smoe_rowdot")
    return out


def switch_aux(probs: torch.Tensor, counts: torch.Tensor):
    """(aux f32 [] , coef f32 [E]): the SwitchGate's load-balance loss E sum_e frac_e prob_e from the router's probabilities [T, E] and
    the plan's kept counts (i32 [E]), and d aux / d probs[t, e] = E frac_e / kept -- two launches, no host-launched [E]-sized ops."""
    _chk(probs, "probs", torch.float32, 2, align=4)
    _chk(counts, "counts", torch.int32, 1)
    T, E = probs.shape
    if counts.numel() != E:
        raise RuntimeError("switch_aux: one count per expert")
    lib = _lib.load()
    out = torch.empty(E + 1, dtype=torch.float32, device=probs.device)
    ws_bytes = lib.smoe_switch_aux_workspace_bytes(T, E)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=probs.device)
    rc = lib.smoe_switch_aux(_ptr(probs), _ptr(counts), T, E, _ptr(out), out.data_ptr() + 4, _ptr(ws), ws_bytes, _stream(probs))
    _lib.check(rc, "smoe_switch_aux")
    return out[0], out[1:]


def switch_gate_bwd(probs: torch.Tensor, idx: torch.Tensor, dscore: Optional[torch.Tensor], coef: Optional[torch.Tensor],
                    coef_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dlogits [T, E] of the SwitchGate's score (= probs[t, idx[t]]) and load-balance loss in one pass (softmax backward of
    g[t, e] = coef[e] * coef_scale + (e == idx[t]) dscore[t]); ``coef_scale``: a one-element f32 device tensor (d loss / d aux)."""
    _chk(probs, "probs", torch.float32, 2, align=4)
    T, E = probs.shape
    _chk(idx, "idx", torch.int64, align=8)
    if idx.numel() != T:
        raise RuntimeError("switch_gate_bwd: one index per token")
    if dscore is not None:
        _chk(dscore, "dscore", torch.float32, align=4)
    if coef is not None:
        _chk(coef, "coef", torch.float32, 1, align=4)
    if coef_scale is not None:
        _chk(coef_scale, "coef_scale", torch.float32, align=4)
    out = torch.empty_like(probs)
    rc = _lib.load().smoe_switch_gate_bwd(_ptr(probs), _ptr(idx), _ptr(dscore), _ptr(coef), _ptr(coef_scale), T, E, _ptr(out),
                                          _stream(probs))
    _lib.check(rc, "smoe_switch_gate_bwd")
    return out


def pad_offsets(offsets: torch.Tensor) -> torch.Tensor:
    _chk(offsets, "offsets", torch.int32, 1)
    out = torch.empty_like(offsets)
    rc = _lib.load().smoe_pad_offsets(_ptr(offsets), offsets.numel() - 1, _ptr(out), _stream(offsets))
    _lib.check(rc, "smoe_pad_offsets")
    return out


def split_offsets(offsets: torch.Tensor, S: int) -> torch.Tensor:
    """i32 [G * S + 1]: every row group of ``offsets`` cut into S pseudo-groups of whole 64-row chunks (smoe_split_offsets)."""
    _chk(offsets, "offsets", torch.int32, 1, align=4)
    G = offsets.numel() - 1
    out = torch.empty(G * int(S) + 1, dtype=torch.int32, device=offsets.device)
    rc = _lib.load().smoe_split_offsets(_ptr(offsets), G, int(S), _ptr(out), _stream(offsets))
    _lib.check(rc, "smoe_split_offsets")
    return out


def expert_wgrad_splits(G: int, R1: int, R2: int, rows: int, device) -> int:
    """Pieces to cut every expert's rows into for its weight gradient [R1, R2]: the kernel's grid is groups x output tiles, and at
    small widths that is a fraction of the chip (DeiT-Tiny, models/resMoE.py:151-187: 8 experts x 3 tiles = 24 workgroups walking 6 k rows
    each, 100 us per launch and a third of the training step).  Enough pieces for ~one round of workgroups, at least ~512 rows each."""
    tiles = G * min(-(-R1 // 256) * -(-R2 // 256), -(-R2 // 256) * -(-R1 // 256))
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    if tiles * 2 > cus:
        return 1
    return max(1, min(16, cus // max(tiles, 1), rows // max(G, 1) // 512))


def split_ranges(starts: torch.Tensor, ends: torch.Tensor, S: int):
    """smoe_split_offsets' rule for SEPARATE row ranges [starts[g], ends[g]) (the slots of the static expert exchange): S pieces of whole
    64-row chunks each, the last ones possibly empty -> (starts i32 [G * S], ends i32 [G * S]); a few [G, S]-sized device ops, no sync."""
    cnt = (ends - starts).to(torch.int64).view(-1, 1)
    step = ((cnt + S * 64 - 1) // (S * 64)) * 64
    at = torch.arange(S, device=starts.device, dtype=torch.int64).view(1, S) * step
    lo = starts.to(torch.int64).view(-1, 1)
    s0 = lo + torch.minimum(at, cnt)
    e0 = lo + torch.minimum(at + step, cnt)
    return s0.reshape(-1).to(torch.int32).contiguous(), e0.reshape(-1).to(torch.int32).contiguous()


def grouped_wgrad_rows_split(P: torch.Tensor, Q: torch.Tensor, offsets: torch.Tensor, S: int,
                             offsets_split=None, group_end: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``grouped_wgrad_rows`` with every group cut into S pieces (``offsets_split`` = split_offsets(offsets, S), reusable by the
    layer's second weight gradient) and the partial products summed in piece order: f32 [G, R1, R2].  With ``group_end`` (separate
    row ranges) ``offsets_split`` is the pair split_ranges(offsets, group_end, S)."""
    if group_end is not None:
        G = group_end.numel()
        if S <= 1:
            return grouped_wgrad_rows(P, Q, offsets, group_end=group_end)
        s0, e0 = offsets_split if offsets_split is not None else split_ranges(offsets, group_end, S)
        part = grouped_wgrad_rows(P, Q, s0, group_end=e0)
        return part.view(G, S, part.shape[1], part.shape[2]).sum(1)
    G = offsets.numel() - 1
    if S <= 1:
        return grouped_wgrad_rows(P, Q, offsets)
    if offsets_split is None:
        offsets_split = split_offsets(offsets, S)
    part = grouped_wgrad_rows(P, Q, offsets_split)
    return part.view(G, S, part.shape[1], part.shape[2]).sum(1)


def padded_len(n_rows: int, E: int) -> int:
    """Static upper bound of offsets_pad[E]: every expert range rounded up to 64."""
    return ((n_rows + 63) // 64) * 64 + 64 * E


def transpose_pad(src: torch.Tensor, offsets: torch.Tensor, offsets_pad: torch.Tensor, Lp: int) -> torch.Tensor:
    """[n, C] expert-sorted rows -> [C, Lp] K-major image with 64-aligned, zero-padded expert ranges."""
    _chk(src, "src", ndim=2)
    n, C = src.shape
    dst = torch.empty((C, Lp), dtype=src.dtype, device=src.device)
    rc = _lib.load().smoe_transpose_pad(_ptr(src), dtype_code(src.dtype), _ptr(offsets), _ptr(offsets_pad),
                                        offsets.numel() - 1, n, C, Lp, _ptr(dst), _stream(src))
    _lib.check(rc, "smoe_transpose_pad")
    return dst


def grouped_wgrad(PT: torch.Tensor, QT: torch.Tensor, offsets_pad: torch.Tensor) -> torch.Tensor:
    """out[e] = PT[:, range e] @ QT[:, range e]^T, f32 [E, R1, R2]."""
    _chk(PT, "PT", ndim=2)
    _chk(QT, "QT", PT.dtype, 2)
    E = offsets_pad.numel() - 1
    R1, Lp = PT.shape
    R2 = QT.shape[0]
    if QT.shape[1] != Lp:
        raise RuntimeError("PT / QT: padded lengths differ")
    out = torch.empty((E, R1, R2), dtype=torch.float32, device=PT.device)
    with _timed("grouped_wgrad", {"K": Lp}, PT):
        rc = _lib.load().smoe_grouped_wgrad(_ptr(PT), _ptr(QT), dtype_code(PT.dtype), _ptr(offsets_pad), E, R1, R2, Lp,
                                            _ptr(out), _stream(PT))
    _lib.check(rc, "smoe_grouped_wgrad")
    return out


_zero_pages = StreamCache()


_ones_cache = {}


def ones_f32(n: int, device) -> torch.Tensor:
    """A cached f32 vector of ones (read-only: unit combine weights of the scatter's adjoint) -- not a fill kernel per backward."""
    key = (torch.device(device).index, int(n))
    t = _ones_cache.get(key)
    if t is None:
        if len(_ones_cache) > 64:
            _ones_cache.clear()
        t = _ones_cache[key] = torch.ones(int(n), dtype=torch.float32, device=device)
    return t


def _zero16(device) -> torch.Tensor:
    return _zero_pages.get(str(device), 0, lambda: torch.zeros(64, dtype=torch.uint8, device=device))


def _wgrad_rounds(E: int, R1: int, R2: int, cus: int) -> float:
    """Cost-weighted rounds of workgroups of smoe_grouped_wgrad_rows (256- or 320-row tiles; csrc/gemm.hip launch_wgrad_rows)."""
    tn = -(-R2 // 256)
    c4 = -(-(E * -(-R1 // 256) * tn) // cus)
    c5 = 1.25 * -(-(E * -(-R1 // 320) * tn) // cus)
    return min(c4, c5)


def grouped_wgrad_rows(P: torch.Tensor, Q: torch.Tensor, offsets: torch.Tensor, allow_swap: bool = True,
                       group_end: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[e] (f32 [R1, R2]) = P[rows of e]^T @ Q[rows of e] from the token-major operands (no transposed copies).
    ``group_end`` (i32 [G]): separate row ranges [offsets[g], group_end[g]) (``offsets`` then needs G entries only).
    The kernel's tile is taller (256 / 320 output rows) than wide along R1 only, so when the transposed problem needs fewer
    rounds of workgroups (ViT-B's dW2 [768, 3072]: 288 tiles = two rounds on 256 CUs; as [3072, 768]: 240 taller tiles = one)
    it is computed as Q^T P and transposed back by smoe_transpose_cast (one pass over the f32 result)."""
    _chk(P, "P", ndim=2)
    _chk(Q, "Q", ndim=2)
    if P.dtype != Q.dtype or P.dtype not in (torch.float16, torch.bfloat16) or P.shape[0] != Q.shape[0]:
        raise RuntimeError("grouped_wgrad_rows: P and Q must be f16 / bf16 with the same row count")
    _chk(offsets, "offsets", torch.int32, 1)
    if group_end is not None:
        _chk(group_end, "group_end", torch.int32, 1)
        if offsets.numel() < group_end.numel() or P.numel() >= 2 ** 32 or Q.numel() >= 2 ** 32:
            raise RuntimeError("grouped_wgrad_rows: group_end needs an offsets entry per group and operands under 2^32 elements")
    E = offsets.numel() - 1 if group_end is None else group_end.numel()
    R1, R2 = P.shape[1], Q.shape[1]
    if P.numel() >= 2 ** 32 or Q.numel() >= 2 ** 32:
        # the token-major kernel addresses its operands with 32-bit ELEMENT offsets (csrc/gemm.hip MODE 2): an operand of
        # 2^32 elements or more takes the K-major path (64-padded transposed images, 64-bit row bases)
        offs_pad = pad_offsets(offsets)
        Lp = padded_len(P.shape[0], E)
        return grouped_wgrad(transpose_pad(P, offsets, offs_pad, Lp), transpose_pad(Q, offsets, offs_pad, Lp), offs_pad)
    if allow_swap and R1 % 64 == 0 and R2 % 64 == 0:
        cus = torch.cuda.get_device_properties(P.device).multi_processor_count
        if _wgrad_rounds(E, R2, R1, cus) + 0.15 < _wgrad_rounds(E, R1, R2, cus):
            return transpose_cast(grouped_wgrad_rows(Q, P, offsets, allow_swap=False, group_end=group_end), torch.float32)
    out = torch.empty((E, R1, R2), dtype=torch.float32, device=P.device)
    rc = _lib.load().smoe_grouped_wgrad_rows(_ptr(P), _ptr(Q), dtype_code(P.dtype), _ptr(offsets), _ptr(group_end), E, R1, R2,
                                              _ptr(_zero16(P.device)), _ptr(out), _stream(P))
    _lib.check(rc, "smoe_grouped_wgrad_rows")
    return out


def gate_wgrad(dl: torch.Tensor, x: torch.Tensor, want_bias: bool = False):
    """dWg [E, d] f32 = dl^T x for dl [T, E] f32, x [T, d]: the router weight gradient as a streaming reduction.
    ``want_bias``: returns (dWg, dbg [E] = column sums of dl) from the same pass."""
    _chk(dl, "dl", torch.float32, 2, align=4)
    _chk(x, "x", ndim=2)
    T, E = dl.shape
    C = x.shape[1]
    if x.shape[0] != T:
        raise RuntimeError("gate_wgrad: row counts differ")
    lib = _lib.load()
    out = torch.empty((E, C), dtype=torch.float32, device=x.device)
    db = torch.empty(E, dtype=torch.float32, device=x.device) if want_bias else None
    ws_bytes = lib.smoe_gate_wgrad_workspace_bytes(T, E, C)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    rc = lib.smoe_gate_wgrad(_ptr(dl), _ptr(x), dtype_code(x.dtype), T, E, C, _ptr(out), _ptr(db), _ptr(ws), ws_bytes, _stream(x))
    _lib.check(rc, "smoe_gate_wgrad")
    return (out, db) if want_bias else out


def gate_dgrad(dl: torch.Tensor, w: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    """dx [T, d] = dl [T, E] @ w [E, d] (f32 inputs): the router linear's input gradient as a streaming kernel."""
    _chk(dl, "dl", torch.float32, 2, align=4)
    _chk(w, "w", torch.float32, 2)
    T, E = dl.shape
    if w.shape[0] != E:
        raise RuntimeError("gate_dgrad: dl / w disagree on E")
    d = w.shape[1]
    out = torch.empty((T, d), dtype=out_dtype, device=dl.device)
    rc = _lib.load().smoe_gate_dgrad(_ptr(dl), _ptr(w), T, E, d, _ptr(out), dtype_code(out_dtype), _stream(dl))
    _lib.check(rc, "smoe_gate_dgrad")
    return out


def zero_group_fold(cs2: torch.Tensor, cs1: Optional[torch.Tensor], A: torch.Tensor, offsets: torch.Tensor, gmap: torch.Tensor,
                    E: int, dW2: torch.Tensor, want_b2: bool = True, want_b1: bool = True):
    """The zero-row groups' share of the expert gradients in one launch (smoe_zero_group_fold): ``dW2`` [E, d, h] gets the rank-1
    terms colsum(dY_g) (x) A_row IN PLACE; returns (db2 [E, d] | None, db1 [E, h] | None) = own group's column sums + those of the
    zero groups mapped to the expert.  cs2 [G, d] / cs1 [G, h]: column sums over all G = E + Z groups."""
    _chk(cs2, "cs2", torch.float32, 2)
    _chk(A, "A", ndim=2)
    _chk(offsets, "offsets", torch.int32, 1, align=4)
    _chk(gmap, "gmap", torch.int32, 1, align=4)
    _chk(dW2, "dW2", torch.float32, 3)
    G, d = cs2.shape
    h = A.shape[1]
    Z = G - E
    if Z < 0 or offsets.numel() != G + 1 or gmap.numel() != G or tuple(dW2.shape) != (E, d, h) or h % 4:
        raise RuntimeError("zero_group_fold: shapes disagree")
    if cs1 is not None:
        _chk(cs1, "cs1", torch.float32, 2)
        if tuple(cs1.shape) != (G, h):
            raise RuntimeError("zero_group_fold: cs1 must be [G, h]")
    want_b1 = want_b1 and cs1 is not None
    db2 = torch.empty((E, d), dtype=torch.float32, device=cs2.device) if want_b2 else None
    db1 = torch.empty((E, h), dtype=torch.float32, device=cs2.device) if want_b1 else None
    rc = _lib.load().smoe_zero_group_fold(_ptr(cs2), _ptr(cs1), _ptr(A), dtype_code(A.dtype), _ptr(offsets), _ptr(gmap), E, Z, d, h,
                                          A.shape[0], _ptr(dW2), _ptr(db2), _ptr(db1), _stream(cs2))
    _lib.check(rc, "smoe_zero_group_fold")
    return db2, db1


def group_colsum(src: torch.Tensor, offsets: torch.Tensor, group_end: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[e, c] = sum of the rows of group e (bias gradients); deterministic two-pass reduction.  ``group_end`` (i32 [G]): separate
    row ranges [offsets[g], group_end[g]) -- ``offsets`` then needs G entries only."""
    _chk(src, "src", ndim=2)
    _chk(offsets, "offsets", torch.int32, 1)
    if group_end is not None:
        _chk(group_end, "group_end", torch.int32, 1)
        if offsets.numel() < group_end.numel():
            raise RuntimeError("group_colsum: offsets must have an entry per group")
    E = offsets.numel() - 1 if group_end is None else group_end.numel()
    n, C = src.shape
    lib = _lib.load()
    out = torch.empty((E, C), dtype=torch.float32, device=src.device)
    ws_bytes = lib.smoe_group_colsum_workspace_bytes(n, E, C)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=src.device)
    rc = lib.smoe_group_colsum(_ptr(src), dtype_code(src.dtype), _ptr(offsets), _ptr(group_end), E, n, C, _ptr(out), _ptr(ws),
                               ws_bytes, _stream(src))
    _lib.check(rc, "smoe_group_colsum")
    return out
