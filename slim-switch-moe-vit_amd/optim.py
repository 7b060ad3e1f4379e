"""Optimizer side of the reference's training step on the HIP path.

The reference trains with ``timm.optim.create_optimizer_v2(..., opt="adamw")`` (main.py:729-731; ``--opt adamw``,
``--opt-eps 1e-8``, ``--weight-decay 0.05``) = ``torch.optim.AdamW``, stepped through ``timm.utils.NativeScaler``
(main.py:732, engine.py:68-74):

    loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=is_second_order)

i.e. ``scaler.scale(loss).backward(); scaler.unscale_(optimizer); clip_grad_norm_(parameters, clip_grad);
scaler.step(optimizer); scaler.update()``.  The MoE's expert tensors are ``[E,h,d]`` / ``[E,d,h]`` f32 -- 150 MB per layer
at ViT-B, E = 8 -- and every one of those calls is a full HBM pass (or two) over them and their gradients.  Here the
same step is TWO passes: ``ops``-level kernels ``smoe_grad_sumsq`` (norm + non-finite check on the still-scaled
gradient) and ``smoe_adamw_step`` (the update, reading the gradient times ``inv_scale x clip coefficient``), with the loss
scale, the non-finite flag, the clip coefficient and the step count all living on the device: no host sync in the step.

``AdamW`` and ``NativeScaler`` keep the call signatures, ``state_dict`` keys and numerics of what they replace; CPU
tensors (and non-f32 parameters) fall back to the torch composition, which is also what the tests compare against.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch

import os

from . import _lib, ops
from ._cache import param_version, shadow_of

# the fused step also writes the 16-bit operand images of the weights it updates (A/B: SLIMMOE_ADAMW_SHADOW=0)
SHADOW_STEP = os.environ.get("SLIMMOE_ADAMW_SHADOW", "1") != "0"


def _hip_ok(p: torch.Tensor) -> bool:
    return p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()


SUMSQ_BLOCK = 16384   # elements per workgroup of the multi-tensor kernels (csrc/optim.hip)
_blk_cache = {}       # (device, numels) -> (blk int32 [2, n_blocks] on the device, blocks per tensor)


def _block_table(numels, device):
    """Workgroup -> (tensor, 16K-element block) map of the multi-tensor kernels; depends on the tensor sizes only, cached."""
    key = (str(device), tuple(numels))
    hit = _blk_cache.get(key)
    if hit is None:
        import numpy as np
        nb = [(n + SUMSQ_BLOCK - 1) // SUMSQ_BLOCK for n in numels]
        tens = np.repeat(np.arange(len(numels), dtype=np.int32), nb)
        first = np.cumsum([0] + nb[:-1]) if nb else np.zeros(0, dtype=np.int64)
        index = (np.arange(int(sum(nb)), dtype=np.int64) - np.repeat(first, nb)).astype(np.int32)
        blk = torch.from_numpy(np.stack([tens, index]) if len(tens) else np.zeros((2, 0), dtype=np.int32)).to(device)
        if len(_blk_cache) > 16:
            _blk_cache.clear()
        hit = _blk_cache[key] = (blk.contiguous(), nb)
    return hit


_staging = []          # pinned staging buffers of tables copied to the device while a HIP graph was being captured: kept for good
_staging_ring = []     # ... and of the eager steps: the last few (the copy is asynchronous)


def _host_table(rows, dtype, device):
    """A small table (addresses, element counts, per-tensor hyper-parameters) on the device.  The values change with every backward,
    so this is one small host-to-device copy per step -- from PINNED memory, asynchronously: it costs the host no wait, and it can be
    captured (a HIP graph of the whole training step replays the copy from the same pinned buffer, which therefore must outlive the
    graph: buffers staged during a capture are never released)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return torch.tensor(rows, dtype=dtype, device=dev)
    host = torch.tensor(rows, dtype=dtype).pin_memory()
    if torch.cuda.is_current_stream_capturing():
        _staging.append(host)
    else:
        _staging_ring.append(host)
        if len(_staging_ring) > 64:
            del _staging_ring[:32]
    return host.to(dev, non_blocking=True)


def _pointer_table(rows, device):
    """int64 [len(rows), n_t] on the device (addresses and element counts)."""
    return _host_table(rows, torch.int64, device)


class AdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW(params, lr, betas, eps, weight_decay)`` (no amsgrad / maximize) with a fused HIP step.

    ``step(grad_mult=None, found_inf=None)``: optional device scalars (f32[1]) -- every gradient is multiplied by
    ``grad_mult`` on the fly, and the whole step (including the step count) is skipped when ``found_inf`` is non-zero;
    this is how ``NativeScaler`` drives it without unscaling gradients in memory or asking the host."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("AdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step_dev = {}       # device -> f32[1] number of updates applied so far (advanced on the device)
        self._step_restore = {}   # device -> count loaded from a checkpoint, applied when the counter is (re)created

    supports_device_scalars = True
    _slimmoe_refreshes_images = True     # (_foreign_step_hook: this step bumps the versions / refreshes the 16-bit images itself)

    def _step_counter(self, device) -> torch.Tensor:
        t = self._step_dev.get(device)
        if t is None:
            t = self._step_dev[device] = torch.full((1,), float(self._step_restore.pop(device, 0.0)), dtype=torch.float32,
                                                    device=device)
        return t

    def _on_hip_path(self, p) -> bool:
        return _hip_ok(p)

    # -- checkpoints (the reference saves / restores `optimizer.state_dict()`: main.py:898, 717) -------------------------------
    def state_dict(self):
        """torch.optim.AdamW's layout.  ``state[p]['step']`` is the number of updates APPLIED to p (the bias-correction
        t): for parameters on the HIP path that is the device-side counter (steps skipped for non-finite gradients are
        not counted), read back here -- a checkpoint is the one place where the host may ask."""
        for dev, t in self._step_dev.items():
            n = int(float(t))
            for group in self.param_groups:
                for p in group["params"]:
                    st = self.state.get(p)
                    if st and p.device == dev and self._on_hip_path(p):
                        st["step"] = n
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Restores exp_avg / exp_avg_sq AND the step count the kernels' bias correction uses (a ``torch.optim.AdamW``
        checkpoint carries ``step`` as a tensor per parameter, this class's own as an int: both load)."""
        super().load_state_dict(state_dict)
        self._step_dev, self._step_restore = {}, {}
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if not st or "step" not in st:
                    continue
                n = float(st["step"])          # tensor or number
                st["step"] = int(n)
                if self._on_hip_path(p):       # one counter per device: every tensor of a step shares t
                    self._step_restore[p.device] = max(self._step_restore.get(p.device, 0.0), n)

    @torch.no_grad()
    def step(self, closure=None, grad_mult: Optional[torch.Tensor] = None, found_inf: Optional[torch.Tensor] = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = None
        advanced = set()
        refreshed = []   # [(parameter, (cache, key, image))]: 16-bit images the fused step wrote
        batches = {}   # (device, grad dtype, betas, eps) -> [(p, g, exp_avg, exp_avg_sq, lr, wd)]: one launch each
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group["lr"], group["betas"], group["eps"], group["weight_decay"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["step"] = 0
                g = p.grad
                if _hip_ok(p) and g.is_cuda and g.is_contiguous() and g.dtype in ops._DT:
                    batches.setdefault((p.device, g.dtype, float(b1), float(b2), float(eps)), []).append(
                        (p, g, st["exp_avg"], st["exp_avg_sq"], float(lr), float(wd)))
                else:                 # torch composition (CPU tensors, other dtypes): same arithmetic
                    if found_inf is not None and float(found_inf) != 0.0:
                        continue
                    gg = g.float() * (float(grad_mult) if grad_mult is not None else 1.0)
                    st["step"] += 1
                    t = st["step"]
                    p.mul_(1 - lr * wd)
                    st["exp_avg"].lerp_(gg, 1 - b1)
                    st["exp_avg_sq"].mul_(b2).addcmul_(gg, gg, value=1 - b2)
                    denom = (st["exp_avg_sq"].sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
                    p.addcdiv_(st["exp_avg"], denom, value=-lr / (1 - b1 ** t))
        for (dev, gdt, b1, b2, eps), items in batches.items():
            if lib is None:
                lib = _lib.load()
            step_t = self._step_counter(dev)
            stream = ops._stream(items[0][0])
            if dev not in advanced:   # one counter per device: every parameter of a step sees the same t
                _lib.check(lib.smoe_step_advance(step_t.data_ptr(), ops._ptr(found_inf), stream), "smoe_step_advance")
                advanced.add(dev)
            if len(items) == 1:
                p, g, m, v, lr, wd = items[0]
                rc = lib.smoe_adamw_step(p.data_ptr(), g.data_ptr(), ops.dtype_code(gdt), m.data_ptr(), v.data_ptr(), p.numel(),
                                         lr, b1, b2, eps, wd, step_t.data_ptr(), ops._ptr(grad_mult), ops._ptr(found_inf), stream)
                _lib.check(rc, "smoe_adamw_step")
                continue
            # every tensor of the batch in ONE launch (a ViT-B/16 MoE has ~180 parameter tensors)
            blk, nb = _block_table([it[0].numel() for it in items], dev)
            tab = _pointer_table([[it[0].data_ptr() for it in items], [it[1].data_ptr() for it in items],
                                  [it[2].data_ptr() for it in items], [it[3].data_ptr() for it in items],
                                  [it[0].numel() for it in items]], dev)
            hyp = _host_table([[it[4] for it in items], [it[5] for it in items]], torch.float32, dev)
            # 16-bit operand images of the parameters that are current NOW (the modules' weight shadows): written in the same pass,
            # so the next forward finds them current (no f32 -> 16-bit cast pass over every weight per step)
            shadows = [shadow_of(it[0]) if SHADOW_STEP else None for it in items]
            sh_tab = None
            if any(sh is not None for sh in shadows):
                sh_tab = _pointer_table([[sh[2].data_ptr() if sh is not None else 0 for sh in shadows],
                                         [ops.dtype_code(sh[2].dtype) if sh is not None else 0 for sh in shadows]], dev)
            rc = lib.smoe_adamw_step_multi(tab.data_ptr(), hyp.data_ptr(), len(items), blk.data_ptr(), sum(nb), ops.dtype_code(gdt),
                                           b1, b2, eps, step_t.data_ptr(), ops._ptr(grad_mult), ops._ptr(found_inf), ops._ptr(sh_tab),
                                           stream)
            _lib.check(rc, "smoe_adamw_step_multi")
            refreshed.extend((it[0], sh) for it, sh in zip(items, shadows) if sh is not None)
        # The kernels wrote the parameters through raw pointers: autograd's version counters did not move, and every derived
        # tensor keyed on them (the 16-bit weight shadows of the expert GEMMs and of the dense projections, the zero-row
        # constants) would go on serving the OLD weights.  Bump the versions, as an in-place torch op would have.
        for items in batches.values():
            _bump_versions([it[0] for it in items])
        for p, (cache, key, img) in refreshed:        # the images written above belong to the NEW version of their parameter
            cache.refresh(key, img, param_version(p))
        return loss


def _bump_versions(tensors) -> None:
    """``torch._C._increment_version`` is private API: newer releases take a list, older ones a single tensor.  If neither form
    works the derived-tensor caches are dropped wholesale -- slower (every shadow is re-cast on its next use), never stale."""
    try:
        torch._C._increment_version(tensors)
        return
    except (TypeError, AttributeError):
        pass
    try:
        for t in tensors:
            torch._C._increment_version(t)
    except (TypeError, AttributeError):
        from ._cache import invalidate_all
        invalidate_all()


def _foreign_step_hook(optimizer, args, kwargs) -> None:
    """Global optimizer step post-hook (registered on import, below).  The modules keep 16-bit operand images of their f32
    weights, keyed on ``(param._version, data_ptr, dtype)`` (_cache.param_version).  An optimizer that updates through
    ``p.data`` -- every timm-native optimizer reachable from the reference's ``--opt`` (main.py:90-96, 729-731) does -- moves the
    weights WITHOUT moving ``_version``, and the next forward would go on reading the old images: training on stale weights, no
    error, no warning.  After ANY optimizer's step other than this module's AdamW (whose fused step refreshes the images itself)
    the version counters of its parameters are therefore bumped here, as an in-place torch op would have: the images are re-cast
    on their next use.  (A hand-written update outside a torch.optim.Optimizer -- ``p.data.add_(...)`` in a loop -- cannot be seen:
    call ``slim_switch_moe_vit_amd.invalidate_weight_images()`` after it.)"""
    if getattr(optimizer, "_slimmoe_refreshes_images", False):
        return
    params = [p for group in optimizer.param_groups for p in group["params"] if isinstance(p, torch.Tensor)]
    if params:
        _bump_versions(params)


def invalidate_weight_images() -> None:
    """Drop every derived tensor (16-bit weight images, transposed images, zero-row constants): call after changing parameters
    behind autograd's back (``p.data`` writes outside an optimizer, raw-pointer writes)."""
    from ._cache import invalidate_all
    invalidate_all()


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post
    _FOREIGN_HOOK = _reg_post(_foreign_step_hook)
except ImportError:      # (a torch without global optimizer hooks: the documented invalidate_weight_images() remains)
    _FOREIGN_HOOK = None


class NativeScaler:
    """``timm.utils.NativeScaler`` (main.py:732; called at engine.py:68-74) on device-side state.

    ``__call__(loss, optimizer, clip_grad=None, parameters=None, create_graph=False)``: scaled backward, gradient-norm
    clipping to ``clip_grad`` when it is not None (``parameters`` then required, as in timm), optimizer step skipped on
    non-finite gradients, scale update (growth 2.0 every 2000 clean steps, backoff 0.5) -- ``torch.cuda.amp.GradScaler``
    defaults.  With an optimizer that takes device scalars (``optim.AdamW``) nothing in the step syncs with the host and
    the gradients stay SCALED in ``.grad`` (``grad_multiplier`` holds ``inv_scale x clip coefficient``); any other optimizer
    gets the stock sequence (unscale in place, ``clip_grad_norm_``, host-checked step).
    ``state_dict()`` has GradScaler's keys (main.py:903 saves it, 723 loads it)."""

    state_dict_key = "amp_scaler"

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.enabled = enabled
        self._init_scale = float(init_scale)
        self._scale = None            # f32[1] on the device, created at the first call
        self._growth_tracker = None   # f32[1]
        self.grad_multiplier = None   # f32[1]: what the last step multiplied the stored gradients by
        self.last_grad_norm = None    # f32[1]: total norm of the unscaled gradients of the last clipped step

    def _lazy(self, device):
        if self._scale is None or self._scale.device != device:
            s = self._init_scale if self._scale is None else float(self._scale)
            t = 0.0 if self._growth_tracker is None else float(self._growth_tracker)
            self._scale = torch.full((1,), s, dtype=torch.float32, device=device)
            self._growth_tracker = torch.full((1,), t, dtype=torch.float32, device=device)

    def get_scale(self) -> float:
        return self._init_scale if self._scale is None else float(self._scale)

    def state_dict(self):
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval,
                "_growth_tracker": 0 if self._growth_tracker is None else int(self._growth_tracker)}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = sd["growth_factor"], sd["backoff_factor"]
        self.growth_interval = sd["growth_interval"]
        self._init_scale = float(sd["scale"])
        dev = None if self._scale is None else self._scale.device
        self._scale = self._growth_tracker = None
        if dev is not None:
            self._lazy(dev)
        if self._growth_tracker is not None:
            self._growth_tracker.fill_(float(sd["_growth_tracker"]))
        else:
            self._pending_tracker = float(sd["_growth_tracker"])

    def __call__(self, loss, optimizer, clip_grad=None, parameters: Optional[Iterable] = None, create_graph: bool = False):
        if not self.enabled:
            loss.backward(create_graph=create_graph)
            if clip_grad is not None:
                assert parameters is not None
                torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
            optimizer.step()
            return
        dev = loss.device
        self._lazy(dev)
        if getattr(self, "_pending_tracker", None) is not None:
            self._growth_tracker.fill_(self._pending_tracker)
            self._pending_tracker = None
        (loss * self._scale[0]).backward(create_graph=create_graph)
        params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
        fused = bool(getattr(optimizer, "supports_device_scalars", False)) and dev.type == "cuda" and all(
            p.grad.is_cuda and p.grad.is_contiguous() and p.grad.dtype in ops._DT for p in params)
        if not fused:
            self._stock_step(optimizer, params, clip_grad, parameters)
            return
        lib = _lib.load()
        found_inf = torch.zeros(1, dtype=torch.float32, device=dev)
        inv_scale = self._scale.reciprocal()
        clip_over = list(parameters) if (clip_grad is not None and parameters is not None) else None
        if clip_grad is not None:
            assert parameters is not None, "clip_grad needs `parameters` (timm NativeScaler)"
        # pass 1 over every gradient the optimizer will use: non-finite check (+ the norm's partial sums where clipped)
        grads = [p.grad for p in params]
        clip_ids = {id(p) for p in clip_over if p.grad is not None} if clip_over is not None else set()
        # one launch per gradient dtype over every gradient (partials in parameter order, 16K-element blocks)
        by_dtype = {}
        for p, g in zip(params, grads):
            if g.numel():
                by_dtype.setdefault(g.dtype, []).append((p, g))
        partials, masks = [], []
        for gdt, items in by_dtype.items():
            blk, nb = _block_table([g.numel() for _, g in items], dev)
            zeros = [0] * len(items)
            tab = _pointer_table([zeros, [g.data_ptr() for _, g in items], zeros, zeros, [g.numel() for _, g in items]], dev)
            part = torch.empty(sum(nb), dtype=torch.float32, device=dev)
            rc = lib.smoe_grad_sumsq_multi(tab.data_ptr(), len(items), blk.data_ptr(), sum(nb), ops.dtype_code(gdt),
                                           inv_scale.data_ptr(), part.data_ptr(), found_inf.data_ptr(), ops._stream(part))
            _lib.check(rc, "smoe_grad_sumsq_multi")
            partials.append(part)
            if clip_grad is not None:
                masks.append(self._clip_mask([id(p) in clip_ids for p, _ in items], nb, dev))
        mult = inv_scale
        if clip_grad is not None:
            total = torch.zeros((), dtype=torch.float32, device=dev)
            for part, mask in zip(partials, masks):
                if mask is not None:
                    total = total + torch.where(mask, part, part.new_zeros(())).sum()
            total = total.sqrt()
            self.last_grad_norm = total.reshape(1)
            coef = (float(clip_grad) / (total + 1e-6)).clamp(max=1.0)      # torch.nn.utils.clip_grad_norm_
            mult = inv_scale * coef
        self.grad_multiplier = mult.reshape(1).contiguous()
        optimizer.step(grad_mult=self.grad_multiplier, found_inf=found_inf)
        rc = lib.smoe_amp_update(self._scale.data_ptr(), self._growth_tracker.data_ptr(), found_inf.data_ptr(),
                                 float(self.growth_factor), float(self.backoff_factor), int(self.growth_interval),
                                 ops._stream(self._scale))
        _lib.check(rc, "smoe_amp_update")

    def _clip_mask(self, clipped, nb, dev):
        """bool [n_blocks]: True for the partial sums of gradients that count towards the clipped norm (cached: the pattern
        only changes with the parameter list); None when nothing is clipped."""
        if not any(clipped):
            return None
        key = (str(dev), tuple(clipped), tuple(nb))
        cache = self.__dict__.setdefault("_mask_cache", {})
        m = cache.get(key)
        if m is None:
            if len(cache) > 8:
                cache.clear()
            m = cache[key] = torch.repeat_interleave(torch.tensor(clipped, dtype=torch.bool), torch.tensor(nb)).to(dev)
        return m

    def _stock_step(self, optimizer, params, clip_grad, parameters):
        """GradScaler's sequence for optimizers that know nothing of device scalars (and for CPU tensors)."""
        inv = 1.0 / float(self._scale)
        finite = True
        for p in params:
            p.grad.mul_(inv)
            finite = finite and bool(torch.isfinite(p.grad).all())
        if clip_grad is not None:
            assert parameters is not None, "clip_grad needs `parameters` (timm NativeScaler)"
            self.last_grad_norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad).reshape(1)
        self.grad_multiplier = None
        if finite:
            optimizer.step()
            self._growth_tracker += 1
            if int(self._growth_tracker) >= self.growth_interval:
                self._scale *= self.growth_factor
                self._growth_tracker.zero_()
        else:
            self._scale *= self.backoff_factor
            self._growth_tracker.zero_()
