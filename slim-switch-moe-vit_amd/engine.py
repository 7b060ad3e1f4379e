"""Eval harness mirroring the reference's engine.evaluate (engine.py:88-121): eval mode, no_grad,
autocast on the GPU, cross-entropy + top-1/top-5, per-batch loop.  Returns the same dict keys plus
images/sec.  MetricLogger / distributed meters are out of scope."""
from __future__ import annotations

import time
from typing import Iterable, Tuple

import torch


def _size_static_exchange(model: torch.nn.Module, data_loader) -> None:
    """Expert-parallel capacity gates exchange through buffers sized for the largest batch (ep.static_slot_tokens): the loader
    knows it a priori -- ``batch_size`` images x the model's tokens per image, the same on every rank -- so the harness sets it
    before the first forward instead of letting the first batch decide (a small first batch would under-size the slots)."""
    bs = getattr(data_loader, "batch_size", None)
    pos = getattr(model, "pos_embed", None)
    if bs is None or pos is None or not any(getattr(m, "ep_active", lambda: False)() for m in model.modules()
                                            if hasattr(m, "ep_active")):
        return
    from .ep import set_static_tokens
    set_static_tokens(model, int(bs) * int(pos.shape[1]), unit=int(pos.shape[1]))


def _speculative_alpha(value):
    """``ep_speculative`` of evaluate(): a number >= 1, None / 0 (off), or "auto" = SLIMMOE_EP_ALPHA (default 1.5; 0 = off)."""
    import os
    if value == "auto":
        value = float(os.environ.get("SLIMMOE_EP_ALPHA", "1.5"))
    return float(value) if value else None


def _ep_world_size(model) -> int:
    return max([int(getattr(m, "world_size", 1)) for m in model.modules() if hasattr(m, "ep_active")] or [1])


class GraphedForward:
    """``model(images)`` under fp16 autocast / no_grad, replayed from ONE HIP graph per input shape.

    The library's launch functions neither allocate nor synchronise (include/slimmoe.h), so the single-rank eval forward captures
    whole (tests/test_gpu_model.py::test_eval_forward_captured_in_a_hip_graph_replays_bit_exact) and a replay computes the same
    bits.  What it buys depends on the model: ViT-B/16 at batch 256 is not launch-bound (12.75 vs 12.80 ms), but the reference's
    own DeiT-Tiny models (models/resMoE.py:151-209, batch 128) run ~150 kernels of 10-45 us per forward and the gaps between
    them are a third of the step -- 3.40 -> 2.26 ms (resmoe_tiny_patch16_224_expert8), 2.75 -> 2.09 ms (moe_tiny_...),
    profiles/r05_tiny_models.md.  One graph per (shape, dtype) of the input; the first call with a new shape runs two eager
    forwards (they fill every cache: 16-bit weight images, constant tables) and captures the third.  Host-side bookkeeping the
    captured kernels cannot do is replayed by hand: the token-skip gates' ``_total_tokens`` (the skipped-token counters live on
    the device and are updated by the captured kernels themselves).

    Expert-parallel models capture too when NOTHING of the forward needs the host: every MoE layer on the static exchange (ep.static_kind:
    a capacity gate, or ep.set_speculative's slots) with its collectives posted on the compute stream itself (ep.exchange_inline: one
    micro-batch, one chunk -- RCCL kernels inside a capture are fine, a cross-stream join around them is what hipStreamEndCapture
    dies on on this stack: tools/ep_graph_debug.py).  The graph holds the slot tables it was captured with: after every replay the
    routing histograms the captured kernels left on the device are queued for the overflow watch (ep.post_captured_stats), so an
    overflow voids the step exactly as it does eagerly (ep.run_guarded repeats it on the counted exchange -- which this object runs
    eagerly), and a re-sized table (overflow, or slots cut to the observed routing) re-captures.  Free the object before
    torch.distributed.destroy_process_group(): tearing down a communicator whose kernels sit in a live graph hangs."""

    def __init__(self, model: torch.nn.Module, autocast: bool = True):
        self.model, self.autocast, self.graphs = model, autocast, {}
        self.failed = None       # the exception of a capture that did not work: from then on every call runs eagerly
        self.captures = 0
        self._ep = self._ep_mods(model)

    @staticmethod
    def _ep_mods(model):
        return [m for m in model.modules() if hasattr(m, "ep_active") and m.ep_active()]

    @staticmethod
    def supported(model: torch.nn.Module, device, ep_graph: bool = True) -> bool:
        """``ep_graph``: whether an expert-parallel model may be captured at all (evaluate(): "auto" = a group of ONE rank only -- no
        run between distinct GPUs exists yet; hip_graph=True asks for it on any group).  The model must run one micro-batch
        (``model.ep_micro_batches = 1``; the class default, 2, keeps the exchanges on RCCL's stream for overlap)."""
        dev = torch.device(device)
        if dev.type != "cuda" or model.training:
            return False
        mods = GraphedForward._ep_mods(model)
        if not mods:
            return True
        from . import ep
        from .fmoe import default_compute_dtype
        if not ep_graph or int(getattr(model, "ep_micro_batches", 1)) > 1 or not ep.inline_possible(model):
            return False
        return all(ep.static_kind(m, m.compute_dtype or default_compute_dtype()) is not None for m in mods)

    def _ep_signature(self):
        """The slot tables the expert-parallel layers use right now (a graph is valid for exactly the tables it was captured with)."""
        return tuple((getattr(m, "ep_speculative", None), getattr(m, "ep_static_tokens", None),
                      tuple(st.table.serial for st in m.__dict__.get("_ep_slots", {}).values())) for m in self._ep)

    def _run(self, images):
        with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.float16, enabled=self.autocast):
            return self.model(images)

    def _gates(self):
        return [m for m in self.model.modules() if hasattr(m, "_total_tokens") and hasattr(m, "skip_counter")]

    def _capture(self, images):
        static_in = images.clone()
        side = torch.cuda.Stream(images.device)
        side.wait_stream(torch.cuda.current_stream(images.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                self._run(static_in)
        torch.cuda.current_stream(images.device).wait_stream(side)
        torch.cuda.synchronize(images.device)
        if self._ep:
            from . import ep
            ep.check_static_overflow(flush=True)     # the warm-up's routing: may re-size the slots (or raise: the caller repeats the step)
        gates = self._gates()
        before = [g._total_tokens for g in gates]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, **({"capture_error_mode": "relaxed"} if self._ep else {})):
            static_out = self._run(static_in)
        tokens = [(g, g._total_tokens - b) for g, b in zip(gates, before)]
        for g, b in zip(gates, before):       # the capture itself computed nothing: its host-side counts are taken back
            g._total_tokens = b
        self.captures += 1
        stats = ep.captured_stats(self.model) if self._ep else []
        return static_in, graph, static_out, tokens, self._ep_signature(), stats

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        if self.failed is not None:
            return self._run(images)
        if self._ep:
            from . import ep
            from .fmoe import default_compute_dtype
            if ep.static_kind(self._ep[0], self._ep[0].compute_dtype or default_compute_dtype()) is None:
                # the repeat of an overflowed step (ep.dynamic_only): on the counted exchange, eagerly
                return self._run(images)
        key = (tuple(images.shape), images.dtype, str(images.device))
        ent = self.graphs.get(key)
        if ent is not None and ent[4] != self._ep_signature():        # slots re-sized since: that graph's buffers have the old layout
            ent = None
            del self.graphs[key]
        if ent is None:
            gates = self._gates()
            before = [(g, g._total_tokens, g._skipped_tokens) for g in gates]   # the two warm-up forwards count for nothing
            try:
                ent = self.graphs[key] = self._capture(images)
            except Exception as exc:      # a model the capture cannot take (a host sync in a foreign module, ...): say so, run eagerly
                if self._ep and isinstance(exc, ep.StaticExchangeOverflow):
                    for g, t, s in before:
                        g._total_tokens, g._skipped_tokens = t, s
                    raise
                import warnings
                self.failed = exc
                torch.cuda.synchronize(images.device)
                warnings.warn(f"evaluate(): the forward could not be captured into a HIP graph ({type(exc).__name__}: {exc}); "
                              "running eagerly (hip_graph=False avoids the attempt)")
            for g, t, s in before:
                g._total_tokens, g._skipped_tokens = t, s
            if self.failed is not None:
                return self._run(images)
        static_in, graph, static_out, tokens, _, stats = ent
        static_in.copy_(images)
        graph.replay()
        for g, n in tokens:
            g._total_tokens += n
        if self._ep:
            ep.post_captured_stats(stats)          # the replay's routing goes to the overflow watch like an eager forward's
        return static_out.clone()        # (the graph's own output buffer is overwritten by the next replay)


def accuracy(output: torch.Tensor, target: torch.Tensor, topk=(1,)):
    """timm.utils.accuracy: top-k accuracy in percent."""
    maxk = min(max(topk), output.shape[1])
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[: min(k, maxk)].reshape(-1).float().sum(0) * 100.0 / target.shape[0] for k in topk]


@torch.no_grad()
def evaluate(data_loader: Iterable[Tuple[torch.Tensor, torch.Tensor]], model: torch.nn.Module, device,
             autocast: bool = True, *, ep_speculative="auto", hip_graph="auto"):
    """engine.py:88-121.  ``hip_graph`` (True / False / "auto" = on for single-rank GPU models unless SLIMMOE_EVAL_GRAPH=0): every
    batch shape's forward is captured once and replayed (GraphedForward: same bits, no launch gaps -- a third of the step of the
    reference's DeiT-Tiny models).  Under expert parallelism the harness owns the step, so it can run the exchange WITHOUT a host round trip
    per layer even for the reference's capacity-less NaiveGate: ``ep_speculative`` (alpha; "auto" = SLIMMOE_EP_ALPHA, default 1.5;
    None / 0 = off) sizes static slots of alpha x the balanced share (ep.set_speculative), and a batch whose routing does not fit
    -- reported by all ranks together -- is evaluated again on the counted exchange (ep.run_guarded): the metrics are those of the
    counted exchange either way.  Returns the reference's dict keys plus images/sec and the number of repeated steps."""
    from . import ep
    criterion = torch.nn.CrossEntropyLoss()
    model.eval()
    dev = torch.device(device)
    _size_static_exchange(model, data_loader)
    if any(m.ep_active() for m in ep._ep_modules(model)):
        ep.set_speculative(model, _speculative_alpha(ep_speculative))
    import os
    ep_graph = hip_graph is True or _ep_world_size(model) == 1     # (captured RCCL between distinct GPUs: only on request so far)
    if hip_graph == "auto":
        hip_graph = os.environ.get("SLIMMOE_EVAL_GRAPH", "1") != "0"
    graphed = GraphedForward(model, autocast) if (hip_graph and GraphedForward.supported(model, dev, ep_graph)) else None
    n, loss_sum, a1, a5, repeats = 0, 0.0, 0.0, 0.0, 0
    t0 = time.perf_counter()
    for images, target in data_loader:
        images = images.to(dev, non_blocking=True)
        target = target.to(dev, non_blocking=True)

        def step():
            with torch.autocast(device_type=dev.type, dtype=torch.float16, enabled=autocast and dev.type == "cuda"):
                output = graphed(images) if graphed is not None else model(images)
                return output, criterion(output, target)
        # (flush: this step's overflow report is read before its numbers are -- the .item() below waits for the batch anyway)
        (output, loss), again = ep.run_guarded(step, flush=True)
        repeats += int(again)
        acc1, acc5 = accuracy(output, target, topk=(1, 5))
        bs = images.shape[0]
        n += bs
        loss_sum += loss.item() * bs
        a1 += acc1.item() * bs
        a5 += acc5.item() * bs
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    n = max(n, 1)
    return {"loss": loss_sum / n, "acc1": a1 / n, "acc5": a5 / n, "images_per_sec": n / dt, "ep_repeated_steps": repeats,
            "hip_graph": graphed is not None}


class GraphedTrainStep:
    """One training step -- autocast forward, criterion (+ the gates' aux losses), ``optimizer.zero_grad``, loss scaling, backward,
    clipping, the optimizer step, the loss-scale update: engine.py:52-74 -- captured into ONE HIP graph per batch shape and replayed.

    Expert-parallel models too when their gates have a capacity (BASELINE cfg 5): on the static exchange with the collectives on the
    compute stream the step has no host round trip either (tools/train_bench.py: cfg 5's ViT-B step through the expert-parallel path
    on a one-rank group 30.2 ms eager -> 29.0 ms replayed).  Drop this object before the process group.
    Possible because nothing in that step touches the host when it runs on this package's ``optim.AdamW`` + ``optim.NativeScaler``
    (the non-finite check, the clip coefficient, the step count and the loss scale live on the device).  What it buys: the reference's
    own model (resmoe_tiny_patch16_224_expert8, batch 128: models/resMoE.py:151-187, cmd.sh:7-13) issues ~1,060 launches per step for
    11 ms of GPU work and took 21.5 ms eager -- the host's launch rate; replayed it takes **10.8 ms**.  ViT-B at batch 128 is GPU-bound
    (28 ms either way).

    The first WARM steps with a given batch shape run eagerly (they are ordinary steps on their own batches and fill every cache); the
    next one is captured -- recording only -- and replayed for that batch and every later one of the same shape.  Inputs are copied
    into the graph's static buffers.  Hyper-parameters that live on the host (lr, weight decay, betas: baked into the captured
    tables) are compared before every replay; a change re-captures (a per-epoch scheduler costs one capture per epoch; under a
    per-step scheduler the harness gives up and runs eagerly).  After the last replay -- ``finish()`` -- every parameter's version
    counter is bumped: the captured kernels refreshed the 16-bit weight images in place, but images that the FORWARD re-derives (the
    transposed dgrad operands) were made before the last update, and host-side version counters do not move under a replay.
    Host-side counters the kernels cannot keep (``Gate._total_tokens``) are advanced by hand."""

    WARM = 3

    def __init__(self, model, criterion, optimizer, loss_scaler, max_norm, with_inputs, aux_loss_weight, autocast):
        from .fmoe import FMoETransformerMLP
        self.model, self.criterion, self.optimizer, self.scaler = model, criterion, optimizer, loss_scaler
        self.max_norm, self.with_inputs, self.aux_w, self.autocast = max_norm, with_inputs, aux_loss_weight, autocast
        self.moes = [m for m in model.modules() if isinstance(m, FMoETransformerMLP)]
        self.gates = [m for m in model.modules() if hasattr(m, "_total_tokens") and hasattr(m, "skip_counter")]
        self.graphs, self.seen, self.recaptures, self.replayed = {}, {}, 0, False
        self.disabled = False

    @staticmethod
    def supported(model, optimizer, loss_scaler, device, model_ema, ep_graph: bool = False) -> bool:
        """``ep_graph``: whether an expert-parallel model may be captured (train_one_epoch: a group of ONE rank, or hip_graph=True).
        It can be when the step has no host round trip: every expert-parallel layer a CAPACITY gate on the static exchange (its slots
        cannot overflow and a fixed batch shape cannot outgrow the agreed row count) with the collectives on the compute stream."""
        from .optim import NativeScaler as _OwnScaler
        dev = torch.device(device)
        mods = [m for m in model.modules() if hasattr(m, "ep_active") and m.ep_active()]
        if mods:
            from . import ep
            from .fmoe import default_compute_dtype
            if not (ep_graph and ep.inline_possible(model, training=True) and not any(m._drop_p > 0 for m in mods)
                    and all(m.gate.capacity(1 << 20) >= 0 and ep.static_kind(m, m.compute_dtype or default_compute_dtype()) == "capacity"
                            for m in mods)):
                return False
        return (dev.type == "cuda" and model_ema is None and getattr(optimizer, "_slimmoe_refreshes_images", False)
                and isinstance(loss_scaler, _OwnScaler) and loss_scaler.enabled
                and not getattr(optimizer, "is_second_order", False))

    def _hyper(self):
        return tuple((g.get("lr"), g.get("weight_decay"), tuple(g.get("betas", ())), g.get("eps")) for g in self.optimizer.param_groups)

    def eager(self, samples, targets):
        """The step itself (also what gets captured).  Returns the detached f32 loss."""
        dev = samples.device
        with torch.autocast(device_type=dev.type, dtype=torch.float16, enabled=self.autocast and dev.type == "cuda"):
            outputs = self.model(samples)
            loss = self.criterion(samples, outputs, targets) if self.with_inputs else self.criterion(outputs, targets)
            if self.aux_w:
                auxes = [a for a in (m.gate.get_loss() for m in self.moes) if a is not None]
                if auxes:
                    loss = loss + self.aux_w * torch.stack([a.reshape(()) for a in auxes]).sum()
        lv = loss.detach().float()
        self.optimizer.zero_grad()
        self.scaler(loss, self.optimizer, clip_grad=self.max_norm, parameters=self.model.parameters(), create_graph=False)
        return lv

    def _capture(self, samples, targets):
        static_s, static_t = samples.clone(), targets.clone()
        if any(m.ep_active() for m in self.moes):
            from . import ep
            ep.check_static_overflow(flush=True)          # the eager steps' reports are read here, not by the captured forward
        before = [g._total_tokens for g in self.gates]
        self.optimizer.zero_grad(set_to_none=True)        # the backward's gradients are allocated from the graph's own pool
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="relaxed"):   # (relaxed: the step allocates small pinned staging buffers)
            lv = self.eager(static_s, static_t)
        tokens = [(g, g._total_tokens - b) for g, b in zip(self.gates, before)]
        for g, b in zip(self.gates, before):              # the capture computed nothing
            g._total_tokens = b
        return static_s, static_t, graph, lv, tokens, self._hyper()

    def __call__(self, samples, targets):
        if self.disabled:
            return self.eager(samples, targets)
        key = (tuple(samples.shape), samples.dtype, tuple(targets.shape), targets.dtype)
        self.seen[key] = self.seen.get(key, 0) + 1
        ent = self.graphs.get(key)
        if ent is not None and ent[5] != self._hyper():   # lr / weight decay moved: the captured tables are stale
            self.recaptures += 1
            if self.recaptures > 8 and self.recaptures > self.seen.get(key, 0) // 4:
                self.finish()                             # a per-step schedule: capturing every step costs more than it saves
                self.disabled = True
                return self.eager(samples, targets)
            ent = None
            self.graphs.pop(key, None)
        if ent is None:
            if self.seen[key] <= self.WARM:
                if self.replayed:
                    self.finish()                         # (a new shape after replays: the eager step must see current images)
                return self.eager(samples, targets)
            try:
                ent = self.graphs[key] = self._capture(samples, targets)
            except Exception as exc:      # (nothing ran on the GPU during the failed capture: the step is still to be done)
                import warnings
                torch.cuda.synchronize(samples.device)
                warnings.warn(f"train_one_epoch(): the step could not be captured into a HIP graph ({type(exc).__name__}: {exc}); "
                              "running eagerly")
                self.finish()
                self.disabled = True
                return self.eager(samples, targets)
        static_s, static_t, graph, lv, tokens, _ = ent
        static_s.copy_(samples)
        static_t.copy_(targets)
        graph.replay()
        self.replayed = True
        for g, n in tokens:
            g._total_tokens += n
        return lv.clone()

    def finish(self):
        """Call when the replays end (end of the epoch, or before an eager step): host-side version counters catch up."""
        if self.replayed:
            from .optim import _bump_versions
            _bump_versions([p for group in self.optimizer.param_groups for p in group["params"]])
            self.replayed = False


def _criterion_takes_inputs(criterion) -> bool:
    """The reference's criterion is ``DistillationLoss.forward(inputs, outputs, labels)`` (losses.py:28; called as
    ``criterion(samples, outputs, targets)`` at engine.py:54); a plain ``nn.CrossEntropyLoss`` takes (outputs, targets)."""
    import inspect
    fn = getattr(criterion, "forward", criterion)
    try:
        params = [p for p in inspect.signature(fn).parameters.values()
                  if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.default is p.empty]
    except (TypeError, ValueError):
        return False
    return len(params) >= 3


def train_one_epoch(model: torch.nn.Module, criterion, data_loader: Iterable[Tuple[torch.Tensor, torch.Tensor]],
                    optimizer: torch.optim.Optimizer, device, epoch: int, loss_scaler, max_norm=None, model_ema=None,
                    mixup_fn=None, set_training_mode=True, args=None, *, aux_loss_weight: float = 0.0, gate_delta=None,
                    autocast: bool = True, check_every: int = 50, hip_graph=False, ep_speculative="auto"):
    """The reference's training loop body (engine.py:22-85) around the HIP path, with the reference's positional
    parameters in the reference's order (main.py:825-838 calls it positionally: ``model_ema`` and ``mixup_fn`` sit in
    positions 9 and 10): autocast forward, criterion, ``loss_scaler(loss, optimizer, clip_grad=max_norm,
    parameters=model.parameters())`` (optim.NativeScaler keeps the whole step on the device), ``model_ema.update(model)``
    after every step and ``mixup_fn(samples, targets)`` before it when given, ``args.bce_loss`` as at engine.py:49-50.
    ``max_norm=None`` = no clipping (what main.py passes by default, ``--clip-grad None``).  A criterion whose forward takes
    three tensors is called like the reference's DistillationLoss, ``criterion(samples, outputs, targets)``.

    Keyword-only extensions: ``aux_loss_weight`` adds the MoE gates' load-balance losses (SwitchGate, BASELINE cfg 5);
    ``gate_delta`` runs the token-skip gates' threshold schedule (``Gate.step(delta)`` for every gate, as main.py:887-891
    does after the epoch's steps); ``autocast``; ``hip_graph`` (default False; True, or "auto" = SLIMMOE_TRAIN_GRAPH=1): the whole step
    replayed from one HIP graph per batch shape (GraphedTrainStep: needs this package's AdamW + NativeScaler, one rank, no EMA; 2 x on
    the reference's DeiT-Tiny model, nothing on ViT-B) -- anything it cannot take runs eagerly, as before; ``ep_speculative`` (under
    expert parallelism; alpha, "auto" = SLIMMOE_EP_ALPHA, None / 0 = off): gates WITHOUT a capacity (the reference's NaiveGate) train on
    the speculative static exchange -- no host round trip per layer; the forward's overflow report is read once, before the backward,
    and a forward whose routing did not fit its slots is repeated on the counted exchange by all ranks together (capacity gates are on
    the static exchange anyway: their slots cannot overflow).

    The reference aborts on a non-finite loss by reading ``loss.item()`` in every step (engine.py:56-60: one host sync per
    step), BEFORE the optimizer step and the EMA update.  Here the check is a device-side count read every ``check_every`` steps
    (default 50) and at the end of the epoch: the same abort (``SystemExit(1)`` after the message), at most ``check_every`` steps
    late.  What runs meanwhile is gated: an enabled ``optim.NativeScaler`` skips the update on a non-finite gradient by itself; with any
    other (or a disabled) scaler, or with a ``model_ema`` (whose update is a host call that cannot be skipped from the device), the loss is read
    in EVERY step, as the reference does, so no weight, EMA or checkpoint ever absorbs a non-finite step.  The metric logger is
    driver plumbing (out of scope).
    Returns {"loss": mean loss, "steps": n, "lr": first group's lr}."""
    from .fmoe import FMoETransformerMLP
    from .resmoe import Gate

    model.train(set_training_mode)
    dev = torch.device(device)
    moes = [m for m in model.modules() if isinstance(m, FMoETransformerMLP)]
    with_inputs = _criterion_takes_inputs(criterion)
    _size_static_exchange(model, data_loader)
    from .optim import NativeScaler as _OwnScaler
    every_step = model_ema is not None or not (isinstance(loss_scaler, _OwnScaler) and loss_scaler.enabled)
    if every_step:
        check_every = 1
    bce = bool(getattr(args, "bce_loss", False)) if args is not None else False
    loss_sum, bad, n = torch.zeros((), device=dev), torch.zeros((), device=dev), 0
    graphed = None
    ep_graph = hip_graph is True or _ep_world_size(model) == 1      # (captured RCCL between distinct GPUs: only on request so far)
    if hip_graph == "auto":
        import os
        hip_graph = os.environ.get("SLIMMOE_TRAIN_GRAPH", "0") == "1"
    if hip_graph and not every_step and GraphedTrainStep.supported(model, optimizer, loss_scaler, dev, model_ema, ep_graph):
        graphed = GraphedTrainStep(model, criterion, optimizer, loss_scaler, max_norm, with_inputs, aux_loss_weight, autocast)
    from . import ep as _ep
    ep_spec, ep_repeats = False, 0
    if any(m.ep_active() for m in _ep._ep_modules(model)):
        alpha = _speculative_alpha(ep_speculative)
        ep_spec = bool(alpha) and _ep.set_speculative(model, alpha, train=True) > 0
        if not alpha:
            _ep.set_speculative(model, None)
    try:
        for samples, targets in data_loader:
            samples = samples.to(dev, non_blocking=True)
            targets = targets.to(dev, non_blocking=True)
            if mixup_fn is not None:
                samples, targets = mixup_fn(samples, targets)
            if bce:
                targets = targets.gt(0.0).type(targets.dtype)
            if graphed is not None:
                # (an enabled optim.NativeScaler skips a non-finite step on the device by itself: the abort can wait for check_every)
                lv = graphed(samples, targets)
                finite = torch.isfinite(lv)
                loss_sum += torch.where(finite, lv, torch.zeros_like(lv))
                bad += (~finite).to(bad.dtype)
                n += 1
                if check_every > 1 and n % check_every == 0 and int(bad):
                    print(f"Loss is non-finite in {int(bad)} of {n} steps, stopping training")
                    raise SystemExit(1)
                continue
            def forward_loss():
                with torch.autocast(device_type=dev.type, dtype=torch.float16, enabled=autocast and dev.type == "cuda"):
                    outputs = model(samples)
                    loss = criterion(samples, outputs, targets) if with_inputs else criterion(outputs, targets)
                    if aux_loss_weight:
                        auxes = [a for a in (m.gate.get_loss() for m in moes) if a is not None]
                        if auxes:     # one stack + sum instead of two tiny kernels (and two backward nodes) per layer
                            loss = loss + aux_loss_weight * torch.stack([a.reshape(()) for a in auxes]).sum()
                return loss
            if ep_spec:
                # speculative static exchange: the forward's overflow report is read BEFORE the backward (one host sync per step where
                # the counted exchange has one per layer); a forward whose routing did not fit is dropped -- by every rank together -- and
                # run again on the counted exchange, so no gradient ever comes from a step that lost rows
                loss, again = _ep.run_guarded(forward_loss, flush=True)
                ep_repeats += int(again)
            else:
                loss = forward_loss()
            lv = loss.detach().float()
            finite = torch.isfinite(lv)
            if every_step and not bool(finite):           # the reference's order: abort before the step and the EMA update
                print(f"Loss is {float(lv)}, stopping training")
                raise SystemExit(1)
            optimizer.zero_grad()
            is_second_order = hasattr(optimizer, "is_second_order") and optimizer.is_second_order
            loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=is_second_order)
            if model_ema is not None:
                model_ema.update(model)
            loss_sum += torch.where(finite, lv, torch.zeros_like(lv))
            bad += (~finite).to(bad.dtype)
            n += 1
            if check_every > 1 and n % check_every == 0 and int(bad):
                print(f"Loss is non-finite in {int(bad)} of {n} steps, stopping training")
                raise SystemExit(1)
    finally:
        if ep_spec:      # outside this harness nobody repeats a forward that lost rows: training forwards go back to the counted exchange
            _ep.set_speculative(model, alpha, train=False)
    if graphed is not None:
        graphed.finish()
    if gate_delta is not None:
        for m in model.modules():
            if isinstance(m, Gate):
                m.step(gate_delta)
    from .ep import check_static_overflow
    check_static_overflow(flush=True)
    n_bad = int(bad)                          # the epoch's host read (with the mean below)
    if n_bad:
        print(f"Loss is non-finite in {n_bad} of {n} steps, stopping training")
        raise SystemExit(1)
    mean = float(loss_sum) / max(n, 1)
    return {"loss": mean, "steps": n, "lr": optimizer.param_groups[0]["lr"], "ep_repeated_steps": ep_repeats,
            "hip_graph_steps": 0 if graphed is None else sum(max(0, c - GraphedTrainStep.WARM) for c in graphed.seen.values())}
