"""Eval harness mirroring the reference's engine.evaluate (engine.py:88-121): eval mode, no_grad,
autocast on the GPU, cross-entropy + top-1/top-5, per-batch loop.  Returns the same dict keys plus
images/sec.  MetricLogger / distributed meters are out of scope."""
from __future__ import annotations

import time
from typing import Iterable, Tuple

import torch


def accuracy(output: torch.Tensor, target: torch.Tensor, topk=(1,)):
    """timm.utils.accuracy: top-k accuracy in percent."""
    maxk = min(max(topk), output.shape[1])
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[: min(k, maxk)].reshape(-1).float().sum(0) * 100.0 / target.shape[0] for k in topk]


@torch.no_grad()
def evaluate(data_loader: Iterable[Tuple[torch.Tensor, torch.Tensor]], model: torch.nn.Module, device,
             autocast: bool = True):
    criterion = torch.nn.CrossEntropyLoss()
    model.eval()
    dev = torch.device(device)
    n, loss_sum, a1, a5 = 0, 0.0, 0.0, 0.0
    t0 = time.perf_counter()
    for images, target in data_loader:
        images = images.to(dev, non_blocking=True)
        target = target.to(dev, non_blocking=True)
        with torch.autocast(device_type=dev.type, dtype=torch.float16, enabled=autocast and dev.type == "cuda"):
            output = model(images)
            loss = criterion(output, target)
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
    return {"loss": loss_sum / n, "acc1": a1 / n, "acc5": a5 / n, "images_per_sec": n / dt}


def train_one_epoch(model: torch.nn.Module, criterion, data_loader: Iterable[Tuple[torch.Tensor, torch.Tensor]],
                    optimizer: torch.optim.Optimizer, device, epoch: int, loss_scaler, max_norm=None,
                    aux_loss_weight: float = 0.0, gate_delta=None, autocast: bool = True):
    """The reference's training loop body (engine.py:21-84) around the HIP path: autocast forward, criterion,
    ``loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters())`` (optim.NativeScaler keeps the
    whole step on the device), then the token-skip gates' threshold schedule (``Gate.step(delta)`` for every gate, as
    main.py:812-815 does after the epoch's steps).  ``criterion(outputs, targets)`` -- the reference's distillation
    criterion also takes the inputs (losses.py, off the hot path).  ``aux_loss_weight`` adds the MoE gates' load-balance
    losses (SwitchGate, BASELINE cfg 5).  Mixup, EMA and the metric logger are training-driver plumbing (out of scope).
    Returns {"loss": mean loss, "steps": n}."""
    from .fmoe import FMoETransformerMLP
    from .resmoe import Gate

    model.train(True)
    dev = torch.device(device)
    moes = [m for m in model.modules() if isinstance(m, FMoETransformerMLP)]
    loss_sum, n = torch.zeros((), device=dev), 0
    for samples, targets in data_loader:
        samples = samples.to(dev, non_blocking=True)
        targets = targets.to(dev, non_blocking=True)
        with torch.autocast(device_type=dev.type, dtype=torch.float16, enabled=autocast and dev.type == "cuda"):
            outputs = model(samples)
            loss = criterion(outputs, targets)
            if aux_loss_weight:
                for m in moes:
                    aux = m.gate.get_loss()
                    if aux is not None:
                        loss = loss + aux_loss_weight * aux
        optimizer.zero_grad()
        loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=False)
        loss_sum += loss.detach().float()
        n += 1
    if gate_delta is not None:
        for m in model.modules():
            if isinstance(m, Gate):
                m.step(gate_delta)
    mean = float(loss_sum) / max(n, 1)   # the epoch's single host read
    return {"loss": mean, "steps": n}
