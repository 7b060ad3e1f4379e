"""Training-side pieces of the MoE operator (BASELINE cfg 5): aux load-balance loss and backward."""
from __future__ import annotations

import torch


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


def moe_forward_train(module, inp):
    raise NotImplementedError("MoE backward is not built yet in this round; run under torch.no_grad()")
