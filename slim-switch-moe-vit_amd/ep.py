"""Expert parallelism for the MoE operator: count exchange + token all-to-all-v, pipelined against the
grouped GEMMs (replaces fmoe_cuda.expert_exchange / global_scatter / global_gather and FasterMoE's
smart schedule).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm, and an all-to-all maps 1:1 onto
the xGMI mesh (every peer pair has its own link).  Experts are partitioned contiguously: rank w owns
global experts [w*E_local, (w+1)*E_local).  Every rank routes its own tokens over all W*E_local experts;
its expert-sorted send buffer is therefore already grouped by destination rank.

Receive layout: rows arrive rank-major ([source rank][local expert][token]).  Instead of re-sorting them
expert-major (an extra HBM pass) the grouped GEMM takes one row group per (source rank, local expert) with
a group -> expert map (``group_expert`` in include/slimmoe.h).

Overlap: the local tokens can be cut into ``ep_chunks`` micro-batches; the all-to-all of chunk c+1 then runs on
RCCL's stream under the expert GEMMs of chunk c (and the return all-to-all of chunk c under the GEMMs of
chunk c+1).  Only ONE host sync per layer: every chunk's count matrix travels in a single small all-to-all.
Default is one chunk: measured on one GPU (bench.py --force-ep) every extra chunk costs ~0.3 ms per layer in
extra launches and in tile quantisation of the halved GEMMs, which the overlap only repays when the exchange
is slower than that.
The functions that only move data (exchange_counts, segment_table, all_to_all_rows) are device-agnostic and
are exercised on CPU with the gloo backend in tests/test_ep_gloo.py.
"""
from __future__ import annotations

import itertools
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from ._cache import StreamCache


class _DoneWork:
    """Stand-in for a collective's work handle when the exchange already completed synchronously."""

    def wait(self):
        return True


_ctx_cache = {}


def _cabi_context(group, device):
    """The library's own RCCL communicator for this group (SLIMMOE_EP_TRANSPORT=cabi), created once per (group, device)."""
    from .comm import ExchangeContext
    key = (id(group) if group is not None else 0, str(device))
    ctx = _ctx_cache.get(key)
    if ctx is None:
        ctx = _ctx_cache[key] = ExchangeContext.from_process_group(group, device)
    return ctx


class _CtxWork:
    """Work handle of an exchange on the library's communication stream: wait() fences torch's current stream against
    THIS exchange (its ticket), not against whatever the context posted since."""

    def __init__(self, ctx, ref):
        self.ctx, self.ref, self.ticket = ctx, ref, ctx.last_ticket()

    def wait(self):
        self.ctx.wait_stream(self.ref, self.ticket)
        return True


def exchange_inline(mod, n_chunks: int = 1) -> bool:
    """Whether this layer's exchanges go on the COMPUTE stream itself.  With one micro-batch and one chunk nothing runs beside an
    exchange, so the communication stream buys no overlap and costs a stream hand-over on either side of each of the 2 x depth
    collectives (~0.8 ms of 15.4 per forward on the one-GPU proxy).  SLIMMOE_EP_INLINE = auto (default) | 0 | 1."""
    mode = os.environ.get("SLIMMOE_EP_INLINE", "auto")
    if mode in ("0", "1"):
        return mode == "1"
    return n_chunks == 1 and max(1, int(getattr(mod, "ep_rows_div", 1))) == 1


def inline_possible(model: torch.nn.Module, training: bool = False) -> bool:
    """exchange_inline for every layer of the model's NEXT forward, from the model's own settings (``ep_rows_div`` on the modules is
    what the previous forward left there).  A training forward is never cut into micro-batches (vit._ep_pipeline_depth)."""
    mode = os.environ.get("SLIMMOE_EP_INLINE", "auto")
    if mode in ("0", "1"):
        return mode == "1"
    return training or int(getattr(model, "ep_micro_batches", 1)) <= 1


def _a2a(out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None, group=None, async_op: bool = False,
         inline: bool = False):
    """``dist.all_to_all_single`` on the group's own transport; ``inline`` = on the caller's stream (torch.distributed runs a
    collective called with async_op=False on the current stream; the C-ABI transport has SMOE_A2A_INLINE), returning a finished
    work handle where one was asked for.  RCCL ("nccl") moves device buffers directly; a gloo
    group cannot take CUDA tensors through all-to-all, so there the buffers are staged through the host -- slow, but
    it lets the complete multi-rank data path (routing, exchange layouts, group->expert GEMMs, the micro-batch
    pipeline) run as several processes on ONE GPU in tests/test_gpu_model.py."""
    if inp.is_cuda and os.environ.get("SLIMMOE_EP_TRANSPORT", "torch") == "cabi":
        # the C-ABI transport (include/slimmoe.h smoe_a2a_*): own communicator, own stream, event fences
        ctx = _cabi_context(group, inp.device)
        W = ctx.world_size
        rows_in = inp.reshape(inp.shape[0], -1) if inp.dim() > 1 else inp.reshape(-1, 1)
        rows_out = out.reshape(out.shape[0], -1) if out.dim() > 1 else out.reshape(-1, 1)
        if not rows_in.is_contiguous():
            rows_in = rows_in.contiguous()
        sr = in_splits if in_splits is not None else [rows_in.shape[0] // W] * W
        rr = out_splits if out_splits is not None else [rows_out.shape[0] // W] * W
        if inline:                                              # on the compute stream itself: no hand-over either side
            ctx.all_to_all_rows(rows_in, sr, rr, wait="inline", out=rows_out)
            return _DoneWork() if async_op else None
        ctx.all_to_all_rows(rows_in, sr, rr, wait=not async_op, out=rows_out)
        return _CtxWork(ctx, out) if async_op else None
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        h_in = inp.cpu()
        h_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, h_in, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(h_out)
        return _DoneWork() if async_op else None
    if inline:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group, async_op=False)
        return _DoneWork() if async_op else None
    return dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group,
                                  async_op=async_op)


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """Sum of ``t`` over the group's ranks, on the group's own transport (a gloo group cannot take CUDA tensors)."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        return h.to(t.device)
    dist.all_reduce(t, group=group)
    return t


class _PinnedPool:
    """Reusable pinned host buffers for the count read-back (hipHostMalloc per layer would cost more than the copy)."""

    def __init__(self):
        self.free = {}

    def take(self, shape, dtype) -> torch.Tensor:
        key = (tuple(shape), dtype)
        lst = self.free.setdefault(key, [])
        return lst.pop() if lst else torch.empty(shape, dtype=dtype, pin_memory=True)

    def give(self, t: torch.Tensor):
        self.free.setdefault((tuple(t.shape), t.dtype), []).append(t)


_pinned = _PinnedPool()

_offs_pool = {}     # (device, length) -> [ring of int32 buffers whose element 0 is 0, next index]
_OFFS_RING = 64     # more than the layers x micro-batches x chunks whose GEMMs can still be queued behind one another


def _offsets_buffer(length: int, device) -> torch.Tensor:
    """An int32 [length] buffer with element 0 == 0 for a received-groups offset table (cumsum fills [1:]).  Taken round-robin from
    a ring per (device, length): the consumer (the grouped GEMM) runs on the stream that filled it, and a buffer comes around again
    only after 63 later tables were built on that stream."""
    stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    key = (str(device), int(length), stream)     # per stream: a side stream's cumsum never rewrites a table another stream's GEMM reads
    ent = _offs_pool.get(key)
    if ent is None:
        ent = _offs_pool[key] = [[torch.zeros(length, dtype=torch.int32, device=device) for _ in range(_OFFS_RING)], 0]
    buf = ent[0][ent[1]]
    ent[1] = (ent[1] + 1) % _OFFS_RING
    return buf


class PendingCounts:
    """Count matrices on their way to the host.  ``finish()`` is the layer's only host sync; between
    ``exchange_counts_start`` and ``finish`` the caller may enqueue unrelated GPU work (another micro-batch) so
    that the GPU stays busy while the host waits."""

    def __init__(self, host: torch.Tensor, event, pooled: bool, dev: torch.Tensor):
        self.host, self.event, self.pooled, self.dev = host, event, pooled, dev

    def group_offsets(self, c: int) -> torch.Tensor:
        """int32 [W*E_local + 1] row offsets of chunk c's received groups ([w][e] order), built on the device
        from the received counts (no host -> device copy on the critical path)."""
        n = self.dev[1][:, c, :].reshape(-1)
        offs = _offsets_buffer(n.numel() + 1, n.device)          # element 0 is zero and stays zero: no clearing launch per layer
        torch.cumsum(n, 0, dtype=torch.int32, out=offs[1:])
        return offs

    def finish_rows(self) -> Tuple[List[List[int]], List[List[int]]]:
        """The host sync, then only what the all-to-all-v needs: (send_rows[c][w], recv_rows[c][w]) as Python ints -- one
        ``tolist`` and integer sums (a handful of numbers) instead of six small CPU tensor ops between the GPU going idle and
        the exchange's launch."""
        if self.event is not None:
            self.event.synchronize()
        both = self.host.tolist()                                   # [2][W][C][E_local]
        if self.pooled:
            _pinned.give(self.host)
            self.pooled = False
        W, C = len(both[0]), len(both[0][0])
        send = [[sum(both[0][w][c]) for w in range(W)] for c in range(C)]
        recv = [[sum(both[1][w][c]) for w in range(W)] for c in range(C)]
        return send, recv

    def finish(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.event is not None:
            self.event.synchronize()
        h = self.host.to(torch.int64)
        if self.pooled:
            _pinned.give(self.host)
        return h[0].permute(1, 0, 2).contiguous(), h[1].permute(1, 0, 2).contiguous()


def exchange_counts_start(counts_per_chunk: List[torch.Tensor], world_size: int, group=None) -> PendingCounts:
    """counts_per_chunk[c]: int32 [W*E_local] rows this rank routes to each GLOBAL expert from chunk c.
    One small all-to-all + one asynchronous device->host copy for all chunks; see ``exchange_counts``."""
    W, C = world_size, len(counts_per_chunk)
    c0 = counts_per_chunk[0]
    E_local = c0.numel() // W
    both = torch.empty((2, W, C, E_local), dtype=c0.dtype, device=c0.device)         # [0] = sent, [1] = received
    for c, cnt in enumerate(counts_per_chunk):                                        # row w of [0] goes to rank w
        both[0, :, c, :].copy_(cnt.reshape(W, E_local))
    _a2a(both[1], both[0], group=group)
    if not both.is_cuda:
        return PendingCounts(both, None, False, both)
    host = _pinned.take(both.shape, both.dtype)
    host.copy_(both, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return PendingCounts(host, ev, True, both)


def exchange_counts(counts_per_chunk: List[torch.Tensor], world_size: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns host int64 tensors (lec, gec), both [C, W, E_local]:
      lec[c, w, e] = rows of chunk c this rank sends to rank w's local expert e
      gec[c, w, e] = rows of chunk c rank w sends to this rank's local expert e."""
    return exchange_counts_start(counts_per_chunk, world_size, group).finish()


def segment_table(gec_c: torch.Tensor) -> Tuple[List[int], List[int]]:
    """gec_c: host [W, E_local] for one chunk.  Received rows are laid out [w][e][token]; returns
    (offsets of the W*E_local row groups in that order, local expert id of every group)."""
    W, E_local = gec_c.shape
    offs = [0]
    for n in gec_c.reshape(-1).tolist():
        offs.append(offs[-1] + int(n))
    return offs, [e for _ in range(W) for e in range(E_local)]


_gexp_cache = StreamCache()


def _group_expert_ids(W: int, E_local: int, device) -> torch.Tensor:
    """Local expert id of every received row group ([w][e] order) -- constant per (W, E_local), kept on the device."""
    return _gexp_cache.get((W, E_local, str(device)), 0,
                           lambda: torch.arange(E_local, dtype=torch.int32, device=device).repeat(W))


def all_to_all_rows(rows: torch.Tensor, send_rows: List[int], recv_rows: List[int], group=None, async_op: bool = False,
                    inline: bool = False):
    """all-to-all-v of whole rows: ``rows[:sum(send_rows)]`` is split by destination rank; returns
    (received [sum(recv_rows), d], work handle or None)."""
    n_send, n_recv = int(sum(send_rows)), int(sum(recv_rows))
    out = torch.empty((n_recv, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    work = _a2a(out, rows[:n_send], [int(v) for v in recv_rows], [int(v) for v in send_rows], group, async_op, inline)
    return out, work


def chunk_bounds(T: int, chunks: int) -> List[Tuple[int, int]]:
    chunks = max(1, min(chunks, T)) if T > 0 else 1
    return [((T * c) // chunks, (T * (c + 1)) // chunks) for c in range(chunks)]


def drain(gen):
    """Run a ``*_steps`` generator to completion, ignoring its yield points; returns its return value."""
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value


class StaticExchangeOverflow(RuntimeError):
    """A rank brought more rows than the static exchange buffers were agreed for, or -- speculative exchange of a gate without a
    capacity -- routed more rows to one expert than a slot holds.  Raised on EVERY rank of the group, at the same point of the
    program (check_static_overflow), never by one rank alone in front of a collective.  The outputs computed since the overflowing
    forward are void; the modules are re-sized when this is raised, and ``run_guarded`` repeats the step on the dynamic path."""


def _control_tensor(values, group, device) -> torch.Tensor:
    """A small int64 tensor for a control-plane collective on this group's transport (gloo cannot take CUDA tensors)."""
    on_cpu = (not torch.device(device).type == "cuda") or (dist.is_initialized() and dist.get_backend(group) == "gloo")
    return torch.tensor(values, dtype=torch.int64, device="cpu" if on_cpu else device)


def _rows_div(mod) -> int:
    """How many micro-batches the caller cuts the local batch into right now (vit.VisionTransformer sets ``ep_rows_div`` on every
    expert-parallel module before a pipelined forward -- shared configuration, the same on every rank): the static buffers are
    agreed, and sized, per micro-batch."""
    return max(1, int(getattr(mod, "ep_rows_div", 1)))


def static_slot_tokens(mod, T: int, device) -> int:
    """The row count T_slot the static exchange buffers of ``mod`` are sized for (every (source rank, expert) slot holds
    capacity(T_slot) rows).  It must be the same on every rank, whatever each rank's own batch is, and it must be known WITHOUT
    communication once the job runs -- so it is agreed ONCE per module (and per micro-batch divisor, ``ep_rows_div``), on its first
    expert-parallel forward, by a collective EVERY rank runs unconditionally (all ranks run the same layers in the same order, so
    it matches): an all-gather of (preset ``ep_static_tokens`` or -1, this rank's row count).  No preset anywhere: the largest row
    count wins.  The same preset everywhere (``set_static_tokens`` / ``mod.ep_static_tokens = n`` on every rank, e.g. from the
    loader's batch size; a preset counts rows of the WHOLE local batch -- under a divisor n it stands for ceil(preset / (n unit))
    unit rows, ``unit`` = ``ep_rows_unit``, the tokens per image): that value.  Presets that differ between ranks: every rank raises
    the same error.  Later batches must fit; one that does not is reported by ``check_static_overflow`` on all ranks together
    (never by the overflowing rank alone: that deadlocked its peers)."""
    div = _rows_div(mod)
    table = mod.__dict__.get("_ep_static_agreed")
    if table is None:
        table = mod.__dict__["_ep_static_agreed"] = {}
    st = table.get(div)
    if st is not None:
        return st
    have = getattr(mod, "ep_static_tokens", None)
    if have is not None and div > 1:
        unit = max(1, int(getattr(mod, "ep_rows_unit", 1)))
        have = -(-int(have) // (div * unit)) * unit
    if mod.world_size > 1:
        group = mod.moe_group
        mine = _control_tensor([-1 if have is None else int(have), int(T)], group, device)
        got = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(got, mine, group=group)
        presets = sorted({int(t[0]) for t in got})
        if presets == [-1]:
            agreed = max(max(int(t[1]) for t in got), 1)
        elif len(presets) == 1:
            agreed = max(presets[0], 1)
        else:
            raise RuntimeError(f"expert-parallel static exchange: `ep_static_tokens` differs between ranks ({presets}; -1 = not set): "
                               "set the same value on every rank's module, or on none")
    else:
        agreed = max(int(have) if have is not None else int(T), 1)
    if div == 1:
        mod.ep_static_tokens = agreed
    table[div] = agreed
    return agreed


def _ep_modules(model: torch.nn.Module):
    return [m for m in model.modules() if hasattr(m, "ep_active") and hasattr(m, "gate") and hasattr(m, "experts")]


def set_static_tokens(model: torch.nn.Module, rows: int, unit: int = 1) -> int:
    """Size the static exchange buffers of every expert-parallel MoE module of ``model`` for batches of up to ``rows`` token rows
    per rank (call it with the same value on every rank BEFORE the first forward -- engine.evaluate / train_one_epoch do, from
    the loader's batch size -- or again later, on every rank, to re-size).  ``unit`` = token rows per image (micro-batches are cut
    between images).  Returns the number of modules touched."""
    n = 0
    for m in _ep_modules(model):
        m.ep_static_tokens = int(rows)
        m.ep_rows_unit = int(unit)
        m.__dict__.pop("_ep_static_agreed", None)
        n += 1
    return n


# ---- speculative static exchange for gates WITHOUT a capacity (the reference's NaiveGate: models/resMoE.py:26 "use naive-gate") ----
# A NaiveGate puts no bound on an expert's share, so its exchange is sized by the routing: upstream (and the dynamic path here) reads
# the count matrix back to the host in every layer.  The speculative exchange sizes every (source rank, expert) slot for
# ceil(alpha * rows * k / E) rows -- alpha x the balanced share -- and runs the capacity gates' static machinery on it: fixed-size
# buffers, equal-split all-to-alls, the counts in-band, NO host round trip.  A routing that does not fit is detected on the device
# (the plan's pre-clamp counts travel in the headers), reported by every rank together (check_static_overflow), and the step is
# repeated on the dynamic path (run_guarded) -- so the results are those of the dynamic path: bit for bit when nothing overflows
# (same rows, same groups, same kernels), and after the repeat when something did.  It is OPT-IN per model (set_speculative):
# somebody has to repeat an overflowing step, and only a harness that owns the step can (engine.evaluate, bench.py).
def set_speculative(model: torch.nn.Module, alpha: Optional[float], train: bool = False) -> int:
    """Switch the speculative static exchange on (alpha >= 1: slot = ceil(alpha * rows * k / E)) or off (None) for every
    expert-parallel MoE module of ``model`` whose gate has no capacity.  Same value on every rank.  ``train``: also for forwards
    WITH autograd (engine.train_one_epoch: it reads the forward's overflow report before the backward and repeats a forward that
    lost rows); without it a training forward keeps the counted exchange.  Returns the modules touched."""
    n = 0
    for m in _ep_modules(model):
        if m.gate.capacity(1 << 20) < 0:
            m.ep_speculative = None if alpha is None else max(1.0, float(alpha))
            m.ep_speculative_train = bool(train) and alpha is not None
            n += 1
    return n


_FORCE_DYNAMIC = 0


class dynamic_only:
    """Context: every expert-parallel forward inside takes the dynamic (counted) exchange -- the repeat of a step whose routing
    overflowed the speculative slots.  Enter it on every rank at the same point of the program."""

    def __enter__(self):
        global _FORCE_DYNAMIC
        _FORCE_DYNAMIC += 1

    def __exit__(self, *a):
        global _FORCE_DYNAMIC
        _FORCE_DYNAMIC -= 1


def static_kind(mod, cd) -> Optional[str]:
    """None (dynamic: count read-back + all-to-all-v), "capacity" (a capacity gate's static exchange) or "speculative" (a gate
    without a capacity on alpha-sized slots).  The kinds issue DIFFERENT collectives, so the choice may only depend on what all
    ranks share -- the module's configuration and switches set on every rank alike -- never on a rank's own batch: <= 63 groups of
    the fused plan kernel and of the persistent GEMM's row ranges, 16-bit operands, the persistent GEMM, the fused GELU
    activation without dropout.  SLIMMOE_EP_STATIC=0 switches both off (A/B; set it on every rank)."""
    if _FORCE_DYNAMIC or os.environ.get("SLIMMOE_EP_STATIC", "1") == "0":
        return None
    g = mod.gate
    if not (g.tot_expert <= 63 and mod.gemm_variant in (9, 10, 11, 12, 13, 14) and cd in (torch.float16, torch.bfloat16)
            and mod.d_model % 64 == 0 and mod.d_hidden % 64 == 0 and mod._fused_gelu and not (mod._drop_p > 0 and mod.training)):
        return None
    if g.capacity(1 << 20) >= 0:
        return "capacity"
    if getattr(mod, "ep_speculative", None) is not None and (not mod.training or getattr(mod, "ep_speculative_train", False)):
        return "speculative"
    return None


def use_static_exchange(mod, cd) -> bool:
    return static_kind(mod, cd) is not None


def speculative_slot(mod, agreed: int) -> int:
    import math
    return max(1, int(math.ceil(float(mod.ep_speculative) * agreed * mod.top_k / mod.gate.tot_expert)))


def static_plan_fits(mod, agreed: int) -> bool:
    """The fused slot-plan kernel's table limit, evaluated on the AGREED row count (the same on every rank)."""
    return (-(-agreed * mod.top_k // 1024)) * mod.gate.tot_expert <= 8192


class _SlotTable:
    """Send layout of one static exchange: global expert e owns ``caps[e]`` payload rows + ONE header row of the send buffer,
    regions in expert order (so that the E_local regions of a destination rank are contiguous).  Host lists for the buffer shapes
    and the all-to-all splits, device tables for the plan / header kernels."""

    _serials = itertools.count(1)

    def __init__(self, caps, rank: int, E_local: int, device):
        self.serial = next(_SlotTable._serials)      # never re-used (an id() can be): what a captured forward is keyed on
        self.caps = [max(1, int(c)) for c in caps]
        E = len(self.caps)
        W = E // E_local
        base = [0]
        for c in self.caps:
            base.append(base[-1] + c + 1)
        self.rows = base[-1]                                             # rows of the send (and the returned) buffer
        self.in_splits = [base[(w + 1) * E_local] - base[w * E_local] for w in range(W)]   # rows this rank sends to rank w
        lb = [base[rank * E_local + i] - base[rank * E_local] for i in range(E_local + 1)]
        self.out_splits = [lb[-1]] * W                                   # every source sends this rank's experts' regions
        self.recv_rows = W * lb[-1]
        self.base_dev = torch.tensor(base, dtype=torch.int32, device=device)
        self.lbase_dev = torch.tensor(lb, dtype=torch.int32, device=device)


HEADROOM = float(os.environ.get("SLIMMOE_EP_HEADROOM", "1.12"))   # speculative slots after adaptation: ceil(HEADROOM x the largest group seen)
# ... but never less than SIGMA standard deviations of a count of that size above it: a group of n rows fluctuates by ~sqrt(n) from batch
# to batch, which is 1 % of the bench's 6,300-row groups and 6 % of the 288-row groups of cfg 4's model at 16 images (E = 32: with 768
# slots per forward at 1.12 x EVERY fresh batch overflowed one of them; tools/ep_static_soak.py)
SIGMA = float(os.environ.get("SLIMMOE_EP_SIGMA", "8"))
ADAPT_MIN_OBS = 2          # exchanges observed before a module's slots are cut to what its routing needs
SHRINK_RATIO = 1.08        # ... and only when that saves more than this factor of the buffer rows


class _SlotState:
    """Per (module, micro-batch divisor): the slot table in use and what the headers of its exchanges reported since it was made."""

    def __init__(self, mod, kind: str, agreed: int, caps, device):
        self.mod, self.kind, self.agreed, self.device = mod, kind, agreed, device
        self.alpha = getattr(mod, "ep_speculative", None)
        self.headroom = HEADROOM
        self.obs, self.n_obs = [0] * len(caps), 0
        self._install(caps)

    def _install(self, caps):
        mod = self.mod
        rank = dist.get_rank(mod.moe_group) if mod.world_size > 1 else 0
        self.table = _SlotTable(caps, rank, mod.num_expert, self.device)

    def fitted_caps(self):
        import math
        return [max(1, int(math.ceil(max(self.headroom * o, o + SIGMA * math.sqrt(o))))) for o in self.obs]


def _slot_state(mod, kind: str, agreed: int, device) -> "_SlotState":
    """The slot state of ``mod`` for the current micro-batch divisor; (re)made -- from shared values only: the gate's capacity or
    alpha x the balanced share of the AGREED row count -- when there is none, or the agreed row count / kind / alpha changed."""
    div = _rows_div(mod)
    table = mod.__dict__.setdefault("_ep_slots", {})
    st = table.get(div)
    alpha = getattr(mod, "ep_speculative", None)
    if st is None or st.agreed != agreed or st.kind != kind or (kind == "speculative" and st.alpha != alpha):
        E_tot = mod.gate.tot_expert
        cap = max(1, mod.gate.capacity(agreed)) if kind == "capacity" else speculative_slot(mod, agreed)
        st = table[div] = _SlotState(mod, kind, agreed, [cap] * E_tot, device)
    return st


# ---- overflow watch: every rank learns every rank's row count and routing histogram from the headers of the exchange; the check is
#      deferred so that it never makes the host wait for the GPU (the static path's point), and it is deterministic so that all
#      ranks raise -- and re-size -- together ------------------------------------------------------------------------------------
OVERFLOW_LAG = 32          # exchanges between posting a stats matrix and reading it on the host
_overflow_pending = []     # [(event | None, host int32 [W, 1 + E], state, caps of that exchange, div, pooled)]


def _watch_overflow(stats: torch.Tensor, st: "_SlotState") -> None:
    """``stats`` i32 [W, 1 + E] = (row count, pre-clamp count per global expert) of every source rank -- identical on all ranks."""
    mod = st.mod
    div = _rows_div(mod)
    if stats.is_cuda:
        if torch.cuda.is_current_stream_capturing():
            mod.__dict__["_ep_last_stats"] = (stats, st, list(st.table.caps))   # a captured forward: read after a replay
            return
        host = _pinned.take(stats.shape, stats.dtype)
        host.copy_(stats, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _overflow_pending.append((ev, host, st, st.table.caps, div, True))
    else:
        _overflow_pending.append((None, stats.clone(), st, st.table.caps, div, False))


def _judge(rows, st: "_SlotState", caps):
    """(largest row count, per-expert largest group, T overflow?, slot overflow?) of one stats matrix."""
    t_max = max((r[0] for r in rows), default=0)
    g_max = [max(r[1 + e] for r in rows) for e in range(len(caps))] if rows else [0] * len(caps)
    over_t = t_max > st.agreed
    over_s = st.kind == "speculative" and any(g > c for g, c in zip(g_max, caps))
    return t_max, g_max, over_t, over_s


def captured_overflow(model: torch.nn.Module) -> bool:
    """After replaying a captured (HIP graph) expert-parallel forward: did any module's last exchange overflow?  Reads the stats
    the captured kernels left on the device (a host sync)."""
    bad = False
    for m in _ep_modules(model):
        ent = m.__dict__.get("_ep_last_stats")
        if ent is not None:
            stats, st, caps = ent
            _, _, over_t, over_s = _judge(stats.cpu().tolist(), st, caps)
            bad |= over_t or over_s
    return bad


def captured_stats(model: torch.nn.Module):
    """Right after capturing an expert-parallel forward: [(module, stats matrix in the graph's memory, slot state, caps)] of THAT graph."""
    return [(m,) + tuple(m.__dict__["_ep_last_stats"]) for m in _ep_modules(model) if m.__dict__.get("_ep_last_stats") is not None]


def post_captured_stats(entries) -> None:
    """After replaying a captured expert-parallel forward (``entries`` = captured_stats() of that graph, or the model = its latest
    capture): queue the stats matrices the replay left on the device for the overflow watch, judged against the slot tables of the
    capture -- the same deferred, collective-free check an eager forward gets."""
    if isinstance(entries, torch.nn.Module):
        entries = captured_stats(entries)
    for m, stats, st, caps in entries:
        host = _pinned.take(stats.shape, stats.dtype)
        host.copy_(stats, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _overflow_pending.append((ev, host, st, caps, _rows_div(m), True))


def check_static_overflow(flush: bool = False) -> None:
    """Reads the stats matrices posted at least OVERFLOW_LAG exchanges ago (all of them with ``flush``: the harness calls that
    at the end of a step).  Every rank holds the SAME matrices and runs the same sequence of exchanges, so every rank takes the same
    decisions at the same call: (1) an exchange whose source brought more rows than agreed, or routed more rows to an expert than its
    speculative slot holds, overflowed -- EVERY module that overflowed among the matrices read is re-sized (row counts: the largest
    batch seen; slots: HEADROOM x the largest group seen, per expert) and StaticExchangeOverflow is raised, so that repeating the
    step (on every rank) works; the outputs computed since the overflowing forward are void: that rank dropped the rows that did not
    fit to keep its buffers' agreed shape.  (2) Without an overflow, a speculative module whose slots are more than SHRINK_RATIO x
    what its observed routing needs gets slots of HEADROOM x the largest group seen PER EXPERT (the first, uniform alpha-sized table
    knows nothing of the experts' shares): the all-to-all then carries ~HEADROOM x the routed rows instead of alpha x.  Slots are cut
    at ``flush`` calls only (step boundaries)."""
    due = []
    while _overflow_pending and (flush or len(_overflow_pending) > OVERFLOW_LAG):
        due.append(_overflow_pending.pop(0))
    if not due:
        return
    touched, overflowed, report = {}, {}, None
    for ev, host, st, caps, div, pooled in due:
        if ev is not None:
            ev.synchronize()
        rows = host.tolist()
        if pooled:
            _pinned.give(host)
        t_max, g_max, over_t, over_s = _judge(rows, st, caps)
        st.obs = [max(a, b) for a, b in zip(st.obs, g_max)]
        st.n_obs += 1
        touched[id(st)] = (st, div)
        if over_t or over_s:
            prev = overflowed.get(id(st), (st, div, 0))
            overflowed[id(st)] = (st, div, max(prev[2], t_max))
            if report is None:
                report = (f"the ranks brought {[r[0] for r in rows]} rows, the buffers were agreed for {st.agreed}" if over_t else
                          f"the routing put up to {g_max} rows (per expert, from one source) into slots of {list(caps)}")
    if report is None:
        # (2) cut a speculative module's slots to its routing -- only at a step boundary (``flush``), where every micro-batch of the
        # step has reported: the deferred check in the middle of a forward has read SOME micro-batches' matrices of a layer only, and
        # slots fitted to one micro-batch's routing overflow under its sibling's (E = 16, three micro-batches: bench.py's grid)
        for st, div in (touched.values() if flush else ()):
            if st.kind == "speculative" and st.n_obs >= ADAPT_MIN_OBS:
                fit = st.fitted_caps()
                if sum(st.table.caps) > SHRINK_RATIO * sum(fit):
                    st._install(fit)
                    st.obs, st.n_obs = [0] * len(fit), 0
        return
    # every later matrix is void as well (computed on dropped rows): return their buffers, forget them
    for ev, host, *_rest, pooled in _overflow_pending:
        if pooled:
            if ev is not None:
                ev.synchronize()
            _pinned.give(host)
    _overflow_pending.clear()
    for st, div, t_max in overflowed.values():
        mod = st.mod
        if t_max > st.agreed:
            table = mod.__dict__.setdefault("_ep_static_agreed", {})
            table[div] = max(t_max, table.get(div, 0))
            if div == 1:
                mod.ep_static_tokens = table[div]
            if st.kind == "speculative":          # the next _slot_state call re-makes the table for the new row count; keep what
                st.obs = [0] * len(st.obs)        # was learnt out of it (the observed groups belong to a dropped batch)
        elif st.kind == "speculative":
            st.headroom = min(2.0, st.headroom * 1.10) if st.n_obs > 1 else st.headroom
            st._install([max(c, f) for c, f in zip(st.table.caps, st.fitted_caps())])
    raise StaticExchangeOverflow(
        f"expert-parallel static exchange: {report}; the outputs since then are void.  {len(overflowed)} module(s) re-sized: "
        "repeat the step on every rank (ep.run_guarded does, on the counted exchange), or size the buffers up front with "
        "ep.set_static_tokens(model, rows) / a larger alpha in ep.set_speculative(model, alpha)")


def run_guarded(fn, flush: bool = True):
    """``fn()`` = one step's forward (every rank calls this at the same point).  An overflow of the static exchange buffers --
    reported by all ranks together -- voids the step: it is repeated ONCE on the counted exchange (results identical to what the
    static path gives when everything fits).  ``flush``: read this step's stats before returning (a host sync at the end of the
    step -- the harness reads the loss there anyway); without it an overflow surfaces up to OVERFLOW_LAG exchanges later and this
    call repeats the step it surfaces in.  Returns (result, repeated: bool)."""
    try:
        out = fn()
        check_static_overflow(flush=flush)
        return out, False
    except StaticExchangeOverflow:
        with dynamic_only():
            return fn(), True


def exchange_counts_static(counts: torch.Tensor, T: int, W: int, group=None):
    """A count exchange device to device (kept for the C-ABI transport's tests; the static forward carries its counts in-band,
    _ep_forward_static): rank r sends [E_local counts for peer w | its own row count T] to every peer w.  Returns (recv_counts i32
    [W * E_local] in [source rank][local expert] order, peer_rows i32 [W] = every rank's row count -- identical on all ranks)."""
    E_local = counts.numel() // W
    both = torch.empty((2, W, E_local + 1), dtype=torch.int32, device=counts.device)
    both[0, :, :E_local].copy_(counts.view(W, E_local))
    both[0, :, E_local].fill_(int(T))
    _a2a(both[1], both[0], group=group)
    return both[1, :, :E_local].reshape(-1), both[1, :, E_local]


def _ep_forward_static(mod, x, src, idx, plan_idx, score, probs, cd, residual, next_norm, agreed: int, kind: str):
    """Expert-parallel forward on STATIC buffers ("cfg 5 (capacity-bounded) can use fixed-size padded
    buffers -> no host sync"; Appendix B's `cap` note) -- for a capacity gate (``kind`` "capacity": a rank keeps at most `cap` of
    its rows per global expert) and, speculatively, for a gate without one ("speculative": slots of alpha x the balanced share,
    cut per expert to HEADROOM x what the routing needs once it has been observed).  Every global expert owns a fixed region of the
    send buffer -- its slot's payload rows plus ONE header row (_SlotTable) -- so both all-to-alls have splits known without
    looking at the routing, the counts travel in the header rows (smoe_ep_pack_headers / _unpack_headers: no count collective),
    and nothing of the layer waits for the host: the received counts stay on the device, where the grouped GEMM takes them as the
    end of each group's row range (group_end) and never schedules a tile over padding.  Two collectives per layer.  Yields at the
    two exchanges (micro-batch pipelining).

    A rank with NO rows takes the same path with empty slots.  A rank with MORE rows than agreed, or a speculative group larger
    than its slot, keeps the agreed buffer shape (it drops what does not fit; its peers must not hang) and the violation is raised
    on every rank by check_static_overflow -- never here, by this rank alone."""
    from . import ops
    from .fmoe import SwitchGate

    g = mod.gate
    W, E_local, k, d = mod.world_size, mod.num_expert, mod.top_k, mod.d_model
    group = mod.moe_group
    T = x.shape[0]
    E_tot = g.tot_expert
    dev = x.device
    st = _slot_state(mod, kind, agreed, dev)
    tab = st.table
    counts = raw = None
    if T > 0:
        cap = g.capacity(T) if kind == "capacity" else -1               # (on top of the slots: what THIS batch may keep)
        if T <= agreed:
            counts, offsets, gend, pos, inv_pos, pruned, raw = ops.dispatch_plan_slots(plan_idx, E_tot, tab.base_dev, tab.rows, cap)
        else:
            # over the agreed size: the plan over the first `agreed` rows only (the fused plan kernel's table is sized for that)
            counts, offsets, gend, pos, inv_h, pruned_h, raw = ops.dispatch_plan_slots(plan_idx[:agreed].contiguous(), E_tot,
                                                                                       tab.base_dev, tab.rows, cap)
            inv_pos = torch.full((T * k,), -1, dtype=torch.int64, device=dev)
            inv_pos[: agreed * k].copy_(inv_h)
            pruned = torch.full((T * k,), -1, dtype=torch.int64, device=dev)
            pruned[: agreed * k].copy_(pruned_h)
        send = ops.scatter_rows(src, pos, k, cd)                        # [tab.rows, d]; unused slots stay unwritten
        mod.last_plan = (idx, score, counts, offsets, pos, inv_pos)
        if isinstance(g, SwitchGate):
            from .autograd import switch_aux_loss
            g.set_loss(switch_aux_loss(pruned, probs, E_tot))
    else:
        inv_pos = torch.empty((0,), dtype=torch.int64, device=dev)
        send = torch.empty((tab.rows, d), dtype=cd, device=dev)
        mod.last_plan = (idx, score, torch.zeros(E_tot, dtype=torch.int32, device=dev),
                         torch.zeros(E_tot + 1, dtype=torch.int32, device=dev), None, inv_pos)
    # the counts, this rank's row count and its whole routing histogram ride in the header rows; nobody on the host reads them
    ops.ep_pack_headers(send, counts, raw, tab.base_dev, T)
    recv = torch.empty((tab.recv_rows, d), dtype=cd, device=dev)
    inline = exchange_inline(mod)
    work = _a2a(recv, send, tab.out_splits, tab.in_splits, group, async_op=True, inline=inline)
    yield                                                              # dispatch all-to-all in flight
    if work is not None:
        work.wait()
    # group l = (source rank, local expert): rows [starts[l], ends[l]) of the received buffer
    starts, ends, stats = ops.ep_unpack_headers(recv, W, tab.lbase_dev, E_tot)
    _watch_overflow(stats, st)
    gexp = _group_expert_ids(W, E_local, dev)
    y = mod._experts_fwd(recv, starts, cd, out_dtype=cd, group_expert=gexp, group_end=ends,
                         rows_hint=(T if T > 0 else agreed) * k)
    back = torch.empty((tab.rows, d), dtype=cd, device=dev)
    work2 = _a2a(back, y, tab.in_splits, tab.out_splits, group, async_op=True, inline=inline)
    yield                                                              # return all-to-all in flight
    if work2 is not None:
        work2.wait()
    if T == 0:
        return torch.empty((0, d), dtype=x.dtype, device=dev)
    return _combine_maybe_ln(back, inv_pos, score, T, k, x, residual, next_norm)


def ep_forward(mod, x: torch.Tensor, cd: torch.dtype, residual: Optional[torch.Tensor] = None,
               norm: Optional[torch.nn.Module] = None) -> torch.Tensor:
    """Expert-parallel FMoETransformerMLP forward for this rank's tokens x [T, d] -> [T, d] (see ep_forward_steps)."""
    return drain(ep_forward_steps(mod, x, cd, residual, norm))


def _ln_fusable(next_norm, x, payload_dtype, k) -> bool:
    d = x.shape[1]
    return (next_norm is not None and x.dtype == torch.float32 and payload_dtype in (torch.float16, torch.bfloat16) and k <= 4
            and d % 8 == 0 and d <= 1024 and next_norm.bias is not None)


def _combine_maybe_ln(back, inv_pos, score, T, k, x, residual, next_norm):
    """The return side's last kernel: gather + combine (+ residual); with ``next_norm`` also that LayerNorm of the produced rows
    (16 bit) in the same pass -> (out, xn) instead of out."""
    from . import ops
    d = x.shape[1]
    if _ln_fusable(next_norm, x, back.dtype, k):
        return ops.gather_combine_ln(back, inv_pos, score, T, k, residual, next_norm.weight.detach().float(),
                                     next_norm.bias.detach().float(), next_norm.eps, torch.float16)
    out = torch.empty((T, d), dtype=x.dtype, device=x.device)
    ops.gather_combine(back, inv_pos, score, T, k, x.dtype, out=out, residual=residual)
    return out


def ep_forward_steps(mod, x: torch.Tensor, cd: torch.dtype, residual: Optional[torch.Tensor] = None,
                     norm: Optional[torch.nn.Module] = None, next_norm: Optional[torch.nn.Module] = None,
                     routed: Optional[dict] = None):
    """Generator form of the expert-parallel forward: ``yield``s wherever this micro-batch has to wait for something
    that is not GPU compute -- (1) the count matrices reaching the host, (2) the dispatch all-to-all, (3) the return
    all-to-all -- so that a caller interleaving several micro-batches (vit.VisionTransformer) keeps the compute
    stream fed with the other micro-batch's attention / GEMMs meanwhile.  Every rank yields at the same points in
    the same order whatever the routing, so the collectives stay matched.  Returns the output via StopIteration.

    With ``norm`` (a LayerNorm whose shape the fused kernel covers) the operator computes ``moe(norm(x))``: LayerNorm
    and router run as one pass over x and the send buffers are gathered from the normalised 16-bit image.  With ``routed`` (the
    residual-MoE block's gated half, FMoETransformerMLP.forward_norm_gate_add: LayerNorm, token-skip gate and router already ran as
    ONE pass) the routing is taken as given: ``idx`` / ``score`` [T, k], ``idx_plan`` (= idx, -1 for the tokens the skip gate
    masked: they are simply not sent), ``src`` = the 16-bit operand image the send buffers are gathered from."""
    from . import ops
    from .fmoe import SwitchGate

    g = mod.gate
    W, E_local, k, d = mod.world_size, mod.num_expert, mod.top_k, mod.d_model
    group = mod.moe_group
    T = x.shape[0]
    cap = g.capacity(T)
    check_static_overflow()      # (deferred, deterministic: every rank reads the same row-count vectors at the same call)
    # capacity is defined over the whole local batch, so dropping gates run un-chunked
    n_chunks = 1 if cap >= 0 else max(1, int(getattr(mod, "ep_chunks", 1)))
    bounds = chunk_bounds(T, n_chunks) if T > 0 else [(0, 0)] * n_chunks
    if len(bounds) < n_chunks:  # tiny batches: keep the collective count identical on every rank
        bounds = bounds + [(T, T)] * (n_chunks - len(bounds))

    noise = g.make_noise(T, x.device) if isinstance(g, SwitchGate) else None
    gw = g.gate.weight.detach().float().contiguous()
    gb = g.gate.bias.detach().float() if g.gate.bias is not None else None
    src = x  # rows the send buffers are gathered from
    plan_idx = None
    if routed is not None:
        idx, score, probs, src = routed["idx"], routed["score"], None, routed["src"]
        plan_idx = routed.get("idx_plan")
    elif T == 0:   # a rank without rows still takes part in every collective of the layer
        idx = torch.empty((0, k), dtype=torch.int64, device=x.device)
        score = torch.empty((0, k), dtype=torch.float32, device=x.device)
        probs = torch.empty((0, g.tot_expert), dtype=torch.float32, device=x.device)
        src = torch.empty((0, d), dtype=cd, device=x.device) if norm is not None else x
    elif norm is not None:
        xn16, _, idx, score, _, probs = ops.ln_router_topk(
            x, norm.weight.detach().float(), norm.bias.detach().float() if norm.bias is not None else None, norm.eps,
            gw, gb, k, g.kind, noise, xn16_dtype=cd, want_probs=isinstance(g, SwitchGate))
        src = xn16
    else:
        idx, score, _, probs = ops.router_topk(x, gw, gb, k, g.kind, noise, want_probs=isinstance(g, SwitchGate))
    if plan_idx is None:
        plan_idx = idx
    kind = static_kind(mod, cd)
    if kind is not None:
        # decided from the configuration and the AGREED row count only: every rank takes the same branch whatever its batch
        agreed = static_slot_tokens(mod, T, x.device)
        if static_plan_fits(mod, agreed):
            return (yield from _ep_forward_static(mod, x, src, idx, plan_idx, score, probs, cd, residual, next_norm, agreed, kind))
    plans = []
    for (t0, t1) in bounds:
        plans.append(ops.dispatch_plan(plan_idx[t0:t1], g.tot_expert, cap))
    mod.last_plan = (idx, score) + tuple(plans[0][:4])
    if isinstance(g, SwitchGate):
        from .autograd import switch_aux_loss
        pruned = plans[0][4] if plans[0][4] is not None else idx
        g.set_loss(switch_aux_loss(pruned, probs, g.tot_expert))

    pending = exchange_counts_start([p[0] for p in plans], W, group)
    # Nothing below depends on the HOST knowing the counts: the send buffers' scatter, the received groups' row offsets (built on
    # the device from the received counts) and the output buffer are all enqueued / allocated before the host goes to wait, so
    # that after the wait only the exchange's own launch stands between the idle GPU and its next work
    sends = [ops.scatter_rows(src[t0:t1], plans[c][2], k, cd) for c, (t0, t1) in enumerate(bounds)]
    offs_dev = [pending.group_offsets(c) for c in range(len(bounds))]
    gexp_dev = _group_expert_ids(W, E_local, x.device)
    out = None
    if not (next_norm is not None and len(bounds) == 1):
        out = torch.empty((T, d), dtype=x.dtype, device=x.device)
    yield                                                                # (1) counts in flight to the host
    send_rows_c, recv_rows_c = pending.finish_rows()                     # the only host sync: [C][W] Python ints

    # stage A: dispatch all-to-all (async; chunk c+1 travels under chunk c's GEMMs)
    inline = exchange_inline(mod, len(bounds))
    inflight = []
    for c, (t0, t1) in enumerate(bounds):
        send = sends[c]
        send_rows, recv_rows = send_rows_c[c], recv_rows_c[c]
        recv, work = all_to_all_rows(send, send_rows, recv_rows, group, async_op=True, inline=inline)
        inflight.append((send, recv, work, send_rows, recv_rows))
    yield                                                                # (2) dispatch all-to-all in flight
    # stage B: expert FFN on the received rows + return all-to-all (async)
    returning = []
    for c in range(len(bounds)):
        send, recv, work, send_rows, recv_rows = inflight[c]
        work.wait()
        n_recv = recv.shape[0]
        if n_recv > 0:
            y = mod._experts_fwd(recv, offs_dev[c], cd, out_dtype=cd, group_expert=gexp_dev)
        else:
            y = recv
        back, work2 = all_to_all_rows(y, recv_rows, send_rows, group, async_op=True, inline=inline)
        returning.append((y, back, work2))
    yield                                                                # (3) return all-to-all in flight
    # stage C: gather + combine in sender order
    if next_norm is not None and len(bounds) == 1 and T > 0 and returning[0][1].shape[0] > 0:
        y, back, work2 = returning[0]
        work2.wait()
        return _combine_maybe_ln(back, plans[0][3], score, T, k, x, residual, next_norm)
    if out is None:
        out = torch.empty((T, d), dtype=x.dtype, device=x.device)
    # several chunks: the next block's LayerNorm still rides on every chunk's combine (rows are independent) when every chunk
    # brought rows back -- a property of the routing, known on the host from the counts
    fuse_ln = (_ln_fusable(next_norm, x, cd, k) and T > 0 and all(r[1].shape[0] > 0 for r in returning)
               and all(t1 > t0 for (t0, t1) in bounds))
    xn = torch.empty((T, d), dtype=torch.float16, device=x.device) if fuse_ln else None
    for c, (t0, t1) in enumerate(bounds):
        y, back, work2 = returning[c]
        work2.wait()
        if t1 > t0:
            inv_pos = plans[c][3]
            if fuse_ln:
                ops.gather_combine_ln(back, inv_pos, score[t0:t1], t1 - t0, k, None if residual is None else residual[t0:t1],
                                      next_norm.weight.detach().float(), next_norm.bias.detach().float(), next_norm.eps,
                                      torch.float16, out=out[t0:t1], xn=xn[t0:t1])
            elif back.shape[0] == 0:  # every entry of the chunk was dropped
                if residual is not None:
                    out[t0:t1].copy_(residual[t0:t1])
                else:
                    out[t0:t1].zero_()
            else:
                ops.gather_combine(back, inv_pos, score[t0:t1], t1 - t0, k, x.dtype, out=out[t0:t1],
                                   residual=None if residual is None else residual[t0:t1])
    return (out, xn) if fuse_ln else out
