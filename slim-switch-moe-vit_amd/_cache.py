"""Derived device tensors that are made once and reused (16-bit shadows of parameters, constant index tables).

Such a tensor is produced by a kernel on whichever stream first asks for it; a later consumer on ANOTHER stream
must not read it before that kernel has run.  Every entry therefore carries the event recorded behind its producer
and the stream it was produced on; ``get`` makes any other stream wait on that event (a no-op for the single-stream
case, which never leaves the producing stream).  Versioned entries are rebuilt when the source tensor changes
(``_version`` / ``data_ptr``); in-place updates through ``.data`` do not bump ``_version`` -- call ``invalidate()``
after those (the modules do it from a load_state_dict post-hook; optim.AdamW bumps the versions itself after its fused
step).  A tensor handed to a stream other than its producer's is also registered with the caching allocator
(``record_stream``), so dropping the entry cannot recycle memory that stream still reads."""
from __future__ import annotations

import weakref
from typing import Callable, Hashable

import torch

_ALL = weakref.WeakSet()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class StreamCache:
    def __init__(self):
        self._c = {}
        _ALL.add(self)

    def invalidate(self):
        self._c.clear()

    def get(self, key: Hashable, version, make: Callable[[], torch.Tensor]) -> torch.Tensor:
        hit = self._c.get(key)
        if hit is None or hit[0] != version:
            t = make()
            ev = sid = None
            if t.is_cuda:
                st = torch.cuda.current_stream(t.device)
                sid = _raw_stream(t.device.index) if _raw_stream is not None else st.cuda_stream
                if not torch.cuda.is_current_stream_capturing():   # (inside a HIP-graph capture everything is on the capturing
                    ev = torch.cuda.Event()                        #  stream; an event recorded there could not be waited for outside)
                    ev.record(st)
            hit = (version, t, ev, sid)
            self._c[key] = hit
            return t
        _, t, ev, sid = hit
        if ev is not None:
            # (the hit path runs a dozen times per layer: the raw handle first, a Stream object only when the stream really differs)
            if _raw_stream is not None and _raw_stream(t.device.index) == sid:
                return t
            st = torch.cuda.current_stream(t.device)
            if (_raw_stream(t.device.index) if _raw_stream is not None else st.cuda_stream) != sid:
                st.wait_event(ev)
                # the consumer's stream is not the one the caching allocator knows this tensor by: without this, a version
                # bump that drops the entry could hand its memory to a new allocation while this stream still reads it
                t.record_stream(st)
        return t


    def peek(self, key: Hashable, version):
        """The entry's tensor if it is current for ``version`` (no rebuild, no stream bookkeeping), else None."""
        hit = self._c.get(key)
        return hit[1] if hit is not None and hit[0] == version else None

    def refresh(self, key: Hashable, tensor: torch.Tensor, version) -> None:
        """``tensor`` (the entry's own, updated IN PLACE by a kernel enqueued on the current stream -- the optimizer's fused step
        writes the 16-bit weight images together with the weights) is now current for ``version``: re-stamp the entry with that
        version and an event behind the producing kernel."""
        hit = self._c.get(key)
        if hit is None or hit[1] is not tensor:
            return
        ev = sid = None
        if tensor.is_cuda:
            st = torch.cuda.current_stream(tensor.device)
            sid = _raw_stream(tensor.device.index) if _raw_stream is not None else st.cuda_stream
            if not torch.cuda.is_current_stream_capturing():
                ev = torch.cuda.Event()
                ev.record(st)
        self._c[key] = (version, tensor, ev, sid)


# parameter -> the 16-bit image a fused optimizer step may refresh in place: id(parameter) -> (weakref(parameter), cache, key).
# Registered by whoever builds the image (FMoELinear.weight_as, vit._HalfCache.get); a dead parameter's entry is dropped on lookup.
_SHADOWS = {}


def register_shadow(p: torch.Tensor, cache: "StreamCache", key: Hashable) -> None:
    _SHADOWS[id(p)] = (weakref.ref(p), cache, key)


def shadow_of(p: torch.Tensor):
    """(cache, key, image) of the plain 16-bit image of ``p`` that is CURRENT right now (same shape, contiguous), or None."""
    ent = _SHADOWS.get(id(p))
    if ent is None:
        return None
    ref, cache, key = ent
    if ref() is not p:
        _SHADOWS.pop(id(p), None)
        return None
    t = cache.peek(key, param_version(p))
    if (t is None or t.shape != p.shape or not t.is_contiguous() or t.device != p.device
            or t.dtype not in (torch.float16, torch.bfloat16)):
        return None
    return cache, key, t


def param_version(p: torch.Tensor):
    return (p._version, p.data_ptr(), p.dtype)


def invalidate_all():
    """Drop every derived tensor (after in-place parameter updates that bypass autograd's version counter)."""
    for c in list(_ALL):
        c.invalidate()
