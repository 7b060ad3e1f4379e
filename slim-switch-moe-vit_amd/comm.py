"""Python face of the C-ABI expert exchange (include/slimmoe.h: smoe_ctx_* / smoe_a2a_*): an RCCL communicator of the
library's own, on a dedicated communication stream with event fences against torch's current stream.

``ExchangeContext.from_process_group(group)`` bootstraps one from an existing ``torch.distributed`` group (rank 0's
unique id travels through a broadcast on that group); ``all_to_all_rows`` / ``exchange_counts`` are the two collectives of
an expert-parallel MoE layer.  ep.py uses this transport when SLIMMOE_EP_TRANSPORT=cabi; the default
stays ``torch.distributed.all_to_all_single`` on the "nccl" (= RCCL) backend."""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from . import _lib, ops


class ExchangeContext:
    def __init__(self, unique_id: bytes, world_size: int, rank: int, device: torch.device):
        lib = _lib.load()
        if len(unique_id) != lib.smoe_unique_id_bytes():
            raise ValueError("unique_id: wrong length")
        self.world_size, self.rank, self.device = world_size, rank, torch.device(device)
        self._h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, len(unique_id))
        with torch.cuda.device(self.device):
            _lib.check(lib.smoe_ctx_create(buf, world_size, rank, ctypes.byref(self._h)), "smoe_ctx_create")

    @staticmethod
    def new_unique_id() -> bytes:
        lib = _lib.load()
        buf = ctypes.create_string_buffer(lib.smoe_unique_id_bytes())
        _lib.check(lib.smoe_unique_id(buf), "smoe_unique_id")
        return bytes(buf.raw)

    @classmethod
    def from_process_group(cls, group=None, device: Optional[torch.device] = None) -> "ExchangeContext":
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        n = _lib.load().smoe_unique_id_bytes()
        box = [cls.new_unique_id() if rank == 0 else bytes(n)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], world, rank, device)

    def close(self):
        if self._h:
            _lib.load().smoe_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exchange_counts(self, send_counts: torch.Tensor, E_local: int, wait: bool = True) -> torch.Tensor:
        """send_counts i32 [W * E_local] -> received counts, same shape ([source rank][local expert])."""
        ops._chk(send_counts, "send_counts", torch.int32, 1, align=4)
        if send_counts.numel() != self.world_size * E_local:
            raise RuntimeError("send_counts: expected W * E_local entries")
        recv = torch.empty_like(send_counts)
        rc = _lib.load().smoe_a2a_counts(self._h, send_counts.data_ptr(), recv.data_ptr(), E_local, ops._stream(send_counts),
                                         1 if wait else 0)
        _lib.check(rc, "smoe_a2a_counts")
        return recv

    def all_to_all_rows(self, rows: torch.Tensor, send_rows: List[int], recv_rows: List[int], wait: bool = True,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """all-to-all-v of whole rows (``rows[:sum(send_rows)]`` split by destination rank); returns the receive buffer.
        With ``wait=False`` the caller's stream does NOT wait for the exchange: call ``wait_stream(ref, ticket)`` with
        ``last_ticket()`` taken right after this call (or ``wait_stream(ref)`` = the latest exchange) before reading it.
        ``wait="inline"`` posts the exchange on the caller's stream itself (SMOE_A2A_INLINE: no second stream, no events)."""
        ops._chk(rows, "rows", ndim=2, align=4)
        W = self.world_size
        if len(send_rows) != W or len(recv_rows) != W:
            raise RuntimeError("send_rows / recv_rows: one entry per rank")
        if out is None:
            out = torch.empty((int(sum(recv_rows)), rows.shape[1]), dtype=rows.dtype, device=rows.device)
        else:
            ops._chk(out, "out", rows.dtype, 2, align=4)
            if out.shape[0] < int(sum(recv_rows)) or out.shape[1] != rows.shape[1]:
                raise RuntimeError("out: too small for the received rows")
        s_arr = (ctypes.c_int64 * W)(*[int(v) for v in send_rows])
        r_arr = (ctypes.c_int64 * W)(*[int(v) for v in recv_rows])
        rc = _lib.load().smoe_a2a_tokens(self._h, rows.data_ptr() if rows.numel() else None, s_arr,
                                         out.data_ptr() if out.numel() else None, r_arr, rows.shape[1],
                                         self._size_code(rows), ops._stream(rows), 2 if wait == "inline" else 1 if wait else 0)
        _lib.check(rc, "smoe_a2a_tokens")
        if wait == "inline":
            return out
        # both buffers are in use on the context's stream: keep the allocator from recycling them under it
        cs = torch.cuda.ExternalStream(_lib.load().smoe_ctx_comm_stream(self._h), device=rows.device)
        rows.record_stream(cs)
        out.record_stream(cs)
        return out

    @staticmethod
    def _size_code(t: torch.Tensor) -> int:
        """Rows travel as bytes: any 2- or 4-byte element type maps to the dtype code of that size."""
        if t.element_size() == 4:
            return ops.F32
        if t.element_size() == 2:
            return ops.F16
        raise TypeError(f"all_to_all_rows: element size {t.element_size()} not supported (2 or 4 bytes)")

    def last_ticket(self) -> int:
        """Ticket of the exchange posted last on this context (0 before the first one)."""
        return int(_lib.load().smoe_a2a_last_ticket(self._h))

    def wait_stream(self, ref: torch.Tensor, ticket: Optional[int] = None):
        """Make torch's current stream (on ``ref``'s device) wait for exchange ``ticket`` of this context (default: the
        latest).  Waiting for an older exchange does not wait for the ones posted after it."""
        lib = _lib.load()
        if ticket is None:
            _lib.check(lib.smoe_a2a_wait(self._h, ops._stream(ref)), "smoe_a2a_wait")
        else:
            _lib.check(lib.smoe_a2a_wait_ticket(self._h, int(ticket), ops._stream(ref)), "smoe_a2a_wait_ticket")
