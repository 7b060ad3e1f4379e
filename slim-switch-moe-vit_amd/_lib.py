"""ctypes loader for libslimmoe_hip.so (the C-ABI in include/slimmoe.h).

There is no CPU fallback: if the library is missing or lacks a symbol this module raises, so a
product path can never silently run on something other than the HIP kernels."""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must be imported first: the library resolves libamdhip64.so.7 to torch's runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLIMMOE_LIB: another build of the same sources (a diagnostic build, `make DIAG=-DSMOE_DIAG`, whose environment switches the
# tools under tools/ use); the ABI and symbol checks below apply to it all the same
LIB_PATH = os.environ.get("SLIMMOE_LIB") or os.path.join(_HERE, "libslimmoe_hip.so")
ABI_VERSION = 26

c_void_p, c_int, c_int64, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/slimmoe.h one to one
SIGNATURES = {
    "smoe_abi_version": (c_int, []),
    "smoe_build_id": (ctypes.c_char_p, []),
    "smoe_init": (c_int, []),
    "smoe_set_reserved_cus": (c_int, [c_int]),
    "smoe_last_error": (ctypes.c_char_p, []),
    "smoe_router_workspace_bytes": (c_size_t, [c_int64]),
    "smoe_router_topk": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_dispatch_plan_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "smoe_dispatch_plan": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_dispatch_plan_padded": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_dispatch_plan_slots": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_ep_pack_headers": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "smoe_ep_unpack_headers": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "smoe_scatter_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "smoe_scatter_rows_fill": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "smoe_gelu": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "smoe_rowdot": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "smoe_pad_offsets": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "smoe_split_offsets": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "smoe_transpose_pad": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "smoe_grouped_gemm_gelu_keep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int,
                                            c_int, c_void_p, c_void_p, c_void_p]),
    "smoe_switch_gate_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "smoe_gate_ln_bwd_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "smoe_gate_ln_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_int,
                                 c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_depth_scale_rows": (c_int, [c_void_p, ctypes.c_float, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "smoe_zero_group_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "smoe_switch_aux_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "smoe_switch_aux": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_transpose_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "smoe_grouped_wgrad": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "smoe_grouped_wgrad_rows": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "smoe_gate_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "smoe_gate_wgrad": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_group_colsum_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "smoe_group_colsum": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_gather_combine": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                    c_int, c_void_p]),
    "smoe_gather_combine_ln": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, ctypes.c_float, c_void_p, c_int, c_void_p]),
    "smoe_grouped_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int,
                                  c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int,
                                  c_int, c_void_p, c_void_p]),
    "smoe_expert_ffn_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "smoe_expert_ffn": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                c_int, c_void_p, c_size_t, c_void_p]),
    "smoe_layernorm": (c_int, [c_void_p, c_int, c_void_p, c_void_p, ctypes.c_float, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "smoe_gate_dgrad": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "smoe_layernorm_bwd_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "smoe_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, ctypes.c_float, c_int64, c_int, c_void_p,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_attention_supported": (c_int, [c_int, c_int]),
    "smoe_attention_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_float, c_void_p, c_void_p]),
    "smoe_attention_bwd_supported": (c_int, [c_int, c_int]),
    "smoe_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                   ctypes.c_float, c_void_p]),
    "smoe_ln_router_supported": (c_int, [c_int, c_int, c_int]),
    "smoe_ln_router_topk": (c_int, [c_void_p, c_int, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_router_chunk_hist_tokens": (c_int, [c_int, c_int, c_int]),
    "smoe_dispatch_plan_hist": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "smoe_gate_ln_router_supported": (c_int, [c_int, c_int, c_int]),
    "smoe_gate_ln_router": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "smoe_grad_sumsq_blocks": (c_int64, [c_int64]),
    "smoe_grad_sumsq": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "smoe_adamw_step": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, ctypes.c_float, ctypes.c_float,
                                ctypes.c_float, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "smoe_grad_sumsq_multi": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "smoe_adamw_step_multi": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "smoe_amp_update": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_float, ctypes.c_float, c_int, c_void_p]),
    "smoe_step_advance": (c_int, [c_void_p, c_void_p, c_void_p]),
    "smoe_unique_id_bytes": (c_int, []),
    "smoe_unique_id": (c_int, [c_void_p]),
    "smoe_ctx_create": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "smoe_ctx_destroy": (c_int, [c_void_p]),
    "smoe_ctx_comm_stream": (c_void_p, [c_void_p]),
    "smoe_ctx_world_size": (c_int, [c_void_p]),
    "smoe_ctx_rank": (c_int, [c_void_p]),
    "smoe_a2a_counts": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int]),
    "smoe_a2a_tokens": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int]),
    "smoe_a2a_wait": (c_int, [c_void_p, c_void_p]),
    "smoe_a2a_last_ticket": (c_int64, [c_void_p]),
    "smoe_a2a_wait_ticket": (c_int, [c_void_p, c_int64, c_void_p]),
    "smoe_patchify_cast": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "smoe_embed_ln": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_int64, c_int, c_int, c_void_p,
                              c_void_p, c_int, c_void_p]),
    "smoe_layernorm_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, ctypes.c_float, c_int64, c_int, c_void_p, c_void_p]),
    "smoe_skip_gate_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p,
                                   c_void_p, c_void_p]),
    "smoe_zero_row_output": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p]),
}

# entry points that only an optional build holds (`make FFN=-DSMOE_FFN_FUSED`): bound when present, never required
OPTIONAL = {"smoe_expert_ffn_workspace_bytes", "smoe_expert_ffn"}

_lib = None

# what csrc/Makefile hashes into smoe_build_id(): the same names, sorted as strings, relative to csrc/
_HASHED = ["api.hip", "router.hip", "router16.hip", "gate.hip", "dispatch.hip", "gemm.hip", "backward.hip", "attention.hip",
           "optim.hip", "comm.hip", "dense_bwd.hip", "attention_bwd.hip", "embed.hip", "smoe_common.h", "router16_kernel.h", "router_mt_kernel.h", "gemm_persistent.h",
           "../../include/slimmoe.h", "Makefile"]


class SlimMoEError(RuntimeError):
    pass


def has_symbol(name: str) -> bool:
    """Whether the loaded library exports an OPTIONAL entry point."""
    lib = load()
    try:
        getattr(lib, name)
        return True
    except AttributeError:
        return False


def source_build_id(extra_flags: str = "") -> str:
    """sha256 (first 16 hex digits) over the library's sources as they lie in the tree -- what ``smoe_build_id()`` of a
    binary built from them returns.  None if the sources are not there (a binary-only install).  ``extra_flags``: appended to the
    flags line as the Makefile appends them for a derived build (" -DSMOE_CLOCK" = the clock-probe library's id)."""
    import hashlib
    import re
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    try:
        for name in sorted(_HASHED):   # GNU make's $(sort) and Python's sorted() agree on these ASCII names
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(f.read())
        # the production flags line, as the Makefile expands it with its defaults (ARCH = gfx950, DIAG empty)
        mk = open(os.path.join(csrc, "Makefile")).read()
        flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, re.M).group(1)
        arch = re.search(r"^ARCH\s*\?=\s*(\S+)", mk, re.M).group(1)
        flags = flags.replace("$(ARCH)", arch).replace("$(DIAG)", "").replace("$(FFN)", "")
        h.update((" ".join(flags.split()) + extra_flags + "\n").encode())
    except (OSError, AttributeError):
        return None
    return h.hexdigest()[:16]


def binary_build_id() -> str:
    return load().smoe_build_id().decode()


def load():
    """Load (once) and return the ctypes handle; raises SlimMoEError if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlimMoEError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the MoE hot path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            if name in OPTIONAL:
                continue
            raise SlimMoEError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    got = lib.smoe_abi_version()
    if got != ABI_VERSION:
        raise SlimMoEError(f"libslimmoe_hip.so ABI {got} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


_inited_devices = set()


def init_device(index: int):
    """smoe_init() once per device of this process: every kernel's dynamic-LDS limit is raised up front, so that no
    launcher touches function attributes afterwards (graph capture, several streams / threads)."""
    if index in _inited_devices:
        return
    lib = load()
    with torch.cuda.device(index):
        check(lib.smoe_init(), "smoe_init")
    _inited_devices.add(index)


def check(rc: int, what: str):
    if rc != 0:
        msg = load().smoe_last_error()
        raise SlimMoEError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
