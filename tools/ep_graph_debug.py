#!/usr/bin/env python3
"""Debug probe: which piece of the world-of-one static EP forward survives HIP-graph capture (run under faulthandler).

    python tools/ep_graph_debug.py a2a               all_to_all_single(async_op=True): RCCL's stream joined into the capture by events
                                                     -> hipStreamEndCapture segfaults (torch 2.10 + rocm 7.0, RCCL 2.26.6)
    python tools/ep_graph_debug.py a2a_sync          the same collective with async_op=False (on the capturing stream): captures, replays
    python tools/ep_graph_debug.py a2a_cabi_inline   the library's own transport with SMOE_A2A_INLINE: captures, replays
    python tools/ep_graph_debug.py model             two blocks of the bench model on the speculative static exchange (exchanges on the
                                                     compute stream: ep.exchange_inline): captures, replays bit-exactly
A live graph holding RCCL kernels makes destroy_process_group() hang: the graph is dropped first."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import slim_switch_moe_vit_amd as sm  # noqa: E402
from slim_switch_moe_vit_amd import ep  # noqa: E402

DEV = "cuda:0"
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=torch.device(DEV))
what = sys.argv[1] if len(sys.argv) > 1 else "a2a"
if what == "a2a":
    a = torch.randn(1024, 768, device=DEV).half()
    b = torch.empty_like(a)
    dist.all_to_all_single(b, a)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_to_all_single(b, a)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print("capturing all_to_all_single", flush=True)
    with torch.cuda.graph(g):
        w = dist.all_to_all_single(b, a, async_op=True)
        w.wait()
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed", bool(torch.equal(a, b)), flush=True)
elif what in ("a2a_sync", "a2a_cabi_inline"):
    # the collective posted on the CAPTURING stream itself (no cross-stream join inside the capture)
    a = torch.randn(1024, 768, device=DEV).half()
    b = torch.empty_like(a)
    if what == "a2a_sync":
        def go():
            dist.all_to_all_single(b, a, async_op=False)
    else:
        from slim_switch_moe_vit_amd.comm import ExchangeContext
        ctx = ExchangeContext(ExchangeContext.new_unique_id(), 1, 0, torch.device(DEV))

        def go():
            ctx.all_to_all_rows(a, [1024], [1024], wait="inline", out=b)
    go()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        go()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    b.zero_()
    print("capturing", what, flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        go()
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed", bool(torch.equal(a, b)), flush=True)
else:
    from test_gpu_model import _init
    model = _init(sm.create_model("moe_base_patch16_224_expert8_top1", num_classes=100, depth=2), 23).eval().to(DEV)
    for blk in model.blocks:
        blk.mlp.force_ep = True
    model.ep_micro_batches = 1
    ep.set_speculative(model, 2.0)
    images = torch.randn(16, 3, 224, 224, generator=torch.Generator().manual_seed(24)).to(DEV)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model(images)
    eager = step().float().clone()
    ep.check_static_overflow(flush=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ep.check_static_overflow(flush=True)
    print("capturing model", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        out = step()
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed", float((out.float() - eager).abs().max()), flush=True)
g = None
torch.cuda.synchronize()
print("graph freed", flush=True)
if os.environ.get("EP_GRAPH_DEBUG_HARD_EXIT", "0") == "1":   # (the process-group teardown after a captured collective is a probe of its own)
    os._exit(0)
dist.destroy_process_group()
print("group destroyed", flush=True)
