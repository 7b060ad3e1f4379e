#!/usr/bin/env python3
"""BASELINE cfg 5 timing: forward + backward (+ AdamW step through NativeScaler) with the SwitchGate, capacity_factor 1.0 and the
aux loss, (a) for ONE MoE layer at ViT-B dims (T = images x 197 rows) and (b) for the whole ViT-B/16 E=8 model.
usage: train_bench.py [layer|model] [images] [iters] [model name]     (run under rocprofv3 --kernel-trace --stats for the per-kernel table)
model name: default moe_base_patch16_224_expert8_top1 with the SwitchGate (cfg 5); a resmoe_* name = the reference's live block
(token-skip gates, residual on the normed activations, naive gate; thresholds set so that ~40 % of the tokens skip).
TRAIN_BENCH_EP=static|counted (model mode): the step through the EXPERT-PARALLEL code path on a one-rank RCCL group -- cfg 5's capacity
gate on the static exchange (fixed slots, counts in-band, no host round trip) or on the counted one (a count read-back per layer)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slim_switch_moe_vit_amd as sm
from slim_switch_moe_vit_amd import optim as smo

what = sys.argv[1] if len(sys.argv) > 1 else "layer"
images = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(0)
g = torch.Generator().manual_seed(5)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if what == "layer":
    d, h, E = 768, 3072, 8
    T = images * 197
    mod = sm.FMoETransformerMLP(E, d, h, torch.nn.GELU(), top_k=1, gate="switch", capacity_factor=1.0).to(dev)
    with torch.no_grad():
        for p in mod.parameters():
            if p.dim() > 1:
                p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(dev))
    mod.train()
    x = torch.randn(T, d, generator=g).to(dev).requires_grad_(True)
    gout = (torch.randn(T, d, generator=g) * 0.01).to(dev)

    def fwd():
        with torch.autocast("cuda", dtype=torch.float16):
            return mod(x)

    def fwd_bwd():
        for p in mod.parameters():
            p.grad = None
        x.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            out = mod(x)
            loss = (out.float() * gout).sum() + 0.01 * mod.gate.get_loss()
        loss.backward()

    with torch.no_grad():
        t_f = timed(fwd, iters)
    t_fb = timed(fwd_bwd, iters)
    fl = 4.0 * T * d * h          # forward expert-GEMM FLOPs (kept rows <= T); backward = 2x that
    print(f"MoE layer cfg 5 (T {T}, d {d}, h {h}, E {E}, switch gate, cf 1.0): forward {t_f:.3f} ms, forward+backward "
          f"{t_fb:.3f} ms  ({3 * fl / (t_fb * 1e-3) / 1e12:.0f} TFLOP/s over the 3 x 4Tdh expert-GEMM FLOPs)", flush=True)
else:
    name = sys.argv[4] if len(sys.argv) > 4 else "moe_base_patch16_224_expert8_top1"
    if name.startswith("resmoe"):
        model = sm.create_model(name, num_classes=1000, drop_path_rate=0.1, starting_threshold=0.55, target_threshold=0.5)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if "_gate.head.1.weight" in n:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.05)
        model = model.to(dev)
    elif os.environ.get("TRAIN_BENCH_GATE") == "naive":     # the reference's capacity-less gate (speculative slots under TRAIN_BENCH_EP=static)
        model = sm.create_model(name, num_classes=1000).to(dev)
    else:
        model = sm.create_model(name, num_classes=1000, gate="switch", capacity_factor=1.0).to(dev)
    ep_mode = os.environ.get("TRAIN_BENCH_EP")
    speculative = False
    if ep_mode:
        import contextlib
        import torch.distributed as dist
        from slim_switch_moe_vit_amd import ep
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29591", rank=0, world_size=1, device_id=dev)
        for m in model.modules():
            if isinstance(m, sm.FMoETransformerMLP):
                m.force_ep = True
        ep.set_static_tokens(model, images * 197, 197)
        if ep_mode == "static":
            speculative = ep.set_speculative(model, 1.25, train=True) > 0
    model.train()
    opt = smo.AdamW(model.parameters(), lr=1e-4, weight_decay=0.05)
    scaler = smo.NativeScaler()
    crit = torch.nn.CrossEntropyLoss()
    x = torch.randn(images, 3, 224, 224, generator=g).to(dev)
    y = torch.randint(0, 1000, (images,), generator=g).to(dev)
    moes = [m for m in model.modules() if isinstance(m, sm.FMoETransformerMLP)]

    repeats = [0]

    def forward_loss():
        with torch.autocast("cuda", dtype=torch.float16):
            out = model(x)
            loss = crit(out, y)
            auxes = [a for a in (m.gate.get_loss() for m in moes) if a is not None]
            if auxes:
                loss = loss + 0.01 * torch.stack([a.reshape(()) for a in auxes]).sum()
        return loss

    def train_step():
        if speculative:      # what engine.train_one_epoch does: the overflow report is read before the backward
            loss, again = ep.run_guarded(forward_loss, flush=True)
            repeats[0] += int(again)
        else:
            loss = forward_loss()
        opt.zero_grad()
        scaler(loss, opt, clip_grad=1.0, parameters=model.parameters())

    if ep_mode:
        calls = {"n": 0}
        orig_counts = ep.exchange_counts

        def counting(*a, **k):
            calls["n"] += 1
            return orig_counts(*a, **k)
        ep.exchange_counts = counting
        with (ep.dynamic_only() if ep_mode == "counted" else contextlib.nullcontext()):
            t = timed(train_step, iters)
            ep.check_static_overflow(flush=True)
        print(f"{name} train step through the expert-parallel path ({ep_mode} exchange, one-rank group), batch {images}: {t:.2f} ms = "
              f"{images / t * 1e3:.0f} images/s; count exchanges with a host read-back: {calls['n'] / (iters + 3):.1f} per step"
              + (f"; speculative slots, {repeats[0]} forwards repeated in {iters + 3} steps" if speculative else ""), flush=True)
        if os.environ.get("TRAIN_BENCH_GRAPH") and ep_mode == "static" and not speculative:
            # no host round trip left in the step and its collectives sit on the compute stream: the whole expert-parallel training
            # step captures into one HIP graph
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    train_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ep.check_static_overflow(flush=True)
            gr = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(gr, capture_error_mode="relaxed"):
                train_step()
            tg = timed(gr.replay, iters)
            print(f"{name} expert-parallel train step (static exchange) replayed from one HIP graph, batch {images}: {tg:.2f} ms = "
                  f"{images / tg * 1e3:.0f} images/s (eager {t:.2f} ms)", flush=True)
            gr = None
        torch.cuda.synchronize()
        dist.destroy_process_group()
        sys.exit(0)
    t = timed(train_step, iters)
    if os.environ.get("TRAIN_BENCH_GRAPH"):
        # the WHOLE step (forward, backward, clip, AdamW, loss-scale update: no host sync anywhere) captured into one HIP graph and
        # replayed: what the ~1,000 launches per step cost the host
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                train_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(gr, capture_error_mode="relaxed"):     # (pinned staging buffers are allocated inside the step)
            train_step()
        tg = timed(gr.replay, iters)
        print(f"{name} train step replayed from one HIP graph, batch {images}: {tg:.2f} ms = {images / tg * 1e3:.0f} images/s (eager {t:.2f} ms)", flush=True)
    gates = [m for m in model.modules() if isinstance(m, sm.Gate)]
    extra = ""
    if gates:
        extra = f"; skipped tokens {sum(gt._skipped_tokens for gt in gates) / max(1, sum(gt._total_tokens for gt in gates)):.2f}"
    print(f"{name} train step (fwd + bwd + clip + AdamW), batch {images}: {t:.2f} ms = {images / t * 1e3:.0f} images/s{extra}", flush=True)

    if os.environ.get("TRAIN_BENCH_TRACE"):
        # which host call starts each torch (non-library) kernel of the step: aten op, input shapes, innermost package frames
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
            train_step()
            torch.cuda.synchronize()
        rows = {}
        for ev in prof.events():
            if not ev.name.startswith("aten::") or not ev.kernels:
                continue
            dt = sum(k.duration for k in ev.kernels)
            frames = [f for f in (ev.stack or []) if "slim" in f or "tools/" in f][:3]
            key = (ev.name, str(ev.input_shapes)[:90], " <- ".join(f.split("/")[-1][:60] for f in frames))
            r = rows.setdefault(key, [0, 0.0, ev.kernels[0].name[:70]])
            r[0] += 1
            r[1] += dt
        print("torch ops that launch kernels in ONE step (calls, us total, op, shapes, frames, first kernel):")
        for key, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
            print(f"{r[0]:4d} {r[1]:9.1f}  {key[0]:28s} {key[1]:90s} {key[2]} [{r[2]}]", flush=True)
