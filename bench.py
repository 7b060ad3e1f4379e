#!/usr/bin/env python3
"""Headline benchmark: images/sec (forward) of ViT-B/16 with a Switch-MoE MLP (E=8, top-1) in every block,
224x224, batch 256 per GPU -- BASELINE.json configs[1] -- plus the roofline of the dominant hot-path kernel
(the grouped expert GEMM) and the CPU oracle timed beside it.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no launcher: starts the N ranks itself as child processes and relays rank 0's line)

A step = one eval-mode forward of the whole model over one batch of synthetic images already resident in
HBM (reference harness: engine.py:88-121 -- eval(), no_grad, fp16 autocast).  The MoE operator (router,
dispatch plan, token scatter, grouped GEMMs, combine) runs on the hand-written HIP kernels, and so does the dense
shell around it under fp16 autocast (patch embedding, qkv / projection / head on the grouped GEMM with one row group,
the attention kernel, LayerNorm, the embedding stage: csrc/embed.hip); nothing of torch is left in the step but a memset.
N > 1 = expert parallel: the 8
experts are partitioned over the ranks, every rank keeps 256 images (weak scaling) and tokens travel by
RCCL all-to-all (slim_switch_moe_vit_amd/ep.py).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}   # dense, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--experts", type=int, default=8, help="global number of experts")
    ap.add_argument("--compute-dtype", default=os.environ.get("SLIMMOE_COMPUTE_DTYPE", "f16"), choices=["f16", "bf16", "f32"])
    ap.add_argument("--ep-chunks", type=int, default=0,
                    help="expert parallel only: token chunks per MoE layer (chunk c + 1's all-to-all travels under chunk c's expert "
                         "GEMMs; 1 = off; 0 = timed together with --ep-micro-batches during warm-up)")
    ap.add_argument("--ep-micro-batches", type=int, default=0,
                    help="expert parallel only: micro-batches of the local batch interleaved through the model so that "
                         "count read-backs and all-to-alls of one run under the other's compute (1 = off; 0 = time "
                         "1, 2 and 3 during warm-up and keep the fastest -- the right depth depends on the link rate)")
    ap.add_argument("--ep-static", default="auto", choices=["auto", "0", "1"],
                    help="expert parallel only: the speculative STATIC exchange for the capacity-less gate (fixed alpha-sized slots, "
                         "counts in-band, no host round trip per layer; an overflowing step is repeated on the counted exchange) -- "
                         "1 = on, 0 = off (count read-back per layer), auto = timed against the counted exchange during warm-up")
    ap.add_argument("--ep-alpha", type=float, default=1.25,
                    help="speculative slot size as a multiple of the balanced share rows * k / E (raised automatically during warm-up "
                         "to 1.1 x the largest group a step needed)")
    ap.add_argument("--ep-reserve-cus", type=int, default=-1,
                    help="expert parallel only: CUs the persistent expert GEMM leaves free (ops.set_reserved_cus) so that RCCL's all-to-all "
                         "of one micro-batch can run BESIDE the other micro-batch's GEMMs (the persistent kernel otherwise owns every CU "
                         "for the whole launch); -1 = time 0 and 16 for every pipelined configuration during warm-up")
    ap.add_argument("--cpu-batch", type=int, default=128, help="images in the CPU-oracle sample")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU budget: 5 timed forwards if they fit ~2x this, else 3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clock-seconds", type=float, default=2.0,
                    help="seconds of back-to-back launches per expert GEMM in the clock probe (0 = skip the probe)")
    ap.add_argument("--gemm-variant", type=int, default=None,
                    help="grouped-GEMM kernel (default: ops.DEFAULT_GEMM_VARIANT = 9, the persistent kernel; 4 = one "
                         "workgroup per tile, bit-identical results)")
    ap.add_argument("--compute-streams", type=int, default=1, choices=[1, 2],
                    help="2 = the two halves of the batch on two compute streams (VisionTransformer.compute_streams; "
                         "opt-in, single rank only)")
    ap.add_argument("--force-ep", action="store_true",
                    help="diagnostic: drive the expert-parallel code path on one GPU (world of one rank)")
    return ap.parse_args()


def build_model(args, world, rank, device):
    """ViT-B/16 shell + MoE MLP in every block; random init of that architecture (no checkpoints offline).
    Router / expert tensors are drawn for all E experts from one seed, then sliced per rank, so the
    N-GPU model is the same function as the 1-GPU model."""
    import slim_switch_moe_vit_amd  # noqa: F401  (registers the package alias for the hyphenated directory)
    from slim_switch_moe_vit_amd.vit import _deit
    from slim_switch_moe_vit_amd.resmoe import patch_blocks_with_moe

    E, d, h = args.experts, 768, 3072
    assert E % world == 0, "experts must divide over ranks"
    E_local = E // world
    cd = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.compute_dtype]
    torch.manual_seed(0)
    model = _deit(768, 12, 12, num_classes=1000)
    patch_blocks_with_moe(model, E_local, 1, False, world_size=world, compute_dtype=cd, gemm_variant=args.gemm_variant)
    g = torch.Generator().manual_seed(1)
    full_sd = {}  # all-E tensors for the oracle (rank 0, N=1 only keeps them)
    with torch.no_grad():
        for i, blk in enumerate(model.blocks):
            wg = torch.randn(E, d, generator=g) * 0.02
            w1 = torch.nn.init.trunc_normal_(torch.empty(E, h, d), std=0.02, a=-2, b=2, generator=g)
            w2 = torch.nn.init.trunc_normal_(torch.empty(E, d, h), std=0.02, a=-2, b=2, generator=g)
            blk.mlp.gate.gate.weight.copy_(wg)
            blk.mlp.gate.gate.bias.zero_()
            sl = slice(rank * E_local, (rank + 1) * E_local)
            blk.mlp.experts.htoh4.weight.copy_(w1[sl]); blk.mlp.experts.htoh4.bias.zero_()
            blk.mlp.experts.h4toh.weight.copy_(w2[sl]); blk.mlp.experts.h4toh.bias.zero_()
            blk.mlp.ep_chunks = max(1, args.ep_chunks)
            blk.mlp.force_ep = bool(getattr(args, "force_ep", False))
            # the head is zero-initialised in the reference (vision_transformer.py:859-861); give it signal
        torch.nn.init.trunc_normal_(model.head.weight, std=0.02, a=-2, b=2, generator=g)
    model.ep_micro_batches = max(1, args.ep_micro_batches)
    model.compute_streams = args.compute_streams if world == 1 else 1
    sd_cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        sd_cpu = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model.to(device).eval(), sd_cpu


def host_cpu_share():
    """CPUs this process may actually use: the cgroup quota when there is one (a GPU box hands a container a share of the
    host, e.g. 16 of 256 logical CPUs -- running 128 torch threads on that share is slower than running 16), else the
    affinity mask / os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(sd, images, seconds):
    """The CPU oracle (oracle/moe_oracle.py: 'port' of the path, fp32 torch on ALL host cores torch sees) on a bounded
    sample of the same workload: one warm-up forward, then >= 5 timed forwards of the same batch (3 if a forward takes
    longer than `seconds` / 5); the reported rate is batch / median forward time."""
    from oracle import moe_oracle as mo

    n = images.shape[0]
    threads = min(torch.get_num_threads(), host_cpu_share())
    torch.set_num_threads(threads)
    with torch.no_grad():
        t0 = time.perf_counter()
        logits = mo.vit_forward(images, sd, depth=12, num_heads=12, k=1, residual_moe=False)
        first = time.perf_counter() - t0
        iters = 5 if first * 5 <= 2.0 * seconds else (3 if first * 3 <= 2.0 * seconds else 2)
        times = []
        for _ in range(iters):
            t0 = time.perf_counter()
            mo.vit_forward(images, sd, depth=12, num_heads=12, k=1, residual_moe=False)
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return n / med, logits, {"iters": iters, "threads": threads, "first_call_s": round(first, 3), "median_s": round(med, 3),
                             "min_s": round(times[0], 3), "max_s": round(times[-1], 3)}


def hot_path_parity(model, sd_cpu, device):
    """Layer-0 MoE operator on cfg-2-sized random tokens vs the oracle (routing on all tokens, expert
    outputs on a 1024-token sample)."""
    from oracle import moe_oracle as mo

    g = torch.Generator().manual_seed(7)
    x = torch.randn(256 * 197, 768, generator=g)
    mlp = model.blocks[0].mlp
    with torch.no_grad():
        out = mlp(x.to(device).reshape(256, 197, 768)).reshape(-1, 768).cpu()
    pre = "blocks.0.mlp."
    wg, bg = sd_cpu[pre + "gate.gate.weight"], sd_cpu[pre + "gate.gate.bias"]
    o_idx, _, _ = mo.naive_gate(x, wg, bg, 1)
    routing_exact = bool(torch.equal(mlp.last_plan[0].cpu(), o_idx))
    sel = torch.randperm(x.shape[0], generator=g)[:1024]
    r = mo.moe_forward(x[sel], wg, bg, sd_cpu[pre + "experts.htoh4.weight"], sd_cpu[pre + "experts.htoh4.bias"],
                       sd_cpu[pre + "experts.h4toh.weight"], sd_cpu[pre + "experts.h4toh.bias"], 1)
    diff = out[sel] - r.out
    # the float bar, stated as the parity tests state it (tests/test_gpu_parity.py:_float_bar):
    #   max |diff| <= 1e-3 * max(1, max |ref|)  and  relative L2 <= 1e-3
    ref_abs_max = float(r.out.abs().max())
    max_abs, rel_l2 = float(diff.abs().max()), float(diff.norm() / r.out.norm())
    bar = 1e-3 * max(1.0, ref_abs_max)
    return {"routing_bit_exact": routing_exact, "expert_out_rel_l2_err": rel_l2, "expert_out_max_abs_err": max_abs,
            "ref_abs_max": ref_abs_max, "max_abs_bar": bar, "rel_l2_bar": 1e-3,
            "within_bar": bool(routing_exact and max_abs <= bar and rel_l2 <= 1e-3),
            # the north star's 1e-3 read as a bare absolute bound on outputs of scale ~3 (the fp16 operand floor)
            "within_strict_abs_1e-3": bool(max_abs <= 1e-3), "sample_tokens": 1024}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # started the way the N = 1 bench is started: launch the N ranks ourselves, as CHILD processes, before this process
        # has touched the GPU (never re-exec a process that initialised it), relay their output and exit with their code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        proc = subprocess.run(cmd, env=env)
        sys.exit(proc.returncode)
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # diagnostics only: SMOE_BENCH_BACKEND=gloo rehearses the multi-rank path with several ranks sharing one GPU
    # (the exchange is staged through the host, ep._a2a); the numbers of such a run mean nothing
    backend = os.environ.get("SMOE_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 and backend != "nccl":
        dist.init_process_group(backend)
    elif world > 1:
        dist.init_process_group("nccl", device_id=device)
    elif args.force_ep:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=device)

    from slim_switch_moe_vit_amd import ops

    model, sd_cpu = build_model(args, world, rank, device)
    gi = torch.Generator().manual_seed(100 + rank)
    images_cpu = torch.randn(args.batch, 3, 224, 224, generator=gi)   # synthetic, never zero-filled
    images = images_cpu.to(device)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model(images)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    ep_tuning = None
    ep_repeats = 0
    ep_static = False
    moes = [blk.mlp for blk in model.blocks]
    if world > 1 or args.force_ep:
        from slim_switch_moe_vit_amd import ep as _ep

        def guarded():
            """One step whose overflow report is read at once (warm-up: the slots grow to what the routing needs)."""
            nonlocal ep_repeats
            _, again = _ep.run_guarded(step, flush=True)
            ep_repeats += int(again)

        # pipeline shape: micro-batches of the local batch interleaved through the whole model (count read-backs and all-to-alls of
        # one run under the other's attention / GEMMs) x token chunks inside a MoE layer (chunk c + 1 travels under chunk c's expert
        # GEMMs) x the exchange itself: counted (one host round trip per layer) or speculative static (fixed slots of alpha x the
        # balanced share, counts in-band, no host round trip, alpha x the all-to-all bytes).  Fewer / bigger = more efficient
        # kernels; more = more transfer hidden; static = no idle GPU behind a read-back, but more bytes on the links.  Which wins
        # depends on the xGMI rate at this world size, so measure (every rank takes the same, all-reduced, decision)
        ep_tuning = {}
        mbs = (1, 2, 3) if args.ep_micro_batches == 0 else (args.ep_micro_batches,)
        chs = (1, 2, 3) if args.ep_chunks == 0 else (args.ep_chunks,)
        statics = {"auto": (False, True), "0": (False,), "1": (True,)}[args.ep_static]
        # ... x CUs the persistent GEMM leaves free: only where there is something to run beside it (a pipelined configuration)
        reserves = (0, 16) if args.ep_reserve_cus < 0 else (args.ep_reserve_cus,)
        grid = [(st, n, c, r) for st in statics for n in mbs for c in (chs if not st else (1,)) if n * c <= 4
                for r in (reserves if n * c > 1 else reserves[:1] if args.ep_reserve_cus >= 0 else (0,))]
        for st, n, c, r in grid if len(grid) > 1 else []:
            _ep.set_speculative(model, args.ep_alpha if st else None)
            ops.set_reserved_cus(r)
            model.ep_micro_batches = n
            for m in moes:
                m.ep_chunks = c
            for _ in range(2):
                guarded()
            fence()
            t0 = time.perf_counter()
            ok = True
            try:
                for _ in range(3):
                    step()
                fence()
                _ep.check_static_overflow(flush=True)
            except _ep.StaticExchangeOverflow:  # (raised by every rank at the same exchange; the same steps fitted a moment ago, so
                ok = False                      #  this is not expected with a fixed batch: the cell is void, not the bench)
                fence()
            t = torch.tensor([(time.perf_counter() - t0) / 3 if ok else 1e9], dtype=torch.float64, device=device)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ep_tuning[("static " if st else "") + f"{n}x{c}" + (f" reserve {r}" if r else "")] = round(float(t.item()) * 1e3, 3)
        ep_reserve = 0
        if ep_tuning:
            best = min(ep_tuning, key=ep_tuning.get)
            ep_static = best.startswith("static ")
            core = best.replace("static ", "")
            if " reserve " in core:
                core, rs = core.split(" reserve ")
                ep_reserve = int(rs)
            args.ep_micro_batches, args.ep_chunks = (int(v) for v in core.split("x"))
        else:
            ep_static, args.ep_micro_batches, args.ep_chunks, ep_reserve = grid[0]
        ops.set_reserved_cus(ep_reserve)
        _ep.set_speculative(model, args.ep_alpha if ep_static else None)
        model.ep_micro_batches = args.ep_micro_batches
        for m in moes:
            m.ep_chunks = args.ep_chunks
    args.ep_chunks = max(1, args.ep_chunks)
    for _ in range(args.warmup):
        if ep_static:
            guarded()
        else:
            step()
    fence()
    # Per-launch HIP events inside the timed region: the roofline kernel only (two event records around each of the
    # ~110 other launches of a step cost ~0.5 ms per step).  The other kernels' table comes from a few extra,
    # untimed steps below.
    ops.profile_begin({"grouped_gemm", "expert_ffn"})
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step times (median reported)
    t0 = time.perf_counter()
    marks[0].record()
    ep_overflow_in_timed_region = False
    for i in range(args.steps):
        if ep_static:
            try:
                step()
            except _ep.StaticExchangeOverflow:      # raised by every rank at the same exchange (the deferred watch inside the forward):
                ep_overflow_in_timed_region = True  # the step is repeated on the counted exchange INSIDE the timed region, and the
                with _ep.dynamic_only():            # line says so (it cannot happen with one fixed batch whose routing fitted in warm-up)
                    step()
        else:
            step()
        marks[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ops.profile_end()
    if ep_static:
        try:     # the timed steps ran without reading their overflow reports (no host sync): read them now
            _ep.check_static_overflow(flush=True)
        except _ep.StaticExchangeOverflow:
            ep_overflow_in_timed_region = True      # the timed outputs are void: the line says so
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    side_steps = 0
    if world == 1 and not args.force_ep:
        side_steps = max(1, min(5, args.steps))
        ops.profile_begin()
        for _ in range(side_steps):
            step()
        fence()
        prof = prof + [(n, m2, ms) for n, m2, ms in ops.profile_end() if n not in ("grouped_gemm", "expert_ffn")]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- expert parallel: the raw exchange beside the step (a2a GB/s per link vs the 153 GB/s link) ---
    ep_info = None
    if world > 1:
        rows = (args.batch * 197 // world) * world          # equal splits: the average layer's volume
        sbuf = torch.randn(rows, 768, device=device).half()
        rbuf = torch.empty_like(sbuf)
        from slim_switch_moe_vit_amd.ep import _a2a
        for _ in range(3):
            _a2a(rbuf, sbuf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        e0.record()
        for _ in range(10):
            _a2a(rbuf, sbuf)
        e1.record()
        torch.cuda.synchronize(device)
        a2a_ms = e0.elapsed_time(e1) / 10
        t = torch.tensor([a2a_ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        a2a_ms = float(t.item())
        per_peer = rows // world * 768 * 2
        ep_info = {"a2a_payload_mb_per_rank": round(rows * 768 * 2 / 1e6, 1), "a2a_ms": round(a2a_ms, 4),
                   "a2a_gb_s_per_link": round(per_peer / (a2a_ms * 1e-3) / 1e9, 1), "link_peak_gb_s_bidir": 153,
                   "a2a_per_step": 24, "a2a_ms_per_step_if_exposed": round(24 * a2a_ms, 3),
                   "micro_batches": args.ep_micro_batches, "chunks_per_layer": args.ep_chunks,
                   "pipeline_tuning_ms_per_step (micro-batches x chunks)": ep_tuning}

    if ep_info is None and ep_tuning is not None:
        ep_info = {"micro_batches": args.ep_micro_batches, "chunks_per_layer": args.ep_chunks,
                   "pipeline_tuning_ms_per_step (micro-batches x chunks)": ep_tuning}
    if ep_info is not None:
        ep_info["cus_left_free_by_the_persistent_gemm"] = ep_reserve
        ep_info["exchange"] = ("speculative static (fixed slots, counts in-band, no host round trip per layer)" if ep_static
                               else "counted (count read-back + all-to-all-v per layer)")
        if ep_static:
            ep_info["speculative_alpha_first_table"] = args.ep_alpha
            div = max(1, int(model.ep_micro_batches))
            rows_mb = -(-args.batch // div) * 197       # routed rows of one (micro-)batch: top-1
            ep_info["send_buffer_rows_over_routed_rows_per_layer"] = [
                round(m.__dict__["_ep_slots"][div].table.rows / rows_mb, 3) if div in m.__dict__.get("_ep_slots", {}) else None for m in moes]
            ep_info["steps_repeated_on_the_counted_exchange_during_warmup"] = ep_repeats
            ep_info["overflow_in_timed_region"] = ep_overflow_in_timed_region

    # ---- per-kernel accounting from the HIP events recorded inside the timed region --------------------
    agg = {}
    for name, meta, ms in prof:
        if name == "expert_ffn":    # smoe_expert_ffn: both expert GEMMs in one persistent launch = the roofline kernel
            for nm in ("grouped_gemm", "expert_ffn"):
                a = agg.setdefault(nm, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
                a["launches"] += 1; a["ms"] += ms; a["flops"] += meta.get("flops", 0.0)
            continue
        if name == "grouped_gemm":
            for nm in ("grouped_gemm", "grouped_gemm_fc1" if meta.get("epilogue") == ops.EPI_GELU else "grouped_gemm_fc2"):
                a = agg.setdefault(nm, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
                a["launches"] += 1; a["ms"] += ms; a["flops"] += meta.get("flops", 0.0)
            continue
        a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        a["launches"] += 1
        a["ms"] += ms
        a["flops"] += meta.get("flops", 0.0)
        a["bytes"] += meta.get("bytes", 0.0)
    kernels = {}
    for name, a in agg.items():
        per = args.steps if (name.startswith("grouped_gemm") or name == "expert_ffn") else max(1, side_steps)
        ent = {"launches_per_step": a["launches"] / per, "avg_ms": a["ms"] / a["launches"]}
        if a["flops"]:
            ent["tflops"] = a["flops"] / (a["ms"] * 1e-3) / 1e12
        if a["bytes"]:
            ent["gbs"] = a["bytes"] / (a["ms"] * 1e-3) / 1e9
        kernels[name] = {k: round(v, 4) for k, v in ent.items()}
    roofline = None
    if "grouped_gemm" in agg:
        a = agg["grouped_gemm"]
        achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[args.compute_dtype]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("grouped_gemm_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"kernel": "expert FFN grouped GEMMs (both linears" + ("; one fused persistent launch per layer)" if "expert_ffn" in agg else "; two launches per layer)"),
                    "bound": "mfma", "achieved": round(achieved, 2),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_source": "profiles/roofline_traffic.json (rocprofv3 --pmc passes of an earlier run of this command, "
                                      "FETCH / WRITE corrected as MI355X_MICROARCH.md prescribes; not a counter of THIS run)",
                    "avg_launch_ms": round(a["ms"] / a["launches"], 4),
                    "flops_per_launch": a["flops"] / a["launches"]}
        # The data-sheet peak assumes 2.4 GHz; under an MFMA-dense kernel on random data the chip holds far less (package power).
        # The clock-probe build of the library (two s_memtime / s_memrealtime stamps per workgroup; tools/gemm_clock.py, a child
        # process) measures the clock held under each expert GEMM: frac_of_clocked_peak prices the kernel against the MFMA rate
        # the chip actually offered -- what is left to the schedule -- beside `frac`, which prices it against the data sheet.
        clock_lib = os.path.join(ROOT, "slim-switch-moe-vit_amd", "libslimmoe_hip_clock.so")
        if (rank == 0 and world == 1 and not args.force_ep and args.clock_seconds > 0 and os.path.exists(clock_lib)
                and args.compute_dtype in ("f16", "bf16")):
            import subprocess
            env = dict(os.environ, SLIMMOE_LIB=clock_lib)
            cp = None
            try:
                cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_clock.py"), str(args.clock_seconds),
                                     str(args.batch)], env=env, capture_output=True, text=True, timeout=120)
                clk = json.loads(cp.stdout.strip().splitlines()[-1])
            except Exception as exc:   # the probe is a report, never a reason to lose the bench line
                clk = {"error": f"{type(exc).__name__}: {exc}", "child_stderr_tail": (cp.stderr[-400:] if cp is not None else "")}
                print(f"bench.py: clock probe failed ({clk['error']}); is libslimmoe_hip_clock.so built from these sources "
                      "(make -C slim-switch-moe-vit_amd/csrc clock)?", file=sys.stderr)
            roofline["clock_probe"] = clk
            per_mhz = 1024 * 1024 * 1e6 / 1e12          # TFLOP/s per MHz: 1,024 SIMDs x 1,024 FLOP per cycle (2.4 GHz -> 2,517)
            f1, f2, ff = agg.get("grouped_gemm_fc1"), agg.get("grouped_gemm_fc2"), agg.get("expert_ffn")
            if "fused_mhz" in clk and ff:             # the fused launch (both GEMMs in one kernel) is what the step ran
                roofline["clocked_peak"] = round(clk["fused_mhz"] * per_mhz, 1)
                roofline["frac_of_clocked_peak"] = round(achieved / (clk["fused_mhz"] * per_mhz), 4)
            elif "gemm1_mhz" in clk and f1 and f2:
                offered = f1["ms"] * clk["gemm1_mhz"] * per_mhz + f2["ms"] * clk["gemm2_mhz"] * per_mhz   # TFLOP/s x ms
                roofline["clocked_peak"] = round(offered / (f1["ms"] + f2["ms"]), 1)
                roofline["frac_of_clocked_peak"] = round(a["flops"] / 1e9 / offered, 4)
                roofline["frac_of_clocked_peak_by_gemm"] = {
                    "gemm1": round(f1["flops"] / 1e9 / (f1["ms"] * clk["gemm1_mhz"] * per_mhz), 4),
                    "gemm2": round(f2["flops"] / 1e9 / (f2["ms"] * clk["gemm2_mhz"] * per_mhz), 4)}
        # the same launches under the names rocprofv3 --stats gives them (profiles/r03_bench_kernel_stats.csv):
        # grouped_gemm_ps<operand, out, AFR, DEEP, KEEP, DIRECT, BUF> (AFR 5 = 320-row tile; DEEP = the half-tile prefetch schedule
        # picked for K >= 2048; DIRECT = 16-bit outputs stored from the registers; BUF = f32 outputs through the buffer-addressed
        # staged epilogue); with --gemm-variant 4: grouped_gemm_pp256<operand, out, ABL, MODE, AFR>
        pers = (args.gemm_variant or ops.DEFAULT_GEMM_VARIANT) == 9
        from slim_switch_moe_vit_amd import fmoe as _fmoe
        tail_ln = _fmoe.TAIL_MODE == "ln"
        names = ({"expert_ffn": "expert_ffn_fused<f16> = GEMM-1 + GEMM-2 of a layer in one persistent launch (smoe_expert_ffn)",
                  "grouped_gemm_fc1": "grouped_gemm_ps<f16,f16,5,false,false,true,false> = GEMM-1 (K 768, gathered rows, bias + GELU)",
                  "grouped_gemm_fc2": ("grouped_gemm_ps<f16,f16,5,true,false,true,false> = GEMM-2 (K 3072, contiguous 16-bit rows; combine + "
                                       "residual + the next norm1 follow in smoe_gather_combine_ln); the last block's launch is the "
                                       "<f16,f32,5,true,false,false,true> instantiation (row-mapped f32 epilogue)"
                                       if tail_ln else
                                       "grouped_gemm_ps<f16,f32,5,true,false,false,true> = GEMM-2 (K 3072, combine + residual)"),
                  "attn_proj_gemm": "grouped_gemm_ps<f16,f32,5,false,false,false,true> = attention projection (K 768, + residual)",
                  "qkv_gemm": "grouped_gemm_ps<f16,f16,4,false,false,true,false> = qkv projection (K 768, N 2304; 256-row tiles)",
                  "patch_embed_gemm": "grouped_gemm_ps<f16,f16,5,false,false,true,false> (the GEMM-1 instantiation: its rocprof average "
                                      "includes these launches) = patch embedding (K 768, N 768)"}
                 if pers else
                 {"grouped_gemm_fc1": "grouped_gemm_pp256<f16,f16,0,0,5> = GEMM-1 (K 768, GELU)",
                  "grouped_gemm_fc2": "grouped_gemm_pp256<f16,f32,16,0,5> = GEMM-2 (K 3072, combine + residual)",
                  "attn_proj_gemm": "grouped_gemm_pp256<f16,f32,0,0,5> = attention projection (K 768, + residual)"})
        sym = {}
        for key, label in names.items():
            a2 = agg.get(key)
            if a2:
                sym[label] = {"launches_per_step": a2["launches"] / (args.steps if (key.startswith("grouped_gemm") or key == "expert_ffn") else max(1, side_steps)),
                              "avg_launch_ms": round(a2["ms"] / a2["launches"], 4)}
        if world == 1 and not args.force_ep:  # (under expert parallelism both GEMMs are the f16-out instantiation)
            roofline["by_rocprof_symbol"] = sym

    if rank == 0:
        total_images = args.batch * world * args.steps
        out = {
            "metric": "images/sec (fwd) ViT-B/16 E=8 @224^2",
            "value": round(total_images / elapsed, 2),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_median": round(step_ms[len(step_ms) // 2], 3),
            "ms_per_step_min_max": [round(step_ms[0], 3), round(step_ms[-1], 3)],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.compute_dtype,
            "data": "synthetic",
            "config": {"workload": f"ViT-B/16 Switch-MoE E={args.experts} top-1, 224^2, batch {args.batch}/GPU, "
                                   f"full eval forward (12 blocks: attention + MoE MLP), fp16 autocast",
                       "global_batch": args.batch * world, "tokens_per_image": 197, "compute_streams": args.compute_streams,
                       "parallelism": ("single" if not args.force_ep else f"single (EP code path forced, {'speculative static' if ep_static else 'counted'} exchange, {args.ep_micro_batches} interleaved micro-batches x {args.ep_chunks} chunks per layer)") if world == 1 else f"ep{world} (experts/{world} per rank, {'speculative static' if ep_static else 'counted'} all-to-all, {args.ep_micro_batches} interleaved micro-batches x {args.ep_chunks} chunks per layer)"},
            "roofline": roofline,
            "kernels": kernels,
        }
        hot = ("router", "ln_router", "plan", "scatter", "combine", "combine_ln", "grouped_gemm")  # the MoE operator's own kernels ("grouped_gemm" includes the fused launches)
        moe_ms = sum(a["ms"] / (args.steps if n.startswith("grouped_gemm") else max(1, side_steps))
                     for n, a in agg.items() if n in hot)
        out["hot_path"] = {"moe_kernels_ms_per_step": round(moe_ms, 3),
                           "share_of_step": round(moe_ms / (elapsed / args.steps * 1e3), 3)}
        if ep_info is not None:
            out["expert_parallel"] = ep_info
        if world == 1 and sd_cpu is not None:
            ips, cpu_logits, info = cpu_baseline(sd_cpu, images_cpu[: args.cpu_batch], args.cpu_seconds)
            out["cpu_baseline"] = {"value": round(ips, 3), "unit": "images/s", "cores": info["threads"],
                                   "kind": "port",
                                   "sample": f"oracle vit_forward (fp32 torch CPU, {info['threads']} threads = this "
                                             f"container's CPU share of os.cpu_count() = {os.cpu_count()}), same model, "
                                             f"batch {args.cpu_batch}, median of {info['iters']} forwards "
                                             f"({info['median_s']} s; min {info['min_s']}, max {info['max_s']})"}
            out["speedup_vs_cpu"] = round(out["value"] / ips, 1)
            # the parity forward runs as ONE batch on one stream: with micro-batches (or two compute streams) `last_plan` would
            # hold the routing of the last sub-forward only
            keep_mb, keep_cs = model.ep_micro_batches, model.compute_streams
            model.ep_micro_batches, model.compute_streams = 1, 1
            try:
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                    gl = model(images[: args.cpu_batch]).float().cpu()
            finally:
                model.ep_micro_batches, model.compute_streams = keep_mb, keep_cs
            forced = [blk.mlp.last_plan[0].reshape(-1, 1).cpu() for blk in model.blocks]   # the GPU's routing of THIS forward
            out["parity"] = hot_path_parity(model, sd_cpu, device)
            # whole model, fp16 autocast on the GPU against the fp32 CPU oracle: NOT a parity bar -- a top-1 router turns a
            # last-bit difference in a near-tied token into a different expert, and that image's logits then differ visibly
            # (tests/test_gpu_model.py attributes every such flip to a logit gap below the fp16 perturbation).  Per image:
            per_img = (gl - cpu_logits).abs().amax(dim=1)
            out["parity"]["model_logits_max_abs_diff_vs_cpu_fp32"] = float(per_img.max())
            out["parity"]["model_logits_per_image_median_max_abs_diff"] = float(per_img.median())
            out["parity"]["model_images_within_2e-2_of_cpu_fp32"] = f"{int((per_img <= 2e-2).sum())} / {per_img.numel()}"
            # ... and the same comparison with the routing taken out of it: the oracle run again on the decisions the GPU made
            # (every block's [T, 1] expert choice).  What is left is arithmetic (fp16 operands against fp32); the flipped tokens
            # are listed with the oracle's own logit gap between its choice and the GPU's -- a precision flip has a tiny one.
            from oracle import moe_oracle as mo
            try:      # a reporting leg: it must never cost the timed result its line
                tr = []
                with torch.no_grad():
                    fl = mo.vit_forward(images_cpu[: args.cpu_batch], sd_cpu, depth=12, num_heads=12, k=1, residual_moe=False,
                                        forced=forced, trace=tr)
                per_img_f = (gl - fl).abs().amax(dim=1)
                out["parity"]["same_routing"] = {
                    "model_logits_max_abs_diff": float(per_img_f.max()), "per_image_median": float(per_img_f.median()),
                    "images_within_2e-2": f"{int((per_img_f <= 2e-2).sum())} / {per_img_f.numel()}",
                    "tokens_routed_differently": f"{sum(t['flips'] for t in tr)} / {sum(t['tokens'] for t in tr)}",
                    "max_oracle_logit_gap_of_a_flipped_token": max((t["max_margin"] for t in tr), default=0.0)}
            except Exception as exc:
                out["parity"]["same_routing"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_ep:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
