"""GPU tests of the MoE backward (BASELINE cfg 5: SwitchGate, capacity_factor 1.0 with token dropping, aux
load-balance loss, fwd + bwd) against torch.autograd through the float64 oracle (oracle.moe_forward_diff)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import moe_oracle as mo  # noqa: E402
import slim_switch_moe_vit_amd as sm  # noqa: E402
from slim_switch_moe_vit_amd import ops  # noqa: E402

DEV = "cuda:0"


def _gen(s):
    return torch.Generator().manual_seed(s)


def _params(T, d, h, E, seed, skew=False):
    g = _gen(seed)
    x = torch.randn(T, d, generator=g)
    wg = torch.randn(E, d, generator=g) * 0.05
    bg = torch.zeros(E)
    if skew:
        bg[0] = 1.0
    w1 = torch.randn(E, h, d, generator=g) * 0.05
    b1 = torch.randn(E, h, generator=g) * 0.05
    w2 = torch.randn(E, d, h, generator=g) * 0.05
    b2 = torch.randn(E, d, generator=g) * 0.05
    gout = torch.randn(T, d, generator=g)
    return x, wg, bg, w1, b1, w2, b2, gout


from _mp import dtype_factor  # noqa: E402


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30)).item()


def test_backward_helper_kernels():
    g = _gen(0)
    counts = [100, 0, 37, 64, 1]
    E = len(counts)
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device=DEV)
    n, C = sum(counts), 200
    src = torch.randn(n + 3, C, generator=g).half().to(DEV)
    offp = ops.pad_offsets(offsets)
    assert offp.tolist() == [0, 128, 128, 192, 256, 320]
    Lp = ops.padded_len(n + 3, E)
    dst = ops.transpose_pad(src, offsets, offp, Lp)
    ref = torch.zeros(C, Lp, dtype=torch.float16)
    o, op = offsets.tolist(), offp.tolist()
    for e in range(E):
        ref[:, op[e]:op[e] + counts[e]] = src[o[e]:o[e + 1]].cpu().t()
    assert torch.equal(dst.cpu(), ref)
    cs = ops.group_colsum(src, offsets).cpu()
    for e in range(E):
        assert torch.allclose(cs[e], src[o[e]:o[e + 1]].float().sum(0).cpu(), atol=1e-2)
    # wgrad vs per-expert matmul
    P = torch.randn(n, 72, generator=g).half().to(DEV)
    Q = torch.randn(n, 136, generator=g).half().to(DEV)
    Lp2 = ops.padded_len(n, E)
    W = ops.grouped_wgrad(ops.transpose_pad(P, offsets, offp, Lp2), ops.transpose_pad(Q, offsets, offp, Lp2), offp).cpu()
    for e in range(E):
        r = P[o[e]:o[e + 1]].double().cpu().t() @ Q[o[e]:o[e + 1]].double().cpu()
        assert (W[e].double() - r).abs().max() < 1e-3 * max(1.0, r.abs().max().item())
    # gelu / gelu-grad epilogue / rowdot
    v = (torch.randn(64, 256, generator=g) * 2).to(DEV)
    assert torch.allclose(ops.gelu(v), torch.nn.functional.gelu(v), atol=2e-6)


@pytest.mark.parametrize("k", [1, 2])
def test_naive_gate_backward_matches_autograd(k):
    T, d, h, E = 600, 128, 256, 4
    x, wg, bg, w1, b1, w2, b2, gout = _params(T, d, h, E, seed=10 + k)
    mod = sm.FMoETransformerMLP(E, d, h, torch.nn.GELU(), top_k=k).to(DEV)
    with torch.no_grad():
        mod.gate.gate.weight.copy_(wg); mod.gate.gate.bias.copy_(bg)
        mod.experts.htoh4.weight.copy_(w1); mod.experts.htoh4.bias.copy_(b1)
        mod.experts.h4toh.weight.copy_(w2); mod.experts.h4toh.bias.copy_(b2)
    mod.train()
    xg = x.to(DEV).requires_grad_(True)
    out = mod(xg)
    (out * gout.to(DEV)).sum().backward()
    leaves = [t.clone().requires_grad_(True) for t in (x, wg, bg, w1, b1, w2, b2)]
    ro, _, plan = mo.moe_forward_diff(*leaves, k)
    (ro * gout.double()).sum().backward()
    assert np.array_equal(mod.last_plan[4].cpu().numpy(), plan.pos)
    assert _rel(out.detach().cpu(), ro.detach()) < 2e-3 * dtype_factor()
    got = [xg.grad, mod.gate.gate.weight.grad, mod.gate.gate.bias.grad, mod.experts.htoh4.weight.grad,
           mod.experts.htoh4.bias.grad, mod.experts.h4toh.weight.grad, mod.experts.h4toh.bias.grad]
    names = ["x", "wg", "bg", "w1", "b1", "w2", "b2"]
    for name, gt, rf in zip(names, got, leaves):
        if k == 1 and name in ("wg", "bg"):
            assert gt is None or float(gt.abs().max()) == 0.0  # top-1 naive gate: score == 1, no router gradient
            continue
        assert _rel(gt.cpu(), rf.grad) < 5e-3 * dtype_factor(), name


@pytest.mark.parametrize("T,d,h,wstd", [(1200, 128, 256, 0.05), (4096, 768, 3072, 0.02)])
def test_cfg5_switch_capacity_aux_backward(T, d, h, wstd):
    """SwitchGate, capacity_factor 1.0 (drops on the skewed router), aux loss in the objective: out, aux, dx, dWg,
    dbg, dW1, db1, dW2, db2 against float64 autograd through the oracle.  Second case = BASELINE cfg 5 at its own
    operator dims (ViT-B/16: d 768, h 3072, E 8, weights drawn like _init_vit_weights)."""
    E = 8
    x, wg, bg, w1, b1, w2, b2, gout = _params(T, d, h, E, seed=77, skew=True)
    if wstd != 0.05:
        w1, b1, w2, b2 = (t * (wstd / 0.05) for t in (w1, b1, w2, b2))
    mod = sm.FMoETransformerMLP(E, d, h, torch.nn.GELU(), top_k=1, gate="switch", capacity_factor=1.0).to(DEV)
    mod.gate.switch_eps = 0.0  # no jitter: oracle and device see the same logits
    with torch.no_grad():
        mod.gate.gate.weight.copy_(wg); mod.gate.gate.bias.copy_(bg)
        mod.experts.htoh4.weight.copy_(w1); mod.experts.htoh4.bias.copy_(b1)
        mod.experts.h4toh.weight.copy_(w2); mod.experts.h4toh.bias.copy_(b2)
    mod.train()
    xg = x.to(DEV).requires_grad_(True)
    out = mod(xg)
    aux = mod.gate.get_loss()
    ((out * gout.to(DEV)).sum() + 3.0 * aux).backward()
    cap = mo.switch_capacity(1.0, T, 1, E)
    leaves = [t.clone().requires_grad_(True) for t in (x, wg, bg, w1, b1, w2, b2)]
    ro, raux, plan = mo.moe_forward_diff(*leaves, 1, mo.GATE_SWITCH, cap)
    ((ro * gout.double()).sum() + 3.0 * raux).backward()
    assert (plan.idx_pruned < 0).sum() > 0
    assert np.array_equal(mod.last_plan[4].cpu().numpy(), plan.pos)
    assert abs(float(aux) - float(raux)) < 1e-4
    assert _rel(out.detach().cpu(), ro.detach()) < 2e-3 * dtype_factor()
    got = [xg.grad, mod.gate.gate.weight.grad, mod.gate.gate.bias.grad, mod.experts.htoh4.weight.grad,
           mod.experts.htoh4.bias.grad, mod.experts.h4toh.weight.grad, mod.experts.h4toh.bias.grad]
    for name, gt, rf in zip(["x", "wg", "bg", "w1", "b1", "w2", "b2"], got, leaves):
        assert _rel(gt.cpu(), rf.grad) < 5e-3 * dtype_factor(), name
    # dropped tokens get no expert gradient (only the router / aux path reaches them)
    dropped = torch.from_numpy(plan.idx_pruned < 0)
    assert _rel(xg.grad.cpu()[dropped], leaves[0].grad[dropped]) < 5e-3 * dtype_factor()


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 2e-2), (torch.float32, 1e-5)])
def test_group_colsum_many_chunks_and_empty_groups(dtype, tol):
    """Bias-gradient reduction over groups spanning several 512-row chunks, chunk-boundary sizes and empty groups;
    deterministic (two runs bit-identical)."""
    counts = [1500, 0, 512, 513, 1, 0, 1023, 7]
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device=DEV)
    n, C = sum(counts), 776
    src = (torch.randn(n + 5, C, generator=_gen(3)) * 0.5).to(dtype).to(DEV)   # rows past offsets[E] are ignored
    got = ops.group_colsum(src, offsets)
    again = ops.group_colsum(src, offsets)
    assert torch.equal(got, again)
    o = offsets.tolist()
    for e in range(len(counts)):
        ref = src[o[e]:o[e + 1]].double().sum(0).cpu()
        assert (got[e].cpu().double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("T", [900, 9000])
def test_expert_parallel_training_path_on_one_gpu(T):
    """fwd + bwd through the expert-parallel code path (count exchange, all-to-all-v each way and their adjoints,
    grouped GEMMs / wgrad with the group -> expert map) on a world of one rank == the single-rank training path."""
    import socket
    import torch.distributed as dist

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        d, h, E = 128, 256, 4      # (T = 9000: narrow weight gradients are cut into row pieces, on separate row ranges too)
        x, wg, bg, w1, b1, w2, b2, gout = _params(T, d, h, E, seed=5, skew=True)
        from slim_switch_moe_vit_amd import ep
        import contextlib
        grads = {}
        # "ep": the capacity gate's STATIC exchange (fixed slots, counts in-band, no host round trip: SURVEY.md 8e on cfg 5), with
        # autograd around it; "ep_counted": the count exchange + all-to-all-v
        for mode in ("single", "ep", "ep_counted"):
            mod = sm.FMoETransformerMLP(E, d, h, torch.nn.GELU(), top_k=1, gate="switch", capacity_factor=1.0).to(DEV)
            mod.gate.switch_eps = 0.0
            with torch.no_grad():
                mod.gate.gate.weight.copy_(wg); mod.gate.gate.bias.copy_(bg)
                mod.experts.htoh4.weight.copy_(w1); mod.experts.htoh4.bias.copy_(b1)
                mod.experts.h4toh.weight.copy_(w2); mod.experts.h4toh.bias.copy_(b2)
            mod.train()
            mod.force_ep = mode != "single"
            xg = x.to(DEV).requires_grad_(True)
            with (ep.dynamic_only() if mode == "ep_counted" else contextlib.nullcontext()):
                assert mode == "single" or (ep.static_kind(mod, torch.float16) == "capacity") == (mode == "ep")
                out = mod(xg)
                ((out * gout.to(DEV)).sum() + 2.0 * mod.gate.get_loss()).backward()
            grads[mode] = [out.detach(), xg.grad] + [p.grad for p in mod.parameters()]
        ep.check_static_overflow(flush=True)
        for a, b in zip(grads["single"], grads["ep"]):
            assert _rel(b.cpu(), a.cpu()) < 1e-3
        worst = max(_rel(b.cpu(), a.cpu()) for a, b in zip(grads["ep_counted"], grads["ep"]))
        print("static vs counted exchange, training step: worst rel-L2", worst,
              "bit-identical" if all(torch.equal(a, b) for a, b in zip(grads["ep_counted"], grads["ep"])) else "")
        # the same rows in the same groups through the same kernels: bit-identical -- unless the weight gradients are cut into pieces,
        # whose number comes from the routed rows here and from the received rows there (another, equally valid summation order)
        assert worst == 0.0 if T == 900 else worst < 1e-5
    finally:
        dist.destroy_process_group()


def test_block_level_training_step_runs():
    m = sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=10).to(DEV).train()
    opt = torch.optim.SGD(m.parameters(), lr=0.01)
    x = torch.randn(4, 3, 224, 224, device=DEV)
    y = torch.randint(0, 10, (4,), device=DEV)
    loss0 = None
    for _ in range(3):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(x), y)
        loss.backward()
        opt.step()
        loss0 = loss0 or float(loss)
    assert float(loss) < loss0
    assert m.blocks[0].mlp.experts.htoh4.weight.grad is not None


def _two_rank_train_worker(rank, world, port, q, gate="naive"):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks on cuda:0; ep._a2a stages through the host
    try:
        d, h, E = 128, 256, 4
        E_local = E // world
        T = [700, 433]
        _, wg, bg, w1, b1, w2, b2, _ = _params(1, d, h, E, seed=21, skew=(gate == "switch"))
        xs = [torch.randn(T[r], d, generator=_gen(30 + r)) for r in range(world)]
        gs = [torch.randn(T[r], d, generator=_gen(40 + r)) for r in range(world)]

        def load(mod, sl):
            with torch.no_grad():
                mod.gate.gate.weight.copy_(wg); mod.gate.gate.bias.copy_(bg)
                mod.experts.htoh4.weight.copy_(w1[sl]); mod.experts.htoh4.bias.copy_(b1[sl])
                mod.experts.h4toh.weight.copy_(w2[sl]); mod.experts.h4toh.bias.copy_(b2[sl])
            return mod.to(DEV).train()

        def make(n_exp, ws):
            if gate == "switch":   # BASELINE cfg 5's gate: capacity 1.0 of the rank's own batch, tokens dropped -> the STATIC exchange
                m = sm.FMoETransformerMLP(n_exp, d, h, torch.nn.GELU(), top_k=1, gate="switch", capacity_factor=1.0, world_size=ws)
                m.gate.switch_eps = 0.0
                return m
            return sm.FMoETransformerMLP(n_exp, d, h, torch.nn.GELU(), top_k=2, world_size=ws)

        # reference: one rank holding all experts, both shards (expert gradients add up over the shards)
        full = load(make(E, 1), slice(0, E))
        ref_dx, ref_out, ref_gate = {}, {}, {}
        for r in range(world):
            full.gate.gate.weight.grad = None
            full.gate.gate.bias.grad = None
            xg = xs[r].to(DEV).requires_grad_(True)
            out = full(xg)
            (out * gs[r].to(DEV)).sum().backward()
            ref_dx[r], ref_out[r] = xg.grad, out.detach()
            ref_gate[r] = (full.gate.gate.weight.grad.clone(), full.gate.gate.bias.grad.clone())
        # this rank of the expert-parallel pair
        sl = slice(rank * E_local, (rank + 1) * E_local)
        part = load(make(E_local, world), sl)
        xg = xs[rank].to(DEV).requires_grad_(True)
        from slim_switch_moe_vit_amd import ep
        if gate == "naive_spec":       # what engine.train_one_epoch switches on: speculative slots for the training forward too
            holder = torch.nn.Module()
            holder.mlp = part
            ep.set_speculative(holder, 3.0, train=True)
        kind = ep.static_kind(part, torch.float16)
        assert kind == {"switch": "capacity", "naive_spec": "speculative", "naive": None}[gate]
        out = part(xg)
        (out * gs[rank].to(DEV)).sum().backward()
        ep.check_static_overflow(flush=True)
        dropped = int((part.last_plan[5] < 0).sum())
        errs = {
            "out": _rel(out.detach().cpu(), ref_out[rank].cpu()),
            "dx": _rel(xg.grad.cpu(), ref_dx[rank].cpu()),
            "dWg": _rel(part.gate.gate.weight.grad.cpu(), ref_gate[rank][0].cpu()),
            "dW1": _rel(part.experts.htoh4.weight.grad.cpu(), full.experts.htoh4.weight.grad[sl].cpu()),
            "db1": _rel(part.experts.htoh4.bias.grad.cpu(), full.experts.htoh4.bias.grad[sl].cpu()),
            "dW2": _rel(part.experts.h4toh.weight.grad.cpu(), full.experts.h4toh.weight.grad[sl].cpu()),
            "db2": _rel(part.experts.h4toh.bias.grad.cpu(), full.experts.h4toh.bias.grad[sl].cpu()),
        }
        if gate == "switch":
            assert dropped > 0, "the capacity must have dropped tokens for this test to mean anything"
        q.put((rank, errs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gate", ["naive", "switch", "naive_spec"])
def test_two_expert_parallel_ranks_training_step_on_one_gpu(gate):
    """fwd + bwd of the MoE operator across TWO expert-parallel ranks (two processes on one GPU, gloo transport):
    outputs, dx and the router gradient of every rank match the single-rank operator on that rank's tokens, and each
    rank's expert weight / bias gradients are the single-rank gradients (summed over both ranks' tokens) of the
    experts it owns.  "naive": top-2 NaiveGate on the counted exchange; "switch": cfg 5's capacity gate on the static exchange
    (fixed slots, counts in-band, ragged ranks: 700 / 433 rows against slots agreed for the larger one); "naive_spec": the top-2
    NaiveGate on speculative slots (roomy: nothing overflows, which the flushed watch confirms on both ranks)."""
    import socket
    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_train_worker, args=(r, 2, port, q, gate)) for r in range(2)]
    for p in procs:
        p.start()
    from _mp import join_or_kill
    join_or_kill(procs, 300)
    got = dict(q.get(timeout=10) for _ in range(2))
    assert sorted(got) == [0, 1]
    for rank, errs in got.items():
        for name, err in errs.items():
            assert err < 2e-3, (rank, name, err)


@pytest.mark.parametrize("T,E,d,dtype", [(1, 4, 64, torch.float32), (5000, 8, 768, torch.float32), (1537, 16, 192, torch.float16),
                                          (700, 3, 1024, torch.bfloat16)])
def test_gate_wgrad_matches_matmul(T, E, d, dtype):
    g = _gen(T + E)
    dl = torch.randn(T, E, generator=g)
    x = torch.randn(T, d, generator=g).to(dtype)
    got = ops.gate_wgrad(dl.to(DEV), x.to(DEV)).cpu().double()
    ref = dl.double().t() @ x.double()
    assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()) * (T ** 0.5)
    assert torch.equal(ops.gate_wgrad(dl.to(DEV), x.to(DEV)).cpu().double(), got)   # deterministic


@pytest.mark.parametrize("counts,R1,R2", [([64, 128], 256, 256), ([100, 0, 37, 64, 1], 72, 136), ([300, 5, 777], 768, 256),
                                          ([1000, 900, 1100, 950], 768, 3072),
                                          # eight experts at ViT-B's weight shapes: 320-row tiles (240 tiles = one round of
                                          # workgroups instead of 288 = two), the last one 192 rows; then dW2's shape, which
                                          # ops computes transposed; then 320-row tiles with ragged row AND column tails
                                          ([250, 0, 300, 310, 280, 333, 64, 201], 3072, 768),
                                          ([250, 0, 300, 310, 280, 333, 64, 201], 768, 3072),
                                          ([130, 70, 0, 65, 129, 64, 1, 200], 3064, 760)])
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.5e-2)])
def test_grouped_wgrad_rows_matches_per_expert_matmul(counts, R1, R2, dtype, tol):
    """Token-major wgrad (transposed LDS reads, no transposed operand copies): ragged and empty experts, row counts
    that are not multiples of the 64-token K-tile (the tail reads the zero page), output tiles with row / column tails;
    and the same numbers as the transposed-copy path."""
    E = len(counts)
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device=DEV)
    n = sum(counts)
    g = _gen(n + R1)
    P = (torch.randn(n + 3, R1, generator=g) * 0.5).to(dtype).to(DEV)   # rows past offsets[E] exist and must not be read into a sum
    Q = (torch.randn(n + 3, R2, generator=g) * 0.5).to(dtype).to(DEV)
    got = ops.grouped_wgrad_rows(P, Q, offsets).cpu().double()
    o = offsets.tolist()
    for e in range(E):
        ref = P[o[e]:o[e + 1]].double().t().cpu() @ Q[o[e]:o[e + 1]].double().cpu()
        scale = max(1.0, ref.abs().max().item())
        assert (got[e] - ref).abs().max().item() <= tol * scale, (e, float((got[e] - ref).abs().max()))
    offp = ops.pad_offsets(offsets)
    Lp = ops.padded_len(n + 3, E)
    old = ops.grouped_wgrad(ops.transpose_pad(P, offsets, offp, Lp), ops.transpose_pad(Q, offsets, offp, Lp), offp).cpu().double()
    assert (got - old).abs().max().item() <= 1e-3 * max(1.0, old.abs().max().item())


def test_grouped_wgrad_rows_tail_rows_never_meet_foreign_bytes():
    """Rows past an expert's range (the tail of its last 64-token K-tile) read a zero page for BOTH operands.  Round 1
    substituted the first bytes of Q for the second operand and relied on 0 * x = 0: with an Inf there (an f16 overflow
    in expert 0's rows) every other expert's gradient turned NaN."""
    counts = [70, 37, 100, 5]
    E, R1, R2 = len(counts), 136, 264
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device=DEV)
    n = sum(counts)
    g = _gen(4242)
    P = (torch.randn(n, R1, generator=g) * 0.5).half()
    Q = (torch.randn(n, R2, generator=g) * 0.5).half()
    Q[0, :16] = float("inf")
    P[0, :16] = float("inf")
    got = ops.grouped_wgrad_rows(P.to(DEV), Q.to(DEV), offsets).cpu().double()
    o = offsets.tolist()
    for e in range(1, E):   # expert 0 owns the Inf row; everyone else must be untouched by it
        ref = P[o[e]:o[e + 1]].double().t() @ Q[o[e]:o[e + 1]].double()
        assert torch.isfinite(got[e]).all(), e
        assert (got[e] - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item()), e


@pytest.mark.parametrize("shape", [(3, 128, 192), (8, 768, 3072), (1, 64, 64)])
@pytest.mark.parametrize("src,dst", [(torch.float32, torch.float16), (torch.float32, torch.bfloat16), (torch.float16, torch.float16),
                                     (torch.float32, torch.float32), (torch.bfloat16, torch.float32)])
def test_transpose_cast_is_the_transposed_rounded_copy(shape, src, dst):
    """smoe_transpose_cast == .transpose(1, 2).to(dst) bit for bit (the dgrad GEMMs' weight operand, and the way back of the
    weight gradient that is computed transposed)."""
    w = (torch.randn(shape, generator=_gen(sum(shape))) * 0.3).to(src).to(DEV)
    got = ops.transpose_cast(w, dst)
    assert got.shape == (shape[0], shape[2], shape[1]) and got.is_contiguous()
    assert torch.equal(got, w.transpose(1, 2).to(dst).contiguous())
    with pytest.raises(ValueError):
        ops.transpose_cast(w[:, :, :-8].contiguous(), dst)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("counts,K,N", [([300, 0, 517, 64], 128, 264), ([700, 650, 1, 900, 0, 320, 321, 1280], 768, 3072)])
def test_grouped_gemm_gelu_keep_matches_the_two_step_form(counts, K, N, dtype):
    """One epilogue, two stores (H and gelu(H)) == SMOE_EPI_NONE followed by smoe_gelu: H bit for bit (same accumulators, same
    rounding), gelu(H) within the 16-bit rounding of the two GELU forms (the fused epilogue applies GELU to the f32 value, the
    pass to the rounded H)."""
    E = len(counts)
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device=DEV)
    n = sum(counts)
    g = _gen(n + K)
    A = (torch.randn(n, K, generator=g) * 0.5).to(dtype).to(DEV)
    W = (torch.randn(E, N, K, generator=g) * 0.05).to(dtype).to(DEV)
    b = (torch.randn(E, N, generator=g) * 0.1).to(DEV)
    pre, act = ops.grouped_gemm_gelu_keep(A, W, b, offsets)
    ref_pre = ops.grouped_gemm(A, W, b, offsets, ops.EPI_NONE, dtype, variant=10)
    assert torch.equal(pre, ref_pre)
    ref_act = torch.nn.functional.gelu(ref_pre.float()).to(dtype)
    tol = 2 ** -10 if dtype == torch.float16 else 2 ** -7
    assert (act.float() - ref_act.float()).abs().max().item() <= tol * max(1.0, float(ref_act.float().abs().max()))
    fused = ops.grouped_gemm(A, W, b, offsets, ops.EPI_GELU, dtype, variant=10)
    assert torch.equal(act, fused)   # the same epilogue arithmetic as the inference kernel's


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_backward_kernels_on_separate_row_ranges_never_touch_the_padding(dtype):
    """group_end (the slots of the static expert exchange in training): weight gradient, bias-gradient column sums and the
    H / gelu(H) forward take row ranges [starts[g], ends[g]) with gaps between them -- the same bits as on the compacted rows, with NaN in
    every row outside the ranges (padding and header rows of a received buffer are never read)."""
    counts = [700, 0, 513, 64, 1, 320]                   # rows in use per slot
    slot = [800, 7, 513, 100, 130, 321]                  # slot sizes (payload + header rows): gaps of 100, 7, 0, 36, 129, 1 rows
    G, K, N = len(counts), 128, 256
    starts = np.concatenate([[0], np.cumsum(slot)])[:-1].astype(np.int32)
    ends = (starts + np.array(counts)).astype(np.int32)
    n_pad = int(sum(slot))
    g = _gen(77)
    X = (torch.randn(n_pad, K, generator=g) * 0.5).to(dtype)
    dY = (torch.randn(n_pad, N, generator=g) * 0.5).to(dtype)
    keep = torch.zeros(n_pad, dtype=torch.bool)
    for s0, e0 in zip(starts, ends):
        keep[s0:e0] = True
    X[~keep] = float("nan")
    dY[~keep] = float("nan")
    Xc, dYc = X[keep].contiguous().to(DEV), dY[keep].contiguous().to(DEV)
    offs_c = torch.tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device=DEV)
    st, en = torch.from_numpy(starts).to(DEV), torch.from_numpy(ends).to(DEV)
    X, dY = X.to(DEV), dY.to(DEV)
    # weight gradient (both orientations: the second is computed transposed where that needs fewer rounds)
    for P, Q, Pc, Qc in ((dY, X, dYc, Xc), (X, dY, Xc, dYc)):
        got = ops.grouped_wgrad_rows(P, Q, st, group_end=en)
        ref = ops.grouped_wgrad_rows(Pc, Qc, offs_c)
        assert torch.isfinite(got).all() and torch.equal(got, ref)
    # column sums
    got = ops.group_colsum(dY, st, en)
    assert torch.isfinite(got).all() and torch.equal(got, ops.group_colsum(dYc, offs_c))
    # the training forward's first linear: H and gelu(H) inside the ranges
    W = (torch.randn(G, N, K, generator=g) * 0.05).to(dtype).to(DEV)
    b = (torch.randn(G, N, generator=g) * 0.1).to(DEV)
    pre, act = ops.grouped_gemm_gelu_keep(X, W, b, st, group_end=en)
    pre_c, act_c = ops.grouped_gemm_gelu_keep(Xc, W, b, offs_c)
    km = keep.to(DEV)
    assert torch.equal(pre[km], pre_c) and torch.equal(act[km], act_c)


@pytest.mark.parametrize("T,E", [(1, 8), (1000, 8), (4097, 16), (333, 27)])
def test_switch_gate_bwd_matches_autograd_of_softmax_gather_and_linear_aux(T, E):
    """smoe_switch_gate_bwd == d/dlogits of  sum_t dscore[t] softmax(logits)[t, idx[t]] + sum_{t,e} coef[e] softmax(logits)[t, e]
    (float64 autograd), with and without either term; entries with idx = -1 select nothing."""
    g = _gen(T * 31 + E)
    logits = torch.randn(T, E, generator=g) * 2
    idx = torch.randint(0, E, (T,), generator=g)
    if T > 10:
        idx[::9] = -1
    dscore = torch.randn(T, generator=g)
    coef = torch.randn(E, generator=g) * 0.1
    probs = torch.softmax(logits, -1)
    for use_ds, use_cf in ((True, True), (True, False), (False, True)):
        l64 = logits.double().requires_grad_(True)
        p64 = torch.softmax(l64, -1)
        obj = torch.zeros((), dtype=torch.float64)
        if use_ds:
            sel = idx >= 0
            obj = obj + (dscore.double()[sel] * p64[sel, idx[sel]]).sum()
        if use_cf:
            obj = obj + (p64 * coef.double()).sum()
        obj.backward()
        got = ops.switch_gate_bwd(probs.to(DEV), idx.to(DEV), dscore.to(DEV) if use_ds else None, coef.to(DEV) if use_cf else None)
        assert (got.cpu().double() - l64.grad).abs().max().item() <= 2e-6 * max(1.0, float(l64.grad.abs().max()))


@pytest.mark.parametrize("T,E", [(1, 8), (1000, 8), (25216, 8), (4097, 16), (333, 27), (5000, 256)])
def test_switch_aux_kernel_matches_the_formula(T, E):
    """smoe_switch_aux: aux = E sum_e frac_e prob_e and coef[e] = E frac_e / kept (fmoe.gates.SwitchGate) against the
    float64 formula, with dropped tokens (counts that sum to less than T) and an all-dropped plan (kept clamps to 1)."""
    g = _gen(T + 7 * E)
    probs = torch.softmax(torch.randn(T, E, generator=g), -1)
    idx = torch.randint(0, E, (T,), generator=g)
    counts = torch.bincount(idx, minlength=E).to(torch.int32)
    counts = (counts.float() * 0.8).floor().to(torch.int32)          # a capacity dropped some
    for cnt in (counts, torch.zeros_like(counts)):
        aux, coef = ops.switch_aux(probs.to(DEV), cnt.to(DEV))
        kept = max(int(cnt.sum()), 1)
        frac = cnt.double() / kept
        ref_aux = E * (frac * (probs.double().sum(0) / kept)).sum()
        ref_coef = E * frac / kept
        assert abs(float(aux) - float(ref_aux)) <= 1e-5 * max(1.0, abs(float(ref_aux)))
        assert (coef.cpu().double() - ref_coef).abs().max().item() <= 1e-6 * max(1.0, float(ref_coef.abs().max()))
        again = ops.switch_aux(probs.to(DEV), cnt.to(DEV))
        assert torch.equal(again[0], aux) and torch.equal(again[1], coef)                      # deterministic


def test_switch_gate_bwd_scales_the_aux_coefficients_on_the_device():
    g = _gen(77)
    T, E = 3000, 8
    probs = torch.softmax(torch.randn(T, E, generator=g), -1).to(DEV)
    idx = torch.randint(0, E, (T,), generator=g).to(DEV)
    ds = torch.randn(T, generator=g).to(DEV)
    coef = (torch.randn(E, generator=g) * 0.1).to(DEV)
    scale = torch.tensor([0.37], device=DEV)
    a = ops.switch_gate_bwd(probs, idx, ds, coef, scale)
    b = ops.switch_gate_bwd(probs, idx, ds, (coef * scale).contiguous())
    assert (a - b).abs().max().item() <= 1e-7
    assert torch.equal(ops.switch_gate_bwd(probs, idx, ds, coef, None), ops.switch_gate_bwd(probs, idx, ds, coef))


@pytest.mark.parametrize("T,E,d,dtype", [(1, 4, 64, torch.float32), (5000, 8, 768, torch.float32), (1537, 16, 192, torch.float16),
                                          (25216, 1, 768, torch.float32)])
def test_gate_wgrad_bias_column_is_the_column_sum_of_dl(T, E, d, dtype):
    g = _gen(T + E + 1)
    dl = torch.randn(T, E, generator=g)
    x = torch.randn(T, d, generator=g).to(dtype)
    dw, db = ops.gate_wgrad(dl.to(DEV), x.to(DEV), want_bias=True)
    assert torch.equal(dw, ops.gate_wgrad(dl.to(DEV), x.to(DEV)))                              # the weights' pass is unchanged
    ref = dl.double().sum(0)
    assert (db.cpu().double() - ref).abs().max().item() <= 1e-5 * max(1.0, float(ref.abs().max())) * (T ** 0.5)
    assert torch.equal(ops.gate_wgrad(dl.to(DEV), x.to(DEV), want_bias=True)[1], db)


@pytest.mark.parametrize("xdt,odt", [(torch.float32, torch.float16), (torch.float16, torch.float16), (torch.float32, torch.float32)])
def test_scatter_rows_clears_the_slots_no_token_maps_to(xdt, odt):
    """``zero_fill``: slots with pos < 0 (past the kept count under a capacity) become zero rows in the scatter pass itself; without
    it they are left alone."""
    g = _gen(5)
    T, d, k = 700, 192, 1
    x = torch.randn(T, d, generator=g).to(xdt).to(DEV)
    pos = torch.randperm(T, generator=g)
    pos[500:] = -1
    pos = pos.to(DEV)
    junk = torch.full((T, d), 7.0, dtype=odt, device=DEV)
    out = ops.scatter_rows(x, pos, k, odt, out=junk.clone(), zero_fill=True)
    ref = torch.zeros(T, d, dtype=odt, device=DEV)
    ref[:500] = x[pos[:500]].to(odt)
    assert torch.equal(out, ref)
    kept = ops.scatter_rows(x, pos, k, odt, out=junk.clone())
    assert torch.equal(kept[:500], ref[:500]) and torch.equal(kept[500:], junk[500:])
    scale = torch.rand(T, generator=g).to(DEV)
    sc = ops.scatter_rows(x, pos, k, odt, zero_fill=True, scale=scale)
    assert torch.equal(sc[500:], ref[500:])
    assert torch.allclose(sc[:500].float(), (x[pos[:500]].float() * scale[pos[:500], None]), rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("E,Z,adt", [(8, 1, torch.float16), (4, 2, torch.float16), (8, 2, torch.float32)])
def test_zero_group_fold_equals_the_index_add_composition(E, Z, adt):
    """smoe_zero_group_fold against the torch composition it replaces (rank-1 index_add_ into dW2, index_add of the column sums),
    two zero groups aimed at the same expert, an empty zero group."""
    g = _gen(E * 10 + Z)
    d, h = 192, 768
    counts = [int(c) for c in torch.randint(50, 300, (E,), generator=g)] + ([400, 0] if Z == 2 else [400])
    G = E + Z
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device=DEV)
    n = sum(counts)
    A = torch.randn(n, h, generator=g).to(adt).to(DEV)
    cs2 = torch.randn(G, d, generator=g).to(DEV)
    cs1 = torch.randn(G, h, generator=g).to(DEV)
    if Z == 2:
        cs2[G - 1] = 0
        cs1[G - 1] = 0                                              # (the column sums of an empty group)
    tgt = torch.full((Z,), 3, dtype=torch.int64) if Z == 2 else torch.tensor([E - 1])
    gmap = torch.cat((torch.arange(E), tgt)).to(torch.int32).to(DEV)
    dW2 = torch.randn(E, d, h, generator=g).to(DEV)
    ref = dW2.clone()
    first = offsets[E:G].long().clamp(max=n - 1)
    a_rows = A.index_select(0, first).float()
    ref.index_add_(0, tgt.to(DEV), cs2[E:, :, None] * a_rows[:, None, :])
    ref_b2 = cs2[:E].index_add(0, tgt.to(DEV), cs2[E:])
    ref_b1 = cs1[:E].index_add(0, tgt.to(DEV), cs1[E:])
    db2, db1 = ops.zero_group_fold(cs2, cs1, A, offsets, gmap, E, dW2)
    assert torch.allclose(dW2, ref, rtol=1e-6, atol=1e-6)
    assert torch.allclose(db2, ref_b2, rtol=1e-6, atol=1e-6) and torch.allclose(db1, ref_b1, rtol=1e-6, atol=1e-6)
    only2, none1 = ops.zero_group_fold(cs2, None, A, offsets, gmap, E, dW2.clone(), want_b1=False)
    assert none1 is None and torch.equal(only2, db2)


def test_fused_adamw_step_keeps_the_16_bit_weight_images_current():
    """optim.AdamW writes the 16-bit operand image of every weight it updates in the same pass (smoe_adamw_step_multi's `shadow`
    table): after the step the modules' caches serve the SAME tensors, already equal to the rounded new weights -- no cast pass --
    and the transposed images are re-made from them.  A step skipped by found_inf leaves weights and images untouched."""
    from slim_switch_moe_vit_amd import optim as smo
    from slim_switch_moe_vit_amd._cache import param_version
    torch.manual_seed(0)
    mod = sm.FMoETransformerMLP(4, 64, 128, torch.nn.GELU(), top_k=1).to(DEV).train()
    lin = torch.nn.Linear(64, 64).to(DEV)
    from slim_switch_moe_vit_amd import vit
    hc = vit._HalfCache()
    opt = smo.AdamW(list(mod.parameters()) + list(lin.parameters()), lr=1e-2, weight_decay=0.05)
    x = torch.randn(300, 64, generator=_gen(1)).to(DEV)

    def backward():
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16):
            y = mod(x.requires_grad_(True))
        (y.float().square().mean() + (lin(x) ** 2).mean()).backward()
    backward()
    ex = mod.experts
    w1, w2 = ex.htoh4.weight, ex.h4toh.weight
    img1, img2, imgl = ex.htoh4.weight_as(torch.float16), ex.h4toh.weight_as(torch.float16), hc.get(lin.weight)
    t1 = ex.htoh4.weight_t_as(torch.float16)
    before = w1.detach().clone()
    opt.step()
    assert not torch.equal(w1, before)
    for w, img, cache, key in ((w1, img1, ex.htoh4._shadow, (torch.float16, w1.device)),
                               (w2, img2, ex.h4toh._shadow, (torch.float16, w2.device)), (lin.weight, imgl, hc, id(lin.weight))):
        assert cache.peek(key, param_version(w)) is img, "the image is current for the new version, and the same tensor"
        assert torch.equal(img, w.detach().half()), "... holding the rounded NEW weights"
    assert ex.htoh4.weight_as(torch.float16) is img1 and hc.get(lin.weight) is imgl
    t1_new = ex.htoh4.weight_t_as(torch.float16)
    assert t1_new is not t1 and torch.equal(t1_new, w1.detach().half().transpose(1, 2).contiguous())
    # a skipped step (non-finite gradients): nothing moves, the images stay what the weights are
    backward()
    snap = w1.detach().clone()
    opt.step(found_inf=torch.ones(1, device=DEV))
    assert torch.equal(w1, snap) and torch.equal(ex.htoh4.weight_as(torch.float16), snap.half())
    # switched off: the images go stale with the step and are re-cast on their next use (same values)
    smo.SHADOW_STEP = False
    try:
        backward()
        opt.step()
        assert ex.htoh4._shadow.peek((torch.float16, w1.device), param_version(w1)) is None
        assert torch.equal(ex.htoh4.weight_as(torch.float16), w1.detach().half())
    finally:
        smo.SHADOW_STEP = True
