"""The reference's multi-GPU mode is plain data parallelism: main.py:611 wraps the whole model in DistributedDataParallel, every rank
holds all experts (world_size = 1 inside FMoETransformerMLP) and the gradients are averaged over the ranks.  This test runs that
mode on the library's training path: W = 2 processes sharing cuda:0, gloo for the gradient all-reduce (RCCL refuses two ranks on one
device), the reference's own model family.

  * gradients of one step under DDP (each rank its half of the batch) = gradients of the whole batch in one process
    (the mean over the global batch is the mean of the ranks' means), to 16-bit rounding;
  * engine.train_one_epoch on the DDP-wrapped model -- own AdamW + NativeScaler, clipping -- leaves bit-identical parameters on
    both ranks (what DDP promises) that differ from the initial ones, with a finite loss."""
import os
import socket
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _mp import dtype_factor, join_or_kill  # noqa: E402
import slim_switch_moe_vit_amd as sm  # noqa: E402
from test_gpu_model import _init  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# (top-2 gates: with top-1 the NaiveGate's score is the constant 1, the router weights get no gradient and stock DDP stops at the
#  second step -- in the reference as here: SURVEY.md "DDP + top-1 NaiveGate")
MODELS = ("resmoe_tiny_patch16_224_expert8", "moe_tiny_patch16_224_expert8")


def _port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _build(name):
    torch.manual_seed(0)
    kw = dict(starting_threshold=0.5, target_threshold=0.5) if name.startswith("resmoe") else {}
    model = sm.create_model(name, num_classes=10, depth=2, drop_path_rate=0.0, **kw)
    model = _init(model, 31)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for blk in model.blocks:
            for gt in (getattr(blk, "dense_gate", None), getattr(blk, "moe_gate", None)):
                if gt is not None:
                    gt.head[1].weight.normal_(0, 0.5, generator=g)
    return model.to(DEV)


def _data(n):
    g = torch.Generator().manual_seed(77)
    return torch.randn(n, 3, 224, 224, generator=g), torch.randint(0, 10, (n,), generator=g)


def _grads(model, images, target):
    model.train()
    for p in model.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.float16):
        loss = torch.nn.functional.cross_entropy(model(images.to(DEV)), target.to(DEV))
    (loss * 256.0).backward()                                    # (a fixed loss scale: fp16 gradients of a tiny loss underflow)
    return float(loss.detach()), {n: (p.grad.detach().float().cpu() / 256.0) for n, p in model.named_parameters() if p.grad is not None}


def _ddp_worker(rank, world, port, name, q):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        images, target = _data(8)
        per = images.shape[0] // world
        mine = slice(rank * per, (rank + 1) * per)
        model = _build(name)
        assert sm.ddp_ignore_expert_parameters(model) == []       # all experts on every rank: ordinary data parallelism
        ddp = DDP(model, device_ids=[0])
        loss, grads = _grads(ddp, images[mine], target[mine])
        grads = {n.replace("module.", "", 1): g for n, g in grads.items()}
        # the harness on the wrapped model: three steps on this rank's share of three different batches
        opt = sm.AdamW(ddp.parameters(), lr=1e-3, weight_decay=0.05)
        scaler = sm.NativeScaler()
        before = {n: p.detach().clone() for n, p in model.named_parameters()}
        loader = []
        for s in range(3):
            gi = torch.Generator().manual_seed(100 + s)
            xb, yb = torch.randn(8, 3, 224, 224, generator=gi), torch.randint(0, 10, (8,), generator=gi)
            loader.append((xb[mine], yb[mine]))
        stats = sm.train_one_epoch(ddp, torch.nn.CrossEntropyLoss(), loader, opt, DEV, 0, scaler, max_norm=1.0)
        torch.cuda.synchronize()
        moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in model.named_parameters())
        params = {n: p.detach().float().cpu().numpy() for n, p in model.named_parameters()}
        # (numpy arrays travel through the queue by value; torch tensors by file descriptor, which dies with this process)
        q.put((rank, loss, {n: g.numpy() for n, g in grads.items()}, stats["loss"], moved, len(before), params))
    except BaseException as exc:      # the parent must not sit out its timeout when a rank dies
        q.put((rank, "error", repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", MODELS)
def test_data_parallel_training_under_ddp_matches_the_whole_batch(name):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r = q.get(timeout=200)
            assert r[1] != "error", f"rank {r[0]}: {r[2]}"
            got[r[0]] = r[1:]
    except BaseException:
        for p in procs:               # no rank stays on the GPU behind a failed test
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(10)
        raise
    join_or_kill(procs, 120)
    # the whole batch in this process
    images, target = _data(8)
    model = _build(name)
    loss_ref, ref = _grads(model, images, target)
    loss0, g0, tl0, moved0, n_params, p0 = got[0]
    loss1, g1, tl1, moved1, _, p1 = got[1]
    g0, g1 = ({n: torch.from_numpy(v) for n, v in g.items()} for g in (g0, g1))
    p0, p1 = ({n: torch.from_numpy(v) for n, v in p.items()} for p in (p0, p1))
    assert abs(0.5 * (loss0 + loss1) - loss_ref) <= 2e-3 * max(1.0, abs(loss_ref)), (loss0, loss1, loss_ref)
    assert set(g0) == set(ref) == set(g1)
    num = den = 0.0
    worst = ("", 0.0)
    for n in ref:
        assert torch.equal(g0[n], g1[n]), f"{n}: DDP left different gradients on the two ranks"
        num += float((g0[n].double() - ref[n].double()).pow(2).sum())
        den += float(ref[n].double().pow(2).sum())
        rel = float((g0[n].double() - ref[n].double()).norm() / ref[n].double().norm().clamp(min=1e-12))
        if rel > worst[1] and float(ref[n].abs().max()) > 1e-4:
            worst = (n, rel)
    rel_all = (num / max(den, 1e-30)) ** 0.5
    print(f"{name}: DDP (2 ranks x 4 images) vs one process (8 images): gradient rel-L2 {rel_all:.2e} over all parameters, worst "
          f"tensor {worst[0]} {worst[1]:.2e}; train_one_epoch loss {tl0:.4f} / {tl1:.4f}, parameters moved {moved0} of {n_params}")
    # 16-bit operands, another summation order over the batch.  Measured (f16): 1.6e-6 / 1.2e-5 over all parameters, worst tensor
    # (patch_embed.proj.weight) 2.7e-4 / 8.7e-5
    assert rel_all <= 4e-5 * dtype_factor() and worst[1] <= 8e-4 * dtype_factor(), (rel_all, worst)
    # the harness under DDP: finite, the parameters moved, and both ranks hold the SAME parameters afterwards
    assert tl0 == tl0 and tl1 == tl1 and moved0 == moved1 and moved0 >= n_params // 2
    for n in p0:
        assert torch.equal(p0[n], p1[n]), f"{n}: the ranks' parameters diverged under DDP"


# ---------------------------------------------------------------------------------------------------------------------------------------
# Expert parallelism UNDER DistributedDataParallel: FMoETransformerMLP(world_size = W) holds a slice of the experts per rank; the shared
# parameters (attention, norms, router, head) are data parallel, the expert slices are not (ddp_ignore_expert_parameters: FastMoE ships
# DistributedGroupedDataParallel for this)
def _ddp_ep_worker(rank, world, port, q):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (the BASELINE factories take the GLOBAL expert count and give every rank its share; the reference's own keep FastMoE's
        #  meaning: experts per rank)
        name, E = "moe_tiny_patch16_224_expert4_top1", 4
        E_local = E // world
        # SwitchGate (cfg 5's): its score carries a gradient to the router -- with the top-1 NaiveGate the router weights get none and
        # DDP's reducer never finishes their bucket (SURVEY.md "DDP + top-1 NaiveGate") -- and its capacity puts the expert-parallel
        # model on the static exchange in training
        kw = dict(num_classes=10, depth=2, drop_path_rate=0.0, gate="switch", capacity_factor=1.0)
        torch.manual_seed(0)
        full = _init(sm.create_model(name, **kw), 31)
        torch.manual_seed(0)
        part = sm.create_model(name, world_size=world, **kw)
        for m in (full, part):
            for blk in m.blocks:
                blk.mlp.gate.switch_eps = 0.0
        sd = full.state_dict()
        sl = slice(rank * E_local, (rank + 1) * E_local)
        for k in list(sd):
            if ".experts." in k:
                sd[k] = sd[k][sl].clone()
        part.load_state_dict(sd)
        full, part = full.to(DEV), part.to(DEV)
        from slim_switch_moe_vit_amd import ep
        assert ep.static_kind(part.blocks[0].mlp, torch.float16) == "capacity"
        ignored = sm.ddp_ignore_expert_parameters(part)
        assert ignored and all(".experts." in n for n in ignored)
        ddp = DDP(part, device_ids=[0])
        images, target = _data(8)
        per = images.shape[0] // world
        # this rank's half through the expert-parallel DDP model ...
        mine = slice(rank * per, (rank + 1) * per)
        loss, grads = _grads(ddp, images[mine], target[mine])
        grads = {n.replace("module.", "", 1): g for n, g in grads.items()}
        # ... against the single-rank model holding all experts: shared parameters = the mean over the ranks' halves (what DDP's
        # all-reduce leaves), an expert slice = the SUM over the halves (rows of both ranks reach the owner through the exchange's
        # adjoint; nobody averages them -- FastMoE's convention)
        ref_sum, ref_losses = {}, []
        for r in range(world):
            l_r, g_r = _grads(full, images[r * per:(r + 1) * per], target[r * per:(r + 1) * per])
            ref_losses.append(l_r)
            for n, g in g_r.items():
                ref_sum[n] = ref_sum.get(n, 0) + g
        worst_shared = worst_expert = 0.0
        for n, g in grads.items():
            if ".experts." in n:
                ref = ref_sum[n][sl]
                worst_expert = max(worst_expert, float((g.double() - ref.double()).norm() / ref.double().norm().clamp(min=1e-12)))
            else:
                ref = ref_sum[n] / world
                if float(ref.abs().max()) > 1e-4:
                    e_ = float((g.double() - ref.double()).norm() / ref.double().norm().clamp(min=1e-12))
                    worst_shared = max(worst_shared, e_)
        shared = {n: g.numpy() for n, g in grads.items() if ".experts." not in n}
        q.put((rank, loss, ref_losses[rank], worst_shared, worst_expert, shared, len(ignored)))
    except BaseException as exc:
        q.put((rank, "error", repr(exc)))
        raise
    finally:
        dist.destroy_process_group()


def test_expert_parallel_model_under_ddp_shares_what_is_shared_and_keeps_the_expert_slices_apart():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_ddp_ep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r = q.get(timeout=200)
            assert r[1] != "error", f"rank {r[0]}: {r[2]}"
            got[r[0]] = r[1:]
    except BaseException:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(10)
        raise
    join_or_kill(procs, 120)
    for r in range(world):
        loss, ref_loss, w_shared, w_expert, _, n_ignored = got[r]
        print(f"rank {r}: loss {loss:.5f} (single-rank model on the same half: {ref_loss:.5f}); gradient rel-L2 worst shared tensor "
              f"{w_shared:.2e}, worst expert slice {w_expert:.2e}; {n_ignored} expert parameters kept out of DDP")
        assert abs(loss - ref_loss) <= 2e-3 * max(1.0, abs(ref_loss))
        # measured: 0.0 / 0.0 -- the expert-parallel rank computes every row exactly as the single-rank model does (static exchange ==
        # single-rank operator), sums the two sources' expert gradients in the same order and DDP averages the same two numbers
        assert w_shared <= 1e-4 * dtype_factor() and w_expert <= 1e-4 * dtype_factor(), (w_shared, w_expert)
    for n in got[0][4]:
        assert (got[0][4][n] == got[1][4][n]).all(), f"{n}: DDP left different shared gradients on the two ranks"
