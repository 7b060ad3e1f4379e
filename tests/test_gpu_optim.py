"""GPU tests of the optimizer side of the training step (engine.py:68-74; optim.AdamW / optim.NativeScaler on the HIP
kernels smoe_grad_sumsq / smoe_adamw_step / smoe_amp_update) against torch.optim.AdamW + torch GradScaler semantics +
torch.nn.utils.clip_grad_norm_."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

import slim_switch_moe_vit_amd as sm  # noqa: E402

DEV = "cuda:0"


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) for s in shapes]


SHAPES = [(8, 96, 64), (8, 64, 96), (8, 96), (3, 7), (1,), (40001,)]   # expert-tensor shaped, odd sizes, a scalar


@pytest.mark.parametrize("gdt", [torch.float32, torch.float16])
def test_adamw_kernel_matches_torch_adamw(gdt):
    init = _params(0, SHAPES)
    ours = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    ref = [torch.nn.Parameter(t.clone().double()) for t in init]
    a = sm.AdamW(ours, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    b = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    for it in range(12):
        grads = _params(100 + it, SHAPES)
        for p, q, g in zip(ours, ref, grads):
            gg = g.to(gdt)
            if gdt != torch.float32:
                p.grad_dtype = gdt          # torch >= 2.10: a gradient dtype other than the parameter's must be declared
            p.grad = gg.to(DEV)
            q.grad = gg.double()
        a.step()
        b.step()
    for p, q in zip(ours, ref):
        assert (p.detach().cpu().double() - q.detach()).abs().max().item() <= 2e-6 * max(1.0, float(q.abs().max()))
    assert float(a._step_counter(torch.device(DEV))) == 12.0


def test_adamw_multi_tensor_launch_equals_per_tensor_launches_bitwise():
    """One launch over every tensor of the step (two parameter groups with their own lr / weight decay) == one launch per
    tensor, bit for bit, and == torch.optim.AdamW with the same groups."""
    init = _params(5, SHAPES)
    multi = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    single = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    ref = [torch.nn.Parameter(t.clone().double()) for t in init]
    groups = lambda ps: [dict(params=ps[:3], lr=2e-3, weight_decay=0.05), dict(params=ps[3:], lr=5e-4, weight_decay=0.0)]
    a = sm.AdamW(groups(multi), betas=(0.9, 0.999))
    singles = [sm.AdamW([p], lr=(2e-3 if i < 3 else 5e-4), weight_decay=(0.05 if i < 3 else 0.0), betas=(0.9, 0.999))
               for i, p in enumerate(single)]
    b = torch.optim.AdamW(groups(ref), betas=(0.9, 0.999))
    for it in range(5):
        grads = _params(200 + it, SHAPES)
        for p, q, r, g in zip(multi, single, ref, grads):
            p.grad, q.grad, r.grad = g.to(DEV), g.to(DEV), g.double()
        a.step()
        for o in singles:
            o.step()
        b.step()
    for p, q, r in zip(multi, single, ref):
        assert torch.equal(p.detach(), q.detach())
        assert (p.detach().cpu().double() - r.detach()).abs().max().item() <= 2e-6 * max(1.0, float(r.abs().max()))


def test_native_scaler_step_matches_the_stock_sequence_and_skips_on_inf():
    """scale -> backward -> (unscale, clip, step, update) on device scalars == the stock sequence on a float64 twin:
    clipping engages (norm > max_norm), a step with an inf gradient is skipped (parameters, Adam moments AND the step
    count untouched) and backs the scale off, clean steps grow it after `growth_interval`."""
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.GELU(), torch.nn.Linear(128, 10)).to(DEV)
    twin = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.GELU(), torch.nn.Linear(128, 10)).double()
    twin.load_state_dict({k: v.double().cpu() for k, v in lin.state_dict().items()})
    opt = sm.AdamW(lin.parameters(), lr=1e-2, weight_decay=0.05)
    ropt = torch.optim.AdamW(twin.parameters(), lr=1e-2, weight_decay=0.05)
    sc = sm.NativeScaler(init_scale=1024.0, growth_interval=3)
    max_norm = 0.5
    g = torch.Generator().manual_seed(1)
    scales = []
    for it in range(7):
        x = torch.randn(32, 64, generator=g)
        y = torch.randint(0, 10, (32,), generator=g)
        poison = (it == 4)
        loss = torch.nn.functional.cross_entropy(lin(x.to(DEV)), y.to(DEV))
        if poison:
            loss = loss + float("inf") * lin[0].weight.sum() * 0   # nan gradients on one tensor
        opt.zero_grad()
        before = [p.detach().clone() for p in lin.parameters()]
        sc(loss, opt, clip_grad=max_norm, parameters=lin.parameters())
        scales.append(sc.get_scale())
        if poison:
            for p, q in zip(lin.parameters(), before):
                assert torch.equal(p.detach(), q), "a non-finite step must leave the parameters alone"
            continue
        rl = torch.nn.functional.cross_entropy(twin(x.double()), y)
        ropt.zero_grad()
        rl.backward()
        norm = torch.nn.utils.clip_grad_norm_(twin.parameters(), max_norm)
        assert float(norm) > max_norm, "the test must exercise clipping"
        assert abs(float(sc.last_grad_norm) - float(norm)) <= 1e-4 * float(norm)
        ropt.step()
    for p, q in zip(lin.parameters(), twin.parameters()):
        assert (p.detach().cpu().double() - q.detach()).abs().max().item() <= 5e-5
    assert float(opt._step_counter(torch.device(DEV))) == 6.0, "the skipped step does not count"
    # 1024 -> (3 clean) 2048 -> (1 clean) -> inf: 1024 -> (2 clean)
    assert scales == [1024.0, 1024.0, 2048.0, 2048.0, 1024.0, 1024.0, 1024.0], scales
    sd = sc.state_dict()
    assert set(sd) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"} and sd["_growth_tracker"] == 2
    sc2 = sm.NativeScaler()
    sc2.load_state_dict(sd)
    assert sc2.get_scale() == 1024.0


def test_cfg5_training_step_through_the_harness():
    """BASELINE cfg 5 shape of a step, end to end on the HIP path: ViT with SwitchGate MoE blocks (capacity 1.0, token
    dropping, aux loss), fp16 autocast forward, backward, NativeScaler + AdamW; the loss goes down over a few steps on a
    fixed batch and every expert tensor receives updates."""
    torch.manual_seed(0)
    model = sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=10, depth=2, gate="switch",
                            capacity_factor=1.0).to(DEV)
    opt = sm.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05)
    sc = sm.NativeScaler()
    g = torch.Generator().manual_seed(3)
    batch = [(torch.randn(8, 3, 224, 224, generator=g), torch.randint(0, 10, (8,), generator=g))]
    w0 = model.blocks[0].mlp.experts.htoh4.weight.detach().clone()
    losses = []
    for _ in range(6):
        st = sm.train_one_epoch(model, torch.nn.CrossEntropyLoss(), batch, opt, DEV, 0, sc, max_norm=1.0, aux_loss_weight=0.01)
        losses.append(st["loss"])
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    assert float((model.blocks[0].mlp.experts.htoh4.weight.detach() - w0).abs().max()) > 0


def test_fused_adamw_refreshes_the_16_bit_weight_shadows_autocast_training_matches_a_torch_adamw_twin():
    """The fused step writes the f32 masters through raw pointers.  Every 16-bit shadow keyed on the parameter's version
    (expert GEMM operands forward and backward, qkv / proj / head / patch-embed) must be rebuilt after it, or every step
    after the first trains against the INITIAL fp16 weights (the reference's AMP configuration, engine.py:52-74).  Two
    copies of one model take the same 5 autocast steps on the same batches: one through sm.AdamW (fused, device scalars),
    one through torch.optim.AdamW behind the same scaler (torch in-place ops, which bump the versions by themselves).
    Their parameters must stay together (with stale shadows they part by ~lr per element from the second step on), the
    shadows must equal the casts of the masters, and a no_grad autocast eval must see the updated weights."""
    import copy
    torch.manual_seed(0)
    a = sm.create_model("moe_tiny_patch16_224_expert8", num_classes=10, depth=2).to(DEV)   # E = 8, top-2
    b = copy.deepcopy(a)
    lr = 2e-3
    init = {n: p.detach().clone() for n, p in a.named_parameters()}
    oa = sm.AdamW(a.parameters(), lr=lr, weight_decay=0.05)
    ob = torch.optim.AdamW(b.parameters(), lr=lr, weight_decay=0.05)
    sa, sb = sm.NativeScaler(init_scale=1024.0), sm.NativeScaler(init_scale=1024.0)
    g = torch.Generator().manual_seed(3)
    images = torch.randn(8, 3, 224, 224, generator=g).to(DEV)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a.eval()
        logits0 = a(images).float().clone()
        a.train()
    crit = torch.nn.CrossEntropyLoss()
    for it in range(5):
        batch = [(torch.randn(8, 3, 224, 224, generator=g), torch.randint(0, 10, (8,), generator=g))]
        sm.train_one_epoch(a, crit, batch, oa, DEV, 0, sa, max_norm=1.0)
        sm.train_one_epoch(b, crit, batch, ob, DEV, 0, sb, max_norm=1.0)
    # Adam's update is m / (sqrt(v) + eps): an element whose gradient is rounding noise moves by +-lr whatever the noise's
    # size, so single elements may part by O(lr) between two CORRECT runs; the update as a whole may not.  With stale
    # shadows every step after the first differentiates at the initial weights and the two updates part by tens of per cent.
    num = den = 0.0
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        da, db = (p.detach() - init[n]).double(), (q.detach() - init[n]).double()
        num += float((da - db).pow(2).sum())
        den += float(db.pow(2).sum())
    rel = (num / den) ** 0.5
    print(f"fused vs torch AdamW after 5 autocast steps: relative L2 difference of the parameter updates {rel:.3e}")
    assert rel <= 0.05, rel
    for blk in a.blocks:
        for lin in (blk.mlp.experts.htoh4, blk.mlp.experts.h4toh):
            assert torch.equal(lin.weight_as(torch.float16), lin.weight.detach().half())
            assert torch.equal(lin.weight_t_as(torch.float16), lin.weight.detach().transpose(1, 2).half().contiguous())
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a.eval(); b.eval()
        la, lb = a(images).float(), b(images).float()
    print(f"eval logits: fused vs twin {float((la - lb).abs().max()):.3e}, vs the initial model {float((la - logits0).abs().max()):.3e}")
    assert float((la - lb).abs().max()) <= 5e-2, "eval after training must read the updated weights"
    assert float((la - logits0).abs().max()) > 5 * float((la - lb).abs().max()), "the weights did move"


def test_adamw_state_dict_round_trip_keeps_the_bias_correction_step():
    """optimizer.load_state_dict(checkpoint['optimizer']) (main.py:717): exp_avg / exp_avg_sq AND the step count of the
    bias correction survive a save / load -- into this class from its own checkpoint, from a torch.optim.AdamW checkpoint
    (tensor 'step'), and back into torch.optim.AdamW -- so that the next step equals the uninterrupted run's."""
    init = _params(7, SHAPES)
    kw = dict(lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)

    def run(opt_cls, params, its, start=0):
        opt = opt_cls(params, **kw) if isinstance(opt_cls, type) else opt_cls
        for it in range(start, start + its):
            for p, gr in zip(params, _params(300 + it, SHAPES)):
                p.grad = gr.to(p.device, p.dtype)
            opt.step()
        return opt

    ref = [torch.nn.Parameter(t.clone().double()) for t in init]
    run(torch.optim.AdamW, ref, 9)                                   # the uninterrupted run, float64
    ours = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    o1 = run(sm.AdamW, ours, 5)
    sd = o1.state_dict()
    assert all(int(st["step"]) == 5 for st in sd["state"].values())
    # (1) own checkpoint -> a fresh instance (through torch.save / torch.load, as the reference does)
    import io
    buf = io.BytesIO(); torch.save(sd, buf); buf.seek(0)
    sd_l = torch.load(buf, weights_only=False)
    resumed = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    o2 = sm.AdamW(resumed, **kw)
    o2.load_state_dict(sd_l)
    run(o2, resumed, 4, start=5)
    assert float(o2._step_counter(torch.device(DEV))) == 9.0
    for p, q in zip(resumed, ref):
        assert (p.detach().cpu().double() - q.detach()).abs().max().item() <= 3e-6 * max(1.0, float(q.abs().max()))
    # (2) a torch.optim.AdamW checkpoint (tensor 'step') -> this class
    tp = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    to = run(torch.optim.AdamW, tp, 5)
    resumed2 = [torch.nn.Parameter(p.detach().clone()) for p in tp]
    o3 = sm.AdamW(resumed2, **kw)
    o3.load_state_dict(to.state_dict())
    run(o3, resumed2, 4, start=5)
    for p, q in zip(resumed2, ref):
        assert (p.detach().cpu().double() - q.detach()).abs().max().item() <= 3e-6 * max(1.0, float(q.abs().max()))
    # (3) this class's checkpoint -> torch.optim.AdamW
    resumed3 = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    o4 = torch.optim.AdamW(resumed3, **kw)
    o4.load_state_dict(sd)
    run(o4, resumed3, 4, start=5)
    for p, q in zip(resumed3, ref):
        assert (p.detach().cpu().double() - q.detach()).abs().max().item() <= 3e-6 * max(1.0, float(q.abs().max()))


class _DataSGD(torch.optim.Optimizer):
    """What timm's native optimizers do (main.py:90-96 --opt ...): the update goes through ``p.data``, which does NOT move
    ``p._version``."""

    def __init__(self, params, lr):
        super().__init__(params, dict(lr=lr))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    p.data.add_(p.grad.data, alpha=-group["lr"])


def test_foreign_optimizer_step_invalidates_the_16_bit_weight_images():
    """the modules' 16-bit weight images are keyed on (param._version, data_ptr, dtype); an optimizer that
    updates through ``p.data`` leaves ``_version`` alone, and the next forward kept reading the OLD images -- silently.  The global
    optimizer step post-hook (optim._foreign_step_hook) bumps the versions after any foreign optimizer's step: one such SGD step
    changes the next forward's output exactly as a fresh model loaded with the updated weights computes it."""
    torch.manual_seed(0)
    model = sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=10, depth=2).to(DEV)
    with torch.no_grad():
        for blk in model.blocks:
            blk.mlp.experts.htoh4.weight.normal_(0, 0.02)
            blk.mlp.experts.h4toh.weight.normal_(0, 0.02)
        model.head.weight.normal_(0, 0.02)
    images = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(DEV)
    target = torch.tensor([1, 2, 3, 4], device=DEV)

    def infer(m):
        m.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return m(images).float()

    before = infer(model)                                    # builds every 16-bit image
    opt = _DataSGD(model.parameters(), lr=0.5)
    model.train()
    with torch.autocast("cuda", dtype=torch.float16):
        loss = torch.nn.functional.cross_entropy(model(images).float(), target)
    loss.backward()
    versions = [p._version for p in model.parameters()]
    opt.step()
    moved = [p._version != v for p, v in zip(model.parameters(), versions) if p.grad is not None]
    assert all(moved), "the hook must have bumped the version of every parameter the optimizer stepped"
    after = infer(model)

    def fresh_copy():                                         # a new model loaded with the CURRENT weights: no cache of any kind
        torch.manual_seed(0)
        m = sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=10, depth=2)
        m.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        return m.to(DEV)
    want = infer(fresh_copy())
    assert float((after - before).abs().max()) > 1e-3, "the step must have changed the output"
    assert torch.equal(after, want), float((after - want).abs().max())
    # the documented escape hatch for updates no optimizer hook can see
    with torch.no_grad():
        for p in model.parameters():
            p.data.mul_(1.01)
    sm.invalidate_weight_images()
    assert torch.equal(infer(model), infer(fresh_copy()))


def test_training_harness_on_one_hip_graph_reproduces_the_eager_harness():
    """engine.train_one_epoch(hip_graph=True): after three eager steps per batch shape the whole step (forward, backward, clipping,
    AdamW, loss-scale update) is captured once and replayed.  Same data, same seeds, stochastic depth off (its random draws come
    from another point of the generator's stream under a graph): after 9 steps the PARAMETERS, the optimizer's moments, the loss scale
    and the mean loss are bit for bit the eager harness', the 16-bit weight images are current afterwards (an eval forward equals a
    freshly loaded model's), and the token-skip counters agree."""
    name, kw = "resmoe_tiny_patch16_224_expert8", dict(num_classes=10, depth=2, starting_threshold=0.55, target_threshold=0.5)
    g = torch.Generator().manual_seed(70)
    batches = [(torch.randn(8, 3, 224, 224, generator=g), torch.randint(0, 10, (8,), generator=g)) for _ in range(9)]

    def run(graph):
        torch.manual_seed(0)
        model = sm.create_model(name, **kw)
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if "_gate.head.1.weight" in n_:
                    p.normal_(0, 0.3, generator=torch.Generator().manual_seed(5))
        model = model.to(DEV)
        opt = sm.AdamW(model.parameters(), lr=1e-3, weight_decay=0.05)
        scaler = sm.NativeScaler()
        stats = sm.train_one_epoch(model, torch.nn.CrossEntropyLoss(), batches, opt, DEV, 0, scaler, 1.0, hip_graph=graph)
        gates = [(m._total_tokens, m._skipped_tokens) for m in model.modules() if isinstance(m, sm.Gate)]
        model.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            logits = model(batches[0][0].to(DEV)).float()
        moments = [opt.state[p]["exp_avg"].clone() for p in model.parameters() if p in opt.state]
        return stats, [p.detach().clone() for p in model.parameters()], moments, scaler.state_dict(), gates, logits, model

    s_e, p_e, m_e, sc_e, g_e, l_e, _ = run(False)
    s_g, p_g, m_g, sc_g, g_g, l_g, model_g = run(True)
    assert s_g["hip_graph_steps"] == 6 and s_e["hip_graph_steps"] == 0
    assert s_g["loss"] == s_e["loss"], (s_g, s_e)
    assert all(torch.equal(a, b) for a, b in zip(p_e, p_g)), "parameters after 9 steps"
    assert all(torch.equal(a, b) for a, b in zip(m_e, m_g)), "AdamW moments after 9 steps"
    assert sc_e == sc_g and g_e == g_g, (sc_e, sc_g, g_e, g_g)
    assert torch.equal(l_e, l_g)
    fresh = sm.create_model(name, **kw)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model_g.state_dict().items()})
    fresh = fresh.to(DEV).eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert torch.equal(fresh(batches[0][0].to(DEV)).float(), l_g), "the weight images are current after the replays"
