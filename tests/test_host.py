"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/slimmoe.h
declares (no compute calls without a GPU), and the nn.Module surface mirrors models/resMoE.py."""
import ctypes
import inspect
import os
import re
import types

import pytest
import torch
import torch.nn as nn

import slim_switch_moe_vit_amd as sm
from slim_switch_moe_vit_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "slimmoe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smoe_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = _declared_symbols()
    assert len(syms) >= 21
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        if s in _lib.OPTIONAL:      # (declared, but only an optional build holds them: `make FFN=-DSMOE_FFN_FUSED`)
            continue
        assert hasattr(lib, s), f"libslimmoe_hip.so does not export {s}"
    assert set(syms) == set(_lib.SIGNATURES), "ctypes signature table out of sync with include/slimmoe.h"


def test_library_loads_and_reports_abi():
    lib = _lib.load()
    assert lib.smoe_abi_version() == _lib.ABI_VERSION
    # the binary that travels with the tree was built from the tree's sources (csrc/Makefile hashes them into the library)
    assert _lib.source_build_id() is not None and len(_lib.binary_build_id()) == 16
    assert _lib.binary_build_id() == _lib.source_build_id(), (
        "libslimmoe_hip.so was not built from the sources in slim-switch-moe-vit_amd/csrc: run __graft_entry__.build()")
    assert lib.smoe_dispatch_plan_workspace_bytes(50432, 8) >= 2 * 50 * 8 * 4


def test_argument_validation_errors_come_back_as_messages():
    lib = _lib.load()
    # null pointers / bad sizes are rejected before any launch (safe without a GPU)
    rc = lib.smoe_router_topk(None, 0, None, None, None, 4, 12, 2, 1, 0, None, None, None, None, None, 0, None)
    assert rc != 0 and b"null" in lib.smoe_last_error()
    rc = lib.smoe_grouped_gemm(None, None, None, None, None, 0, 0, 8, 64, 64, 1, 0, None, None, None, None, 1, None, 0, 1, 0, None, None)
    assert rc != 0 and lib.smoe_last_error()
    with pytest.raises(_lib.SlimMoEError):
        _lib.check(rc, "smoe_grouped_gemm")
    # out_rows (ABI 20): without a row map the output cannot have fewer rows than are dispatched -- refused before any launch
    fake = 4096                                            # never dereferenced: the argument checks come first
    rc = lib.smoe_grouped_gemm(fake, fake, None, fake, None, 1, 1, 8, 64, 64, 1, 0, None, None, None, None, 1, fake, 4, 1, 9, None, None)
    assert rc != 0 and b"out_rows" in lib.smoe_last_error()


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the shipped module raises instead of silently computing elsewhere."""
    m = sm.CustomizedMoEMLP(32, 64, 4, 1, 0.0).eval()
    with pytest.raises(RuntimeError, match="GPU"):
        with torch.no_grad():
            m(torch.randn(2, 3, 32))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.dispatch_plan(torch.zeros(4, dtype=torch.int64), 2)


def test_customized_moe_mlp_signature_and_state_dict_keys():
    sig = inspect.signature(sm.CustomizedMoEMLP.__init__)
    assert list(sig.parameters)[:7] == ["self", "in_features", "hidden_features", "moe_num_experts", "moe_top_k",
                                        "drop", "act_layer"]  # models/resMoE.py:15-24
    m = sm.CustomizedMoEMLP(192, 768, 8, 2, 0.0)
    sd = m.state_dict()
    assert sd["gate.gate.weight"].shape == (8, 192) and sd["gate.gate.bias"].shape == (8,)
    assert sd["experts.htoh4.weight"].shape == (8, 768, 192) and sd["experts.htoh4.bias"].shape == (8, 768)
    assert sd["experts.h4toh.weight"].shape == (8, 192, 768) and sd["experts.h4toh.bias"].shape == (8, 192)
    assert m.gate.gate.in_features == 192 and m.gate.gate.out_features == 8  # models/resmoe_flop_hook.py:7
    assert isinstance(m.experts.activation, nn.Sequential) and isinstance(m.experts.activation[0], nn.GELU)


def test_gate_module_mirrors_reference():
    g = sm.Gate(16, 1.0, starting_threshold=1.0, target_threshold=0.9)
    assert set(g.state_dict()) == {"head.1.weight", "head.1.bias", "_threshold", "threshold"}  # resMoE.py:43-45
    x = torch.randn(2, 5, 16)
    g.disable = True
    m = g(x)
    assert m.shape == (2, 5, 2) and torch.all(m[..., 1] == 1) and torch.all(m[..., 0] == 0)
    g.disable = False
    g.eval()
    m = g(x)
    p = torch.sigmoid(g.head(x))
    assert torch.equal(m[..., 0:1] > 0.5, p > 0.9) and torch.allclose(m.sum(-1), torch.ones(2, 5))
    assert g._total_tokens == 10 and g._skipped_tokens == float((p > 0.9).sum())
    g.step(torch.tensor(0.25))
    assert float(g._threshold) == pytest.approx(0.9)  # clamps at the target (resMoE.py:53-57)


def test_factories_registered_and_patch_every_block():
    for name in ("resmoe_tiny_patch16_224_expert8", "moe_tiny_patch16_224_expert8"):  # resMoE.py:151,190
        assert name in sm.list_models()
    m = sm.create_model("resmoe_tiny_patch16_224_expert8", pretrained=False, num_classes=10, drop_block_rate=None,
                        starting_threshold=1.0, target_threshold=0.9)
    blocks = [b for b in m.modules() if isinstance(b, sm.Block)]
    assert len(blocks) == 12
    for b in blocks:
        assert isinstance(b.mlp, sm.CustomizedMoEMLP) and b.mlp.top_k == 2 and b.mlp.num_expert == 8
        assert isinstance(b.dense_gate, sm.Gate) and isinstance(b.moe_gate, sm.Gate)
        assert b.forward.__func__ is sm.forward_residule_moe
    names = [n for n, _ in m.named_parameters()]
    assert any("moe_gate" in n for n in names) and any("dense_gate" in n for n in names)  # main.py:623
    m2 = sm.create_model("moe_base_patch16_224_expert8_top1", num_classes=1000)
    assert m2.blocks[0].mlp.d_model == 768 and m2.blocks[0].mlp.d_hidden == 3072 and m2.blocks[0].mlp.top_k == 1


def test_switch_gate_capacity_definitions():
    g = sm.SwitchGate(768, 8, 1, capacity=1.0)
    assert g.capacity(50432) == 6304  # BASELINE.md cfg 5: ceil(cf * T_local * k / E)
    g2 = sm.SwitchGate(768, 8, 1, capacity=(1.2, 2.4), capacity_mode="fmoe")
    g2.eval()
    assert g2.capacity(1000) == 2400


def test_dense_shell_runs_on_cpu():
    m = sm.create_model("deit_tiny_patch16_224", num_classes=10).eval()
    with torch.no_grad():
        y = m(torch.randn(1, 3, 224, 224))
    assert y.shape == (1, 10)


def test_checkpoint_round_trip_in_both_fastmoe_layouts(tmp_path):
    """main.py:893-907 saves ``model.state_dict()``, main.py:703-724 loads it back: a reference-layout checkpoint
    (FastMoE < 1.1 keys ``experts.htoh4.weight [E,h,d]``) and FastMoE >= 1.1's per-expert layout
    (``experts.{e}.htoh4.weight [1,h,d]``) must both load into the modules here and give back the same tensors."""
    torch.manual_seed(0)
    a = sm.create_model("resmoe_tiny_patch16_224_expert8", num_classes=7, depth=2)
    sd = a.state_dict()
    assert "blocks.0.mlp.experts.htoh4.weight" in sd and "blocks.1.moe_gate.head.1.weight" in sd
    path = tmp_path / "ckpt.pth"
    torch.save({"model": sd, "epoch": 3}, path)
    ck = torch.load(path, map_location="cpu")
    torch.manual_seed(1)
    b = sm.create_model("resmoe_tiny_patch16_224_expert8", num_classes=7, depth=2)
    b.load_state_dict(ck["model"])                                   # strict
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    v11 = sm.fastmoe_v11_state_dict(sd)
    assert "blocks.0.mlp.experts.3.h4toh.bias" in v11 and "blocks.0.mlp.experts.htoh4.weight" not in v11
    assert tuple(v11["blocks.0.mlp.experts.3.htoh4.weight"].shape) == (1, 768, 192)
    torch.manual_seed(2)
    c = sm.create_model("resmoe_tiny_patch16_224_expert8", num_classes=7, depth=2)
    c.load_state_dict(v11)                                           # strict: nothing missing, nothing unexpected
    for (ka, va), (kc, vc) in zip(a.state_dict().items(), c.state_dict().items()):
        assert ka == kc and torch.equal(va, vc), ka
    # a per-expert checkpoint with the wrong number of experts is an error, not a silent truncation
    bad = {k: v for k, v in v11.items() if ".experts.7." not in k}
    with pytest.raises(RuntimeError):
        c.load_state_dict(bad)
    # an in-place load refreshes the 16-bit weight shadows (they are keyed on tensor versions, which .data updates skip)
    lin = c.blocks[0].mlp.experts.htoh4
    lin._shadow._c[("probe", None)] = (0, torch.zeros(1), None, None)
    c.load_state_dict(v11)
    assert ("probe", None) not in lin._shadow._c


def test_expert_parameters_are_tagged_and_kept_out_of_ddp():
    """under expert parallelism plain DDP would broadcast / average DIFFERENT experts.  Parameters carry
    FastMoE's dp_comm tags and ddp_ignore_expert_parameters registers the rank-private ones with DDP's ignore list."""
    m = sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=5, depth=2, world_size=2)
    mlp = m.blocks[0].mlp
    assert all(p.dp_comm == "none" for p in mlp.experts.parameters())
    assert all(p.dp_comm == "dp" for p in mlp.gate.parameters())
    names = sm.ddp_ignore_expert_parameters(m)
    assert "blocks.0.mlp.experts.htoh4.weight" in names and "blocks.1.mlp.experts.h4toh.bias" in names
    assert not any("gate" in n for n in names)
    assert set(names) <= set(m._ddp_params_and_buffers_to_ignore)
    single = sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=5, depth=1)
    assert sm.ddp_ignore_expert_parameters(single) == []            # single rank: ordinary data parallelism is right


def test_adamw_and_native_scaler_cpu_composition_matches_torch():
    """optim.AdamW / optim.NativeScaler on CPU tensors take the torch composition (the HIP kernels need the GPU): same
    numbers as torch.optim.AdamW stepped through unscale -> clip_grad_norm_ -> step -> update, a non-finite step is
    skipped and backs the scale off, the state dict has GradScaler's keys."""
    torch.manual_seed(0)
    a = torch.nn.Linear(7, 5)
    b = torch.nn.Linear(7, 5)
    b.load_state_dict(a.state_dict())
    oa = sm.AdamW(a.parameters(), lr=1e-2, weight_decay=0.05)
    ob = torch.optim.AdamW(b.parameters(), lr=1e-2, weight_decay=0.05)
    sc = sm.NativeScaler(init_scale=256.0, growth_interval=2)
    g = torch.Generator().manual_seed(1)
    scales = []
    for it in range(5):
        x = torch.randn(16, 7, generator=g) * 3
        la = a(x).square().mean()
        if it == 2:
            la = la * float("inf")
        oa.zero_grad()
        sc(la, oa, clip_grad=0.3, parameters=a.parameters())
        scales.append(sc.get_scale())
        if it == 2:
            continue
        ob.zero_grad()
        b(x).square().mean().backward()
        torch.nn.utils.clip_grad_norm_(b.parameters(), 0.3)
        ob.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.allclose(p, q, rtol=0, atol=1e-6)
    assert scales == [256.0, 512.0, 256.0, 256.0, 512.0], scales
    assert set(sc.state_dict()) == {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}


def test_gate_threshold_schedule_and_counters_on_cpu():
    """Gate.step anneals `_threshold` towards `threshold` (main.py:812-815 calls it every iteration), never below it;
    the CPU composition counts skipped tokens; masks are exactly 0 / 1 and sum to 1."""
    gate = sm.Gate(16, 1.0, target_threshold=0.7, starting_threshold=0.9)
    for _ in range(5):
        gate.step(torch.tensor(0.06))
    assert abs(float(gate._threshold) - 0.7) < 1e-6
    gate.eval()
    x = torch.randn(2, 9, 16, generator=torch.Generator().manual_seed(0)) * 4
    m = gate(x)
    assert m.shape == (2, 9, 2) and torch.all((m == 0) | (m == 1)) and torch.all(m.sum(-1) == 1)
    assert gate._total_tokens == 18 and gate._skipped_tokens == float(m[..., 0].sum())
    gate._skipped_tokens = 0
    assert gate._skipped_tokens == 0


def test_gate_module_cpu_composition_against_the_reference_gate_fixture(golden_dir):
    """The ``Gate`` module's differentiable composition (what CPU tensors take) against outputs AND gradients of the reference's own
    ``Gate`` (models/resMoE.py:59-85; tests/golden/make_golden_resmoe.py) in eval, train-hard (straight-through: d mask / d p = +1
    / -1), train-soft ((1 - p, p)) and disabled mode, and the ``step`` schedule (53-57).  This path compares the f32 sigmoid as the
    reference does: every decision is the reference's, incl. the rows engineered onto the threshold."""
    import os
    import numpy as np
    g = {k: v for k, v in np.load(os.path.join(golden_dir, "ref_gate_tiny.npz")).items()}
    x, dret = torch.from_numpy(g["x"]), torch.from_numpy(g["dret"])

    def make(is_hard=True):
        gate = sm.Gate(192, 1.0, target_threshold=float(g["thr_eval"]), starting_threshold=float(g["thr_train"]), is_hard=is_hard)
        with torch.no_grad():
            gate.head[1].weight.copy_(torch.from_numpy(g["w"])); gate.head[1].bias.copy_(torch.from_numpy(g["b"]))
        return gate
    # Rows engineered onto the threshold decide by the last bit of the f32 sigmoid, and that bit belongs to the CPU's vector exp
    # (the fixture was written on the build container's CPU; another host may round a row the other way): a decision may differ
    # from the fixture only where the float64 sigmoid is within 2 ulp of the threshold.
    p64 = torch.sigmoid(x.double() @ torch.from_numpy(g["w"]).double().t() + torch.from_numpy(g["b"]).double()).squeeze(-1).numpy()

    def same_decisions(mask, ref_mask, thr):
        on_the_line = np.abs(p64 - float(thr)) <= 2.4e-7
        differ = (np.rint(mask) != np.rint(ref_mask)).any(-1)
        assert not (differ & ~on_the_line).any(), "a decision differs from the reference's away from the threshold"
        return int(differ.sum())
    gate = make().eval()
    with torch.no_grad():
        m = gate(x)
    flips = same_decisions(m.numpy(), g["eval_mask"], g["thr_eval"])
    keep = (np.rint(m.numpy()) == np.rint(g["eval_mask"])).all(-1)
    assert np.abs(m.numpy() - g["eval_mask"])[keep].max() <= 1.2e-7
    assert gate._total_tokens == int(g["eval_total"])
    assert abs(gate._skipped_tokens - float(np.rint(g["eval_mask"])[..., 0].sum())) <= flips
    for mode, hard in (("train_hard", True), ("train_soft", False)):
        gate = make(hard).train()
        xg = x.clone().requires_grad_(True)
        m = gate(xg)
        m.backward(dret)
        flips = 0
        if hard:
            flips = same_decisions(m.detach().numpy(), g[f"{mode}_mask"], g["thr_train"])
            keep = (np.rint(m.detach().numpy()) == np.rint(g[f"{mode}_mask"])).all(-1)
        else:
            keep = np.ones(p64.shape, dtype=bool)
        assert np.abs(m.detach().numpy() - g[f"{mode}_mask"])[keep].max() <= 1.2e-7, mode
        for got, key in ((xg.grad, "dx"), (gate.head[1].weight.grad, "dw"), (gate.head[1].bias.grad, "db")):
            ref = torch.from_numpy(g[f"{mode}_{key}"])
            assert torch.allclose(got, ref, rtol=1e-5, atol=1e-6 * max(1.0, float(ref.abs().max()))), (mode, key)
        assert gate._total_tokens == int(g[f"{mode}_total"])
        assert abs(gate._skipped_tokens - float(g[f"{mode}_skipped"])) <= 1e-3 * max(1.0, float(g[f"{mode}_skipped"])) + flips
    gate = make()
    gate.disable = True
    assert np.array_equal(gate(x).detach().numpy(), g["disabled_mask"])
    gate = make()
    seq = []
    for _ in range(4):
        gate.step(torch.tensor(0.04))
        seq.append(float(gate._threshold))
    assert np.allclose(seq, g["step_sequence"], rtol=0, atol=1e-7), (seq, g["step_sequence"])


def test_multi_tensor_block_table_and_wgrad_round_model():
    """Host-side tables of the round's launch-count work: the workgroup -> (tensor, 16K block) map of the multi-tensor optimizer
    kernels, and the cost model that picks the weight-gradient tile height / orientation."""
    import numpy as np
    from slim_switch_moe_vit_amd import optim as smo
    from slim_switch_moe_vit_amd import ops

    blk, nb = smo._block_table([5, 16384, 16385, 40000], "cpu")
    assert nb == [1, 1, 2, 3]
    assert blk.shape == (2, 7) and blk.dtype == torch.int32
    assert blk[0].tolist() == [0, 1, 2, 2, 3, 3, 3]      # tensor of each workgroup
    assert blk[1].tolist() == [0, 0, 0, 1, 0, 1, 2]      # 16K-element block inside it
    blk2, _ = smo._block_table([5, 16384, 16385, 40000], "cpu")
    assert blk2 is blk                                    # cached by the size tuple
    empty, nb0 = smo._block_table([], "cpu")
    assert nb0 == [] and empty.shape == (2, 0)
    # ViT-B expert weights on 256 CUs: dW1 [3072, 768] x 8 experts = 240 tiles of 320 rows (one round, weighted 1.25) against
    # 288 tiles of 256 rows (two rounds); dW2 [768, 3072] gains nothing from the taller tile as it stands, so ops swaps it
    assert ops._wgrad_rounds(8, 3072, 768, 256) == 1.25
    assert ops._wgrad_rounds(8, 768, 3072, 256) == 2
    assert ops._wgrad_rounds(4, 768, 3072, 256) == 1      # 144 tiles: one round either way
    assert np.isclose(ops._wgrad_rounds(1, 256, 256, 256), 1.0)


def test_train_one_epoch_takes_the_reference_positional_arguments():
    """main.py:825-838 calls ``train_one_epoch(model, criterion, loader, optimizer, device, epoch, loss_scaler, clip_grad,
    model_ema, mixup_fn, set_training_mode=..., args=...)``: positions 9 and 10 are the EMA and the mixup function
    (engine.py:22-35); the build's own extensions are keyword-only; a three-tensor criterion gets the inputs
    (engine.py:54); a non-finite loss ends the run as engine.py:56-60 does."""
    sig = inspect.signature(sm.train_one_epoch)
    names = list(sig.parameters)
    assert names[:12] == ["model", "criterion", "data_loader", "optimizer", "device", "epoch", "loss_scaler", "max_norm",
                          "model_ema", "mixup_fn", "set_training_mode", "args"]
    for extra in ("aux_loss_weight", "gate_delta", "autocast"):
        assert sig.parameters[extra].kind is inspect.Parameter.KEYWORD_ONLY
    torch.manual_seed(0)
    model = nn.Sequential(nn.Flatten(), nn.Linear(12, 5))
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    calls = {"ema": 0, "mix": 0, "crit3": 0}

    class Ema:
        def update(self, m):
            assert m is model
            calls["ema"] += 1

    def mix(x, y):
        calls["mix"] += 1
        return x, y

    class Crit3(nn.Module):
        def forward(self, inputs, outputs, labels):
            calls["crit3"] += 1
            assert inputs.shape[0] == outputs.shape[0]
            return nn.functional.cross_entropy(outputs, labels)

    data = [(torch.randn(4, 3, 2, 2), torch.randint(0, 5, (4,))) for _ in range(3)]
    st = sm.train_one_epoch(model, Crit3(), data, opt, "cpu", 0, sm.NativeScaler(enabled=False), None, Ema(), mix, True, None)
    assert calls == {"ema": 3, "mix": 3, "crit3": 3} and st["steps"] == 3 and st["lr"] == 0.1
    st = sm.train_one_epoch(model, nn.CrossEntropyLoss(), data, opt, "cpu", 1, sm.NativeScaler(enabled=False))
    assert st["steps"] == 3 and st["loss"] == st["loss"]
    bad = [(torch.full((4, 3, 2, 2), float("nan")), torch.randint(0, 5, (4,)))]
    before = [p.detach().clone() for p in model.parameters()]
    ema_calls = calls["ema"]
    with pytest.raises(SystemExit):
        sm.train_one_epoch(model, nn.CrossEntropyLoss(), bad, opt, "cpu", 2, sm.NativeScaler(enabled=False), None, Ema())
    # a scaler that does not skip non-finite steps by itself (disabled / foreign) or an EMA: the abort comes BEFORE the optimizer
    # step and the EMA update, as at engine.py:56-60 -- nothing absorbed the NaN
    assert all(torch.equal(p, b) for p, b in zip(model.parameters(), before)) and calls["ema"] == ema_calls


def test_adamw_state_dict_carries_the_step_count_on_cpu():
    """The torch-composition path of optim.AdamW (CPU tensors): state_dict()['state'][i]['step'] = updates applied, loads
    into torch.optim.AdamW and back (tensor step), and the resumed run equals the uninterrupted one."""
    g = torch.Generator().manual_seed(0)
    init = torch.randn(6, 5, generator=g)
    grads = [torch.randn(6, 5, generator=g) for _ in range(6)]

    def run(opt, p, gs):
        for gr in gs:
            p.grad = gr.clone()
            opt.step()

    ref = nn.Parameter(init.clone())
    run(torch.optim.AdamW([ref], lr=1e-2, weight_decay=0.05), ref, grads)
    a = nn.Parameter(init.clone())
    oa = sm.AdamW([a], lr=1e-2, weight_decay=0.05)
    run(oa, a, grads[:3])
    sd = oa.state_dict()
    assert int(sd["state"][0]["step"]) == 3
    t = nn.Parameter(a.detach().clone())
    ot = torch.optim.AdamW([t], lr=1e-2, weight_decay=0.05)
    ot.load_state_dict(sd)
    run(ot, t, grads[3:4])
    b = nn.Parameter(t.detach().clone())
    ob = sm.AdamW([b], lr=1e-2, weight_decay=0.05)
    ob.load_state_dict(ot.state_dict())          # torch's tensor 'step'
    assert ob.state_dict()["state"][0]["step"] == 4
    run(ob, b, grads[4:])
    assert torch.allclose(b.detach(), ref.detach(), rtol=0, atol=1e-6)


def test_foreign_optimizer_step_bumps_parameter_versions():
    """optim._foreign_step_hook (a global optimizer step post-hook): an optimizer that writes through ``p.data`` leaves ``_version``
    where it was -- the key of the 16-bit weight images -- so the hook moves it; the package's own AdamW is exempt (its fused step
    refreshes the images itself)."""
    import slim_switch_moe_vit_amd as sm
    from slim_switch_moe_vit_amd import optim

    class DataSGD(torch.optim.Optimizer):
        def __init__(self, params):
            super().__init__(params, dict(lr=0.1))

        def step(self, closure=None):
            for g in self.param_groups:
                for p in g["params"]:
                    p.data.add_(p.grad.data, alpha=-g["lr"])

    p = torch.nn.Parameter(torch.ones(4))
    p.grad = torch.ones(4)
    v0 = p._version
    DataSGD([p]).step()
    assert p._version > v0 and torch.allclose(p.detach(), torch.full((4,), 0.9))
    assert optim._FOREIGN_HOOK is not None and getattr(sm.AdamW, "_slimmoe_refreshes_images", False)
    seen = []
    real = optim._bump_versions
    optim._bump_versions = lambda ts: seen.append(len(ts))
    try:
        optim._foreign_step_hook(types.SimpleNamespace(_slimmoe_refreshes_images=True, param_groups=[{"params": [p]}]), (), {})
        assert seen == []
        optim._foreign_step_hook(types.SimpleNamespace(param_groups=[{"params": [p]}]), (), {})
        assert seen == [1]
    finally:
        optim._bump_versions = real


def test_slot_table_layout_and_all_to_all_splits():
    """ep._SlotTable: global expert e owns caps[e] payload rows + one header row of the send buffer, regions in expert order; the
    all-to-all splits follow from the table alone (no count exchange): what rank r sends to rank w = w's experts' regions; what it
    receives from every source = its own experts' regions."""
    from slim_switch_moe_vit_amd import ep
    caps = [5, 1, 7, 3, 2, 9]
    for rank in range(3):
        t = ep._SlotTable(caps, rank, 2, "cpu")
        base = t.base_dev.tolist()
        assert base == [0, 6, 8, 16, 20, 23, 33] and t.rows == 33
        assert t.in_splits == [8, 12, 13]
        own = t.in_splits[rank]
        assert t.out_splits == [own] * 3 and t.recv_rows == 3 * own
        lb = t.lbase_dev.tolist()
        assert lb == [0, caps[2 * rank] + 1, caps[2 * rank] + caps[2 * rank + 1] + 2]
    assert ep._SlotTable([0, -3], 0, 1, "cpu").caps == [1, 1]            # a slot never has fewer than one payload row


def test_harness_switch_parsing():
    from slim_switch_moe_vit_amd import engine
    assert engine._speculative_alpha(None) is None and engine._speculative_alpha(0) is None
    assert engine._speculative_alpha(1.3) == 1.3
    old = os.environ.pop("SLIMMOE_EP_ALPHA", None)
    try:
        assert engine._speculative_alpha("auto") == 1.5
        os.environ["SLIMMOE_EP_ALPHA"] = "0"
        assert engine._speculative_alpha("auto") is None
    finally:
        os.environ.pop("SLIMMOE_EP_ALPHA", None)
        if old is not None:
            os.environ["SLIMMOE_EP_ALPHA"] = old
    m = sm.create_model("moe_tiny_patch16_224_expert4_top1", depth=1, num_classes=10)
    assert not engine.GraphedForward.supported(m.eval(), "cpu")            # graphs are a GPU matter; CPU models run eagerly


def test_exchange_inline_decision(monkeypatch):
    """ep.exchange_inline: the exchanges go on the compute stream exactly when nothing could run beside them (one micro-batch, one
    chunk); SLIMMOE_EP_INLINE = 0 / 1 overrides."""
    from slim_switch_moe_vit_amd import ep

    class M:
        pass

    m = M()
    monkeypatch.delenv("SLIMMOE_EP_INLINE", raising=False)
    assert ep.exchange_inline(m) and ep.exchange_inline(m, 1) and not ep.exchange_inline(m, 2)
    m.ep_rows_div = 2
    assert not ep.exchange_inline(m)
    monkeypatch.setenv("SLIMMOE_EP_INLINE", "1")
    assert ep.exchange_inline(m, 3)
    monkeypatch.setenv("SLIMMOE_EP_INLINE", "0")
    m.ep_rows_div = 1
    assert not ep.exchange_inline(m)
    # the model-level form (what decides whether a forward can be captured) looks at the model's setting, not at what the previous
    # forward left on the modules
    model = M()
    assert not ep.inline_possible(model)
    monkeypatch.delenv("SLIMMOE_EP_INLINE")
    assert ep.inline_possible(model)
    model.ep_micro_batches = 2
    assert not ep.inline_possible(model)
    # slot tables carry a serial that is never re-used (a captured forward is keyed on it; id() of a freed table can come back)
    a, b = ep._SlotTable([3, 4], 0, 2, "cpu"), ep._SlotTable([3, 4], 0, 2, "cpu")
    assert b.serial > a.serial


def test_speculative_exchange_in_training_is_opt_in_per_harness():
    """ep.static_kind: a gate without a capacity takes the speculative static exchange in eval once ep.set_speculative switched it on, in
    TRAINING only while a harness that repeats void forwards holds the flag (set_speculative(train=True): engine.train_one_epoch
    sets and clears it); a capacity gate is "capacity" either way; ep.dynamic_only() switches both off."""
    import torch
    import slim_switch_moe_vit_amd as sm
    from slim_switch_moe_vit_amd import ep

    holder = torch.nn.Module()
    holder.naive = sm.FMoETransformerMLP(4, 64, 128, torch.nn.GELU(), top_k=2)
    holder.switch = sm.FMoETransformerMLP(4, 64, 128, torch.nn.GELU(), top_k=1, gate="switch", capacity_factor=1.0)
    cd = torch.float16
    holder.eval()
    assert ep.static_kind(holder.naive, cd) is None and ep.static_kind(holder.switch, cd) == "capacity"
    assert ep.set_speculative(holder, 1.5) == 1                     # only the gate without a capacity is touched
    assert ep.static_kind(holder.naive, cd) == "speculative"
    holder.train()
    assert ep.static_kind(holder.naive, cd) is None and ep.static_kind(holder.switch, cd) == "capacity"
    ep.set_speculative(holder, 1.5, train=True)
    assert ep.static_kind(holder.naive, cd) == "speculative"
    with ep.dynamic_only():
        assert ep.static_kind(holder.naive, cd) is None and ep.static_kind(holder.switch, cd) is None
    ep.set_speculative(holder, 1.5, train=False)
    assert ep.static_kind(holder.naive, cd) is None
    ep.set_speculative(holder, None, train=True)                    # off is off
    holder.eval()
    assert ep.static_kind(holder.naive, cd) is None


def test_graph_harnesses_decide_capturability_of_expert_parallel_models_from_shared_settings(monkeypatch):
    """engine.GraphedForward.supported / GraphedTrainStep.supported under expert parallelism (pure host logic): the forward captures
    when every expert-parallel layer is on a static exchange AND the exchanges can sit on the compute stream (one micro-batch,
    SLIMMOE_EP_INLINE not 0) AND the caller allows it (ep_graph: a group of one rank, or hip_graph=True); the training step only with
    capacity gates (speculative slots need the host once per step)."""
    import torch
    import slim_switch_moe_vit_amd as sm
    from slim_switch_moe_vit_amd import ep

    monkeypatch.delenv("SLIMMOE_EP_INLINE", raising=False)

    def model(gate):
        kw = dict(gate="switch", capacity_factor=1.0) if gate == "switch" else {}
        m = sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=10, depth=1, **kw)
        for blk in m.blocks:
            blk.mlp.force_ep = True
        m.ep_micro_batches = 1
        return m.eval()

    naive, switch = model("naive"), model("switch")
    assert not sm.GraphedForward.supported(naive, "cuda")             # counted exchange: a host round trip per layer
    ep.set_speculative(naive, 1.5)
    assert sm.GraphedForward.supported(naive, "cuda") and sm.GraphedForward.supported(switch, "cuda")
    assert not sm.GraphedForward.supported(naive, "cuda", ep_graph=False)
    naive.ep_micro_batches = 2                                        # pipelined: the exchanges stay on RCCL's stream
    assert not sm.GraphedForward.supported(naive, "cuda")
    naive.ep_micro_batches = 1
    monkeypatch.setenv("SLIMMOE_EP_INLINE", "0")
    assert not sm.GraphedForward.supported(naive, "cuda")
    monkeypatch.delenv("SLIMMOE_EP_INLINE")
    assert not sm.GraphedForward.supported(naive.train(), "cuda")     # (a training-mode model is not the eval harness' business)
    # the training step
    opt_n, opt_s = sm.AdamW(naive.parameters(), lr=1e-3), sm.AdamW(switch.parameters(), lr=1e-3)
    scaler = sm.NativeScaler()
    switch.train()
    switch.ep_micro_batches = 2                                       # (irrelevant in training: never cut into micro-batches)
    assert sm.GraphedTrainStep.supported(switch, opt_s, scaler, "cuda", None, ep_graph=True)
    assert not sm.GraphedTrainStep.supported(switch, opt_s, scaler, "cuda", None)             # expert parallel: only when allowed
    ep.set_speculative(naive, 1.5, train=True)
    assert not sm.GraphedTrainStep.supported(naive, opt_n, scaler, "cuda", None, ep_graph=True)
    for blk in switch.blocks:
        blk.mlp.force_ep = False
    assert sm.GraphedTrainStep.supported(switch, opt_s, scaler, "cuda", None)                 # one rank: as before


def test_speculative_slots_leave_room_for_the_fluctuation_of_small_groups(monkeypatch):
    """ep._SlotState.fitted_caps: HEADROOM x the largest group seen, but at least SIGMA standard deviations of a count of that size
    above it -- 12 % is 9.6 sigma for a 6,300-row group (the bench) and under 2 sigma for a 288-row one (cfg 4's model at 16 images)."""
    import math
    from slim_switch_moe_vit_amd import ep

    class St(ep._SlotState):
        def __init__(self, obs):
            self.obs, self.headroom = obs, ep.HEADROOM
    monkeypatch.setattr(ep, "SIGMA", 4.0)
    caps = St([6300, 288, 0, 1]).fitted_caps()
    assert caps == [math.ceil(1.12 * 6300), math.ceil(288 + 4 * math.sqrt(288)), 1, 5]
    monkeypatch.setattr(ep, "SIGMA", 0.0)
    assert St([6300, 288, 0, 1]).fitted_caps() == [math.ceil(1.12 * 6300), math.ceil(1.12 * 288), 1, 2]
