"""smoe_expert_ffn: both expert GEMMs of a MoE layer (with the scatter in GEMM-1's operand fetch and the top-1 combine + residual
in GEMM-2's store) as ONE persistent launch whose workgroups walk slots that hold tiles of both GEMMs (csrc/gemm_persistent.h,
expert_ffn_fused; opt-in, SLIMMOE_FFN_FUSED=1: it measured slower than the two launches, profiles/r04_fused_ffn.md).  It must compute, tile for tile, what the two smoe_grouped_gemm launches compute: every test here is bit-wise
against them (which the parity tests pin against the oracle / float64), over routing patterns that stress the list order (empty
experts, one hot expert, fewer tiles than CUs, dropped tokens), repeated launches on the kept-zero workspace, and the model."""
import pytest
import torch

import slim_switch_moe_vit_amd as sm  # noqa: E402
from slim_switch_moe_vit_amd import ops  # noqa: E402

# the fused launch left the default build in round 5 (slower than the two launches in every design): these tests run against a
# library made with `make -C slim-switch-moe-vit_amd/csrc FFN=-DSMOE_FFN_FUSED` (SLIMMOE_LIB may point at it)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ops.ffn_fused_available(),
                                                  reason="smoe_expert_ffn is not in this build (make FFN=-DSMOE_FFN_FUSED)")]

DEV = "cuda:0"


def _gen(s):
    return torch.Generator().manual_seed(s)


def _case(T, d, h, E, dt, seed, skew=None, cap=-1, bias=True):
    g = _gen(seed)
    if skew == "one_hot":
        idx = torch.full((T, 1), E - 1, dtype=torch.int64)
    elif skew == "empty":                                    # experts 0 and 2 receive nothing
        idx = torch.randint(0, E, (T, 1), generator=g)
        idx[idx == 0] = 1
        idx[idx == 2] = min(3, E - 1)
    else:
        idx = torch.randint(0, E, (T, 1), generator=g)
    counts, offsets, pos, inv_pos, _ = ops.dispatch_plan(idx.to(DEV), E, cap)
    x = torch.randn(T, d, generator=g).to(dt).to(DEV)
    w1 = (torch.randn(E, h, d, generator=g) * 0.03).to(dt).to(DEV)
    w2 = (torch.randn(E, d, h, generator=g) * 0.03).to(dt).to(DEV)
    b1 = (torch.randn(E, h, generator=g) * 0.1).to(DEV) if bias else None
    b2 = (torch.randn(E, d, generator=g) * 0.1).to(DEV) if bias else None
    score = torch.rand(T, generator=g).to(DEV)
    res = torch.randn(T, d, generator=g).to(DEV)
    return x, w1, b1, w2, b2, offsets, pos, score, res


def _two_launches(x, w1, b1, w2, b2, offsets, pos, score, res, prefill):
    h = ops.grouped_gemm(x, w1, b1, offsets, ops.EPI_GELU, x.dtype, variant=9, a_gather=pos)
    out = prefill.clone()
    ops.grouped_gemm(h, w2, b2, offsets, ops.EPI_NONE, torch.float32, row_map=pos, row_scale=score, out=out, variant=9, residual=res)
    return h, out


@pytest.mark.parametrize("T,d,h,E,dt,skew,cap", [
    (50432, 768, 3072, 8, torch.float16, None, -1),          # BASELINE cfg 2, full size: 1,896 + 474 tiles on 256 CUs
    (50432, 768, 3072, 8, torch.float16, "one_hot", -1),     # one row group holds everything
    (20000, 768, 3072, 8, torch.float16, "empty", -1),       # empty row groups in the lane table
    (1576, 192, 768, 4, torch.float16, None, -1),            # cfg 1: fewer tiles than CUs
    (333, 192, 768, 4, torch.bfloat16, None, -1),            # a handful of tiles
    (12000, 1024, 4096, 32, torch.float16, None, -1),        # cfg 4's dims: 16 / 4 n-tiles, 32 row groups
    (9000, 768, 3072, 8, torch.float16, None, 700),          # capacity: dropped tokens keep their pre-filled rows
    (7000, 384, 1536, 1, torch.bfloat16, None, -1),          # E = 1
    (5000, 768, 3072, 8, torch.float16, None, -1),
])
def test_fused_expert_ffn_is_bitwise_the_two_grouped_gemm_launches(T, d, h, E, dt, skew, cap):
    args = _case(T, d, h, E, dt, seed=T + E, skew=skew, cap=cap)
    x, w1, b1, w2, b2, offsets, pos, score, res = args
    prefill = res.clone()                                   # rows of dropped tokens stay what they were
    h_ref, out_ref = _two_launches(*args, prefill)
    out = prefill.clone()
    H = torch.zeros_like(h_ref)
    got = ops.expert_ffn(x, w1, b1, w2, b2, offsets, out, a_gather=pos, row_map=pos, row_scale=score, residual=res, H=H)
    assert got is not None, "this shape is inside the fused launch's reach"
    torch.cuda.synchronize()
    n_rows = int(offsets[-1])
    assert torch.equal(H[:n_rows], h_ref[:n_rows]), "GEMM-1 rows (the hidden activations)"
    assert torch.equal(out, out_ref), "GEMM-2 rows (combine + residual)"
    assert not ops.ffn_workspace_error(x.device)
    ws = next(t for (dv, _), t in ops._ffn_ws.items() if dv == 0).view(torch.int32)
    assert int(ws.abs().sum()) == 0, "the launch leaves its workspace zero"



def test_fused_expert_ffn_without_bias_scale_and_residual_and_in_place_residual():
    T, d, h, E = 6000, 768, 3072, 8
    x, w1, _, w2, _, offsets, pos, score, res = _case(T, d, h, E, torch.float16, 5, bias=False)
    hh = ops.grouped_gemm(x, w1, None, offsets, ops.EPI_GELU, x.dtype, variant=9, a_gather=pos)
    ref = torch.empty(T, d, device=DEV)
    ops.grouped_gemm(hh, w2, None, offsets, ops.EPI_NONE, torch.float32, row_map=pos, out=ref, variant=9)
    out = torch.empty(T, d, device=DEV)
    assert ops.expert_ffn(x, w1, None, w2, None, offsets, out, a_gather=pos, row_map=pos) is not None
    assert torch.equal(out, ref)
    # residual and out the same tensor (the residual-MoE block's in-place form)
    ref2 = res.clone()
    ops.grouped_gemm(hh, w2, None, offsets, ops.EPI_NONE, torch.float32, row_map=pos, row_scale=score, out=ref2, variant=9, residual=ref2)
    out2 = res.clone()
    assert ops.expert_ffn(x, w1, None, w2, None, offsets, out2, a_gather=pos, row_map=pos, row_scale=score, residual=out2) is not None
    assert torch.equal(out2, ref2)


def test_fused_expert_ffn_repeated_launches_and_changing_routing_reuse_the_workspace():
    """60 launches back to back, the routing (hence the number of m-tiles and the list geometry) changing every time: the row
    counters of one launch must never leak into the next (the last workgroup to leave clears them)."""
    T, d, h, E = 30000, 768, 3072, 8
    x, w1, b1, w2, b2, _, _, score, res = _case(T, d, h, E, torch.float16, 11)
    outs, refs = [], []
    plans = []
    for i in range(6):
        idx = torch.randint(0, E if i % 2 == 0 else E // 2, (T - 997 * i, 1), generator=_gen(100 + i)).to(DEV)
        plans.append(ops.dispatch_plan(idx, E))
    for rep in range(10):
        for i, (counts, offsets, pos, inv_pos, _) in enumerate(plans):
            n = pos.numel()
            out = torch.empty(n, d, device=DEV)
            assert ops.expert_ffn(x[:n], w1, b1, w2, b2, offsets, out, a_gather=pos, row_map=pos, row_scale=score[:n],
                                  residual=res[:n]) is not None
            if rep == 0:
                refs.append(_two_launches(x[:n], w1, b1, w2, b2, offsets, pos, score[:n], res[:n], res[:n])[1])
            outs.append((i, out))
    torch.cuda.synchronize()
    for i, out in outs:
        assert torch.equal(out, refs[i]), i
    assert not ops.ffn_workspace_error(x.device)


def test_two_fused_launches_side_by_side_on_two_streams_do_not_deadlock():
    """Two fused launches in flight at once (two compute streams; or two ranks sharing a device): each has fewer resident
    workgroups than its grid while the other runs.  Positions are only ever claimed by RUNNING workgroups, so both finish."""
    T, d, h, E = 50432, 768, 3072, 8
    a = _case(T, d, h, E, torch.float16, 21)
    b = _case(T, d, h, E, torch.float16, 22)
    refs = [_two_launches(*c, c[-1])[1] for c in (a, b)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty(T, d, device=DEV) for _ in range(2)]
    for rep in range(5):
        for st, c, out in zip(streams, (a, b), outs):
            with torch.cuda.stream(st):
                x, w1, b1, w2, b2, offsets, pos, score, res = c
                assert ops.expert_ffn(x, w1, b1, w2, b2, offsets, out, a_gather=pos, row_map=pos, row_scale=score, residual=res) is not None
    for st in streams:
        st.synchronize()
    assert torch.equal(outs[0], refs[0]) and torch.equal(outs[1], refs[1])
    assert not ops.ffn_workspace_error(torch.device(DEV))


@pytest.mark.parametrize("name,kw", [("moe_base_patch16_224_expert8_top1", {}),
                                     ("resmoe_base_patch16_224_expert8_top1", dict(starting_threshold=0.6, target_threshold=0.5))])
def test_model_forward_is_bitwise_the_same_with_and_without_the_fused_launch(name, kw):
    torch.manual_seed(0)
    model = sm.create_model(name, depth=3, num_classes=1000, **kw)
    g = _gen(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() >= 2 and "norm" not in n:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 if "gate.head" in n else 0.02))
    model = model.to(DEV).eval()
    images = torch.randn(16, 3, 224, 224, generator=g).to(DEV)

    def run():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model(images)
    was = ops.FFN_FUSED
    try:
        ops.FFN_FUSED = True
        fused = run()
        ops.FFN_FUSED = False
        split = run()
    finally:
        ops.FFN_FUSED = was
    assert torch.equal(fused, split)
