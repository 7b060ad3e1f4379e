"""Model-level parity on the GPU: BASELINE cfg 1 (ViT-Ti/16, E=4, top-1, 224^2, batch 8) through the eval harness,
HIP path vs the CPU oracle's vit_forward on the same weights and images; plus the resmoe (token-skip gate) variant."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import moe_oracle as mo  # noqa: E402
from _mp import join_or_kill as _join_or_kill, float_bar as _float_bar  # noqa: E402
import slim_switch_moe_vit_amd as sm  # noqa: E402

DEV = "cuda:0"


def _init(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for blk in model.blocks:
            m = blk.mlp
            m.gate.gate.weight.copy_(torch.randn(m.gate.gate.weight.shape, generator=g) * 0.1)
            m.gate.gate.bias.zero_()
            m.experts.htoh4.weight.copy_(torch.randn(m.experts.htoh4.weight.shape, generator=g) * 0.02)
            m.experts.h4toh.weight.copy_(torch.randn(m.experts.h4toh.weight.shape, generator=g) * 0.02)
        model.head.weight.copy_(torch.randn(model.head.weight.shape, generator=g) * 0.02)
    return model


def _block_halves(blk, x):
    """Block.forward (stock form, no stochastic depth) in its two halves through the product code paths, returning
    the MoE half's input as well: (x + attn(norm1(x)), that + mlp(norm2(that)))."""
    a, added = blk.attn(blk._norm1(x), residual=x)
    mid = a if added else x + a
    return mid, blk.mlp.forward_norm_add(mid, blk.norm2)


def _router_logits64(blk, mid):
    """float64 router logits of the MoE half's input (LayerNorm norm2, then gate.gate), for attribution only."""
    n, g = blk.norm2, blk.mlp.gate.gate
    xn = torch.nn.functional.layer_norm(mid.double(), (mid.shape[-1],), n.weight.double(), n.bias.double(), n.eps)
    return xn.reshape(-1, mid.shape[-1]) @ g.weight.double().t() + g.bias.double()


def _flip_attribution(full, part, images, before_full=None, before_part=None, pre=5e-3, post=2e-2):
    """Runs two models that compute the same function with different rounding (single rank vs expert parallel: the
    exchanged rows travel in 16 bit) END TO END, each on its own activations, and attributes every difference:

    * while no token of an image has been routed differently, the image's activations agree to ``pre`` (``post`` once
      other images have diverged -- they cannot influence this one, the looser bound only covers accumulated rounding);
    * every token whose routing differs ("flip") sits on a routing boundary: the gap between its two largest router
      logits is no larger than twice the perturbation of its logits between the two models (anything else -- a wrong
      row, a wrong expert's weights, a layout error -- shows up as a flip whose gap is NOT explained, or as an
      un-flipped token that disagrees);
    * an image is excluded from later comparisons from the block in which one of its tokens flipped (attention mixes
      the differing expert output into the whole image), and the final outputs of the untouched images agree.

    Returns {"flips": [(block, token, gap, logit_perturbation)], "clean_images": n, "final_err": e}."""
    B = images.shape[0]
    xf, xp = full._embed(images), part._embed(images)
    N = xf.shape[1]
    clean = torch.ones(B, dtype=torch.bool, device=xf.device)
    flips, worst = [], 0.0
    for i, (bf, bp) in enumerate(zip(full.blocks, part.blocks)):
        if before_full is not None:
            before_full()
        mf, of = _block_halves(bf, xf)
        idx_f = bf.mlp.last_plan[0].clone()
        if before_part is not None:
            before_part()
        mp_, op = _block_halves(bp, xp)
        idx_p = bp.mlp.last_plan[0].clone()
        bound = pre if bool(clean.all()) else post
        err_mid = float((mf - mp_)[clean].abs().max()) if bool(clean.any()) else 0.0
        assert err_mid <= bound, (i, "MoE input of un-flipped images", err_mid)
        lf, lp = _router_logits64(bf, mf), _router_logits64(bp, mp_)
        differ = (idx_f != idx_p).any(dim=1).reshape(B, N) & clean[:, None]
        for t in differ.reshape(-1).nonzero().reshape(-1).tolist():
            top2 = lf[t].topk(2).values
            gap, pert = float(top2[0] - top2[1]), float((lf[t] - lp[t]).abs().max())
            assert gap <= 2 * pert + 1e-7, (i, t, "flip not explained by the logit perturbation", gap, pert)
            assert pert <= 10 * bound, (i, t, "logit perturbation too large for 16-bit payload rounding", pert)
            flips.append((i, t, gap, pert))
        same = clean[:, None] & ~differ
        sel = same.reshape(B, N, 1).expand_as(of)
        if bool(sel.any()):
            err_out = float((of - op)[sel].abs().max())
            assert err_out <= bound, (i, "block output of un-flipped tokens", err_out)
            worst = max(worst, err_out)
        clean = clean & ~differ.any(dim=1)
        xf, xp = of, op
    outs = []
    for m, x in ((full, xf), (part, xp)):
        f = m.pre_logits(m._final_norm_cls(x))
        outs.append(m.head(f).float())
    final = float((outs[0] - outs[1])[clean].abs().max()) if bool(clean.any()) else 0.0
    assert final <= post, ("final logits of un-flipped images", final)
    return {"flips": flips, "clean_images": int(clean.sum()), "final_err": final, "worst_block_err": worst}


def _oracle_block(x, p, heads, k, eps=1e-6):
    """oracle.block_forward (stock form, models/vision_transformer.py:319-322) in its two halves, keeping what the
    attribution needs: (MoE-half input, block output, routing indices)."""
    F = torch.nn.functional
    d = x.shape[-1]
    mid = x + mo.attention(F.layer_norm(x, (d,), p["norm1.weight"], p["norm1.bias"], eps), p["attn.qkv.weight"],
                           p["attn.qkv.bias"], p["attn.proj.weight"], p["attn.proj.bias"], heads)
    r = mo.moe_forward(F.layer_norm(mid, (d,), p["norm2.weight"], p["norm2.bias"], eps), p["mlp.gate.gate.weight"],
                       p["mlp.gate.gate.bias"], p["mlp.experts.htoh4.weight"], p["mlp.experts.htoh4.bias"],
                       p["mlp.experts.h4toh.weight"], p["mlp.experts.h4toh.bias"], k)
    return mid, mid + r.out, r.idx


@pytest.mark.parametrize("name,depth,heads,B", [("moe_tiny_patch16_224_expert4_top1", 12, 3, 8),
                                                ("moe_base_patch16_224_expert8_top1", 4, 12, 6)])
def test_fp16_gpu_blocks_against_the_fp32_oracle_every_difference_attributed(name, depth, heads, B):
    """The benchmarked mode (fp16 autocast on the GPU) against the fp32 CPU oracle, block by block, with every
    difference attributed instead of bounded by a loose whole-model tolerance (bench.py's `parity` block reports 2 of
    128 images off by up to 1.3 in a logit: routing flips, or an error?).

    (1) TEACHER-FORCED: every GPU block gets the ORACLE's block input.  A token whose routing differs from the
        oracle's must sit on a routing boundary -- the gap between the oracle's two largest router logits is no
        larger than twice the measured perturbation of that token's logits (f64 logits of the GPU block's MoE-half
        input vs of the oracle's; the perturbation is the fp16 attention half) -- and every OTHER token's block output
        meets the float bar (`_mp.float_bar`: max |diff| <= 1e-3 max(1, max |ref|) and relative L2 <= 1e-3).
    (2) FREE-RUNNING: the GPU model on its own activations; an image stays "clean" until one of its tokens is routed
        differently from the oracle's token (each such flip explained as in (1)); the final logits of the clean images
        agree with the oracle to 2e-2, so whatever differs by more is an image with an explained flip.
    Reference semantics: models/vision_transformer.py:319-322 (block), FastMoE routing as restated in SURVEY Appendix B."""
    torch.manual_seed(0)
    model = _init(sm.create_model(name, num_classes=100, depth=depth), 31).eval()
    sd = {k_: v.detach().clone() for k_, v in model.state_dict().items()}
    images = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(32))
    model = model.to(DEV)
    F = torch.nn.functional
    x_o = F.conv2d(images, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16).flatten(2).transpose(1, 2)
    x_o = torch.cat((sd["cls_token"].expand(B, -1, -1), x_o), dim=1) + sd["pos_embed"]
    N = x_o.shape[1]
    forced_flips, free_flips, worst_forced = [], [], 0.0
    clean = torch.ones(B, dtype=torch.bool)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        x_g = model._embed(images.to(DEV))
        assert (x_g.cpu() - x_o).abs().max().item() <= 2e-3 * max(1.0, float(x_o.abs().max()))
        for i, blk in enumerate(model.blocks):
            pre = f"blocks.{i}."
            p = {key[len(pre):]: v for key, v in sd.items() if key.startswith(pre)}
            mid_o, out_o, idx_o = _oracle_block(x_o, p, heads, 1)
            l_o = _router_logits64(blk, mid_o.to(DEV)).cpu()
            top2 = l_o.topk(2, dim=1).values
            gap = (top2[:, 0] - top2[:, 1])

            def explain(mid_dev, idx_dev, where, sink, only=None):
                pert = (_router_logits64(blk, mid_dev).cpu() - l_o).abs().max(dim=1).values
                differ = (idx_dev.cpu() != idx_o).any(dim=1)
                if only is not None:
                    differ = differ & only
                for t in differ.nonzero().reshape(-1).tolist():
                    assert float(gap[t]) <= 2 * float(pert[t]) + 1e-7, (i, t, where, "flip not explained", float(gap[t]), float(pert[t]))
                    sink.append((i, t, float(gap[t]), float(pert[t])))
                return differ, pert

            # (1) teacher-forced on the oracle's block input
            mid_t, out_t = _block_halves(blk, x_o.to(DEV))
            differ_t, pert_t = explain(mid_t, blk.mlp.last_plan[0], "teacher-forced", forced_flips)
            assert float(pert_t.max()) <= 5e-2, (i, "logit perturbation of an fp16 attention half", float(pert_t.max()))
            same = ~differ_t
            got, ref = out_t.cpu().reshape(-1, out_t.shape[-1])[same], out_o.reshape(-1, out_o.shape[-1])[same]
            worst_forced = max(worst_forced, _float_bar(got, ref, 1e-3)[0] / max(1.0, float(ref.abs().max())))
            # (2) free-running
            mid_g, out_g = _block_halves(blk, x_g)
            differ_g, _ = explain(mid_g, blk.mlp.last_plan[0], "free-running", free_flips, only=clean.repeat_interleave(N))
            clean = clean & ~differ_g.reshape(B, N).any(dim=1)
            x_o, x_g = out_o, out_g
        logits_g = model.head(model.pre_logits(model._final_norm_cls(x_g))).float().cpu()
    d = x_o.shape[-1]
    logits_o = F.linear(F.layer_norm(x_o, (d,), sd["norm.weight"], sd["norm.bias"], 1e-6)[:, 0], sd["head.weight"], sd["head.bias"])
    err = (logits_g - logits_o).abs().max(dim=1).values
    print(f"{name}: teacher-forced flips {forced_flips}; worst un-flipped block error / scale {worst_forced:.2e}; "
          f"free-running flips {free_flips}; clean images {int(clean.sum())} / {B}; per-image logit error {err.tolist()}")
    assert int(clean.sum()) >= 2
    assert float(err[clean].max()) <= 2e-2, ("final logits of images without a flipped token", err.tolist(), clean.tolist())


@pytest.mark.parametrize("autocast,cd,tol", [(False, torch.float32, 2e-3), (True, None, 5e-2)])
def test_cfg1_vit_tiny_e4_top1_through_eval_harness(autocast, cd, tol):
    torch.manual_seed(0)
    kw = {"compute_dtype": cd} if cd is not None else {}
    model = _init(sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=100, **kw), 1).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    images = torch.randn(8, 3, 224, 224, generator=g)
    target = torch.randint(0, 100, (8,), generator=g)
    ref = mo.vit_forward(images, sd, depth=12, num_heads=3, k=1, residual_moe=False)
    model = model.to(DEV)
    stats = sm.evaluate([(images, target)], model, DEV, autocast=autocast)
    assert set(stats) >= {"loss", "acc1", "acc5", "images_per_sec"}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=autocast):
        out = model(images.to(DEV)).float().cpu()
    err = (out - ref).abs().max().item()
    assert err <= tol, err
    ref_loss = torch.nn.functional.cross_entropy(ref, target).item()
    assert abs(stats["loss"] - ref_loss) < 10 * tol


def test_resmoe_factory_forward_matches_oracle_block_semantics():
    """resmoe_tiny_patch16_224_expert8 (E=8, top-2, token-skip gates, residual on the normed activations):
    eval forward on the GPU vs the oracle with the same state dict; gates at the reference defaults
    (target threshold 0.9) skip only a few tokens but those must route as all-zero rows."""
    torch.manual_seed(0)
    model = _init(sm.create_model("resmoe_tiny_patch16_224_expert8", num_classes=10, compute_dtype=torch.float32,
                                  starting_threshold=1.0, target_threshold=0.9), 3).eval()
    with torch.no_grad():
        for blk in model.blocks:  # make the skip gates fire on a visible fraction of tokens
            blk.moe_gate.head[1].bias.fill_(1.5)
            blk.dense_gate.head[1].bias.fill_(1.5)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    images = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    ref = mo.vit_forward(images, sd, depth=12, num_heads=3, k=2, residual_moe=True)
    model = model.to(DEV)
    with torch.no_grad():
        out = model(images.to(DEV)).cpu()
    assert model.blocks[0].moe_gate._skipped_tokens > 0
    assert (out - ref).abs().max().item() <= 5e-3


def test_cfg4_vit_large_384_e32_shapes_run_and_match_oracle():
    """BASELINE cfg 4 shapes (ViT-L/16 @384: 577 tokens, d 1024, h 4096, E=32, top-1) on a 1-block copy of the model:
    general router (E > 8), the long-sequence attention kernel (N = 577, under autocast), default GEMM variant; vs the
    oracle."""
    torch.manual_seed(0)
    model = _init(sm.create_model("moe_large_patch16_384_expert32_top1", num_classes=10, depth=1,
                                  compute_dtype=torch.float32), 9).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    images = torch.randn(2, 3, 384, 384, generator=torch.Generator().manual_seed(4))
    ref = mo.vit_forward(images, sd, depth=1, num_heads=16, k=1, residual_moe=False)
    model = model.to(DEV)
    with torch.no_grad():
        out = model(images.to(DEV)).cpu()
        with torch.autocast("cuda", dtype=torch.float16):
            out16 = model(images.to(DEV)).float().cpu()
    assert (out - ref).abs().max().item() <= 2e-3
    assert (out16 - ref).abs().max().item() <= 5e-2


def test_expert_parallel_micro_batch_pipeline_equals_plain_forward():
    """Expert-parallel inference interleaves micro-batches of the local batch through the whole model
    (VisionTransformer._forward_features_pipelined, ep.ep_forward_steps).  Images are independent in eval mode, so
    the pipelined forward must reproduce the single-stream forward: routing bit-exact, outputs to fp16 rounding
    (the GEMM tile schedule differs with the row count).  Driven here on one GPU through a one-rank RCCL group."""
    import socket
    import torch.distributed as dist

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        torch.manual_seed(0)
        model = _init(sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=100), 7).eval().to(DEV)
        images = torch.randn(10, 3, 224, 224, generator=torch.Generator().manual_seed(8)).to(DEV)
        plans = {}

        def run(force_ep, n_micro):
            for blk in model.blocks:
                blk.mlp.force_ep = force_ep
            model.ep_micro_batches = n_micro
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                out = model(images).float()
            return out, [blk.mlp.last_plan[0].clone() for blk in model.blocks]

        ref, _ = run(False, 1)
        ep1, idx1 = run(True, 1)
        with torch.no_grad():
            assert model._ep_pipeline_depth(images) == 1
        for n in (2, 3):
            model.ep_micro_batches = n
            for blk in model.blocks:
                blk.mlp.force_ep = True
            with torch.no_grad():
                assert model._ep_pipeline_depth(images) == n
            got, idxn = run(True, n)
            assert got.shape == ref.shape
            # same model, same per-image arithmetic: the pipeline reproduces the un-pipelined expert-parallel forward
            assert (got - ep1).abs().max().item() <= 1e-3
            tail = idxn[-1].reshape(-1)   # last micro-batch's routing of the last block = tail of the whole batch's
            assert torch.equal(tail, idx1[-1].reshape(-1)[-tail.numel():])
        # token chunks inside a MoE layer (chunk c + 1's exchange under chunk c's expert GEMMs): every row's arithmetic is its own, so
        # the chunked forward is the un-chunked one BIT FOR BIT -- also with the next block's LayerNorm riding on each chunk's combine
        for chunks in (2, 3):
            for blk in model.blocks:
                blk.mlp.ep_chunks = chunks
            got, idxc = run(True, 1)
            assert torch.equal(got, ep1), chunks
            assert all(torch.equal(a, b) for a, b in zip(idxc, idx1))
        for blk in model.blocks:
            blk.mlp.ep_chunks = 1
        # against the single-rank path the exchanged rows round differently (16-bit payload), so a token that sits on a
        # routing boundary may flip: every difference is attributed, end to end (see _flip_attribution)
        model.ep_micro_batches = 1

        def _set(flag):
            def f():
                for blk in model.blocks:
                    blk.mlp.force_ep = flag
            return f

        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            att = _flip_attribution(model, model, images, before_full=_set(False), before_part=_set(True))
        print("flip attribution (one-rank EP vs single-rank path):", att)
        assert att["clean_images"] >= 2
        # training / grad mode and dropping gates never pipeline
        model.train()
        with torch.no_grad():
            assert model._ep_pipeline_depth(images) == 1
        model.eval()
        with torch.enable_grad():
            assert model._ep_pipeline_depth(images) == 1
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, q):
    """One expert-parallel rank; both ranks share cuda:0 and talk over gloo (ep._a2a stages through the host)."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        E, E_local = 4, 4 // world
        torch.manual_seed(0)
        full = _init(sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=50), 11).eval()
        torch.manual_seed(0)
        part = sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=50, world_size=world).eval()
        sd = full.state_dict()
        sl = slice(rank * E_local, (rank + 1) * E_local)
        for k in list(sd):
            if ".experts." in k:   # [E, ...] expert tensors: this rank keeps its slice
                sd[k] = sd[k][sl].clone()
        part.load_state_dict(sd)
        full, part = full.to(DEV), part.to(DEV)
        images = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(100 + rank)).to(DEV)
        res = {}
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            # (1) block by block on the SAME input (the single-rank model's activations): the two models round
            # differently (the exchanged rows are 16-bit), and over 12 layers one token in ~10^4 sits close enough to a
            # routing boundary to flip on a 1e-4 difference, which an end-to-end comparison would report as an error
            x = full._embed(images)
            blk_err, route_equal = 0.0, True
            for bf, bp in zip(full.blocks, part.blocks):
                yf, yp = bf(x), bp(x)
                route_equal &= bool(torch.equal(bf.mlp.last_plan[0], bp.mlp.last_plan[0]))
                blk_err = max(blk_err, float((yf - yp).abs().max()))
                x = yf
            res["blocks"] = blk_err
            res["route_equal"] = route_equal
            # (1b) END TO END, each model on its own activations, every difference attributed (gpurun_out/t25.log of
            # round 1: rank 3 of 4 differed from the single-rank model by 0.83 in one logit -- a routing flip or a bug?)
            res["attribution"] = _flip_attribution(full, part, images)
            # (2) the micro-batch pipeline reproduces the un-pipelined expert-parallel forward of the same model
            part.ep_micro_batches = 1
            base = part(images).float()
            for n in (2, 3):
                part.ep_micro_batches = n
                assert part._ep_pipeline_depth(images) == n
                res[n] = float((part(images).float() - base).abs().max())
            # (3) token chunks inside a layer (chunk c + 1's exchange under chunk c's expert GEMMs, the next block's LayerNorm on
            # every chunk's combine): every row's arithmetic is its own -- bit for bit the un-chunked forward, on every rank
            part.ep_micro_batches = 1
            for blk in part.blocks:
                blk.mlp.ep_chunks = 2
            res["chunks_bitwise"] = bool(torch.equal(part(images).float(), base))
            for blk in part.blocks:
                blk.mlp.ep_chunks = 1
            # (4) the speculative static exchange through the same pipeline shapes: slots shared by the micro-batches of a step (three
            # of them put 36 stats matrices into one forward: more than the deferred watch's lag, so it reads some of them MID-forward
            # and must not cut the slots to those).  Whatever gets repeated on the way, the third guarded forward fits and gives the
            # counted exchange's numbers (bit for bit un-pipelined; to the pipeline's own rounding otherwise)
            from slim_switch_moe_vit_amd import ep
            ep.set_speculative(part, 1.25)
            spec = {}
            for n in (1, 2, 3):
                part.ep_micro_batches = n
                outs = [ep.run_guarded(lambda: part(images).float()) for _ in range(3)]
                spec[n] = (float((outs[-1][0] - base).abs().max()), [r for _, r in outs])
            res["speculative"] = spec
            ep.set_speculative(part, None)
            part.ep_micro_batches = 1
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_expert_parallel_ranks_on_one_gpu_match_single_rank_model(world):
    """The whole W = 2 / W = 4 inference path (router over all experts, count exchange, all-to-all-v in both directions with
    the [source rank][local expert] receive layout, group -> expert GEMMs, combine, and the micro-batch pipeline that
    interleaves collectives of several micro-batches) as W processes on one GPU; every block of each rank's model must
    match the block of the single-rank model holding all four experts on the same input, with identical routing
    (W = 4: one expert per rank, as the 8-GPU bench has), and the pipelined forward must reproduce the plain one."""
    import socket
    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    _join_or_kill(procs, 300)
    got = dict(q.get(timeout=10) for _ in range(world))
    assert sorted(got) == list(range(world))
    for rank, res in got.items():
        print(f"rank {rank}/{world}: {res}")
        assert res["route_equal"], rank
        assert res["blocks"] <= 5e-3, (rank, res)
        assert res[2] <= 1e-3 and res[3] <= 1e-3, (rank, res)
        assert res["chunks_bitwise"], (rank, res)
        assert res["attribution"]["clean_images"] >= 2, (rank, res)
        for n, (err, repeats) in res["speculative"].items():
            assert not repeats[-1], (rank, n, res)                       # adapted: the third guarded forward fits
            assert err == 0.0 if n == 1 else err <= 1e-3, (rank, n, res)



def _two_stream_worker(q):
    """Two compute streams (VisionTransformer.compute_streams = 2): each half of the batch must come out bit for bit
    as it does alone on one stream, over several steps, with the derived tensors (16-bit weight shadows, the constant
    offset tables keyed by the HALF batch's row count) first produced inside the two-stream run."""
    torch.manual_seed(0)
    model = _init(sm.create_model("moe_base_patch16_224_expert8_top1", num_classes=100, depth=4), 21).eval().to(DEV)
    images = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(22)).to(DEV)
    worst = 0.0
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        model.compute_streams = 2
        first = model(images).float()          # caches are cold here: filled on one side stream, used by the other
        for _ in range(25):
            got = model(images).float()
            worst = max(worst, float((got - first).abs().max()))
        model.compute_streams = 1
        alone = torch.cat([model(images[:16]).float(), model(images[16:]).float()], 0)
    torch.cuda.synchronize()
    q.put({"repeat_err": worst, "vs_single_stream": float((first - alone).abs().max())})


def test_two_compute_streams_reproduce_the_single_stream_forward():
    """Round 1 saw 2 of ~20 two-stream probe runs never finish.  What the code had that could explain it: derived
    tensors (fp16 weight shadows, the [0, rows] int32 offset table of the single-group GEMMs, keyed by the half
    batch's row count) were produced on whichever stream arrived first and consumed on the other WITHOUT an event --
    a GEMM reading a not-yet-written offset table runs with wild row ranges; the router's redo pass took its trip
    count from device memory unclamped; per-launcher `static bool` attribute flags.  All three are fixed
    (_cache.StreamCache, clamps in router16.hip / router.hip, smoe_init + per-device atomic bits); this runs the
    two-stream forward ONCE, in a child process with a finite timeout."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_two_stream_worker, args=(q,))
    p.start()
    _join_or_kill([p], 240)
    res = q.get(timeout=10)
    print("two compute streams:", res)
    assert res["repeat_err"] == 0.0 and res["vs_single_stream"] == 0.0, res


def _graph_worker(q):
    """Capture the eval forward into a HIP graph and replay it: include/slimmoe.h promises launch functions that neither
    allocate nor synchronise, i.e. capturable ones."""
    torch.manual_seed(0)
    model = _init(sm.create_model("moe_base_patch16_224_expert8_top1", num_classes=100, depth=3), 23).eval().to(DEV)
    images = torch.randn(16, 3, 224, 224, generator=torch.Generator().manual_seed(24)).to(DEV)

    def step():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return model(images)

    eager = step().float().clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    worst = 0.0
    for _ in range(6):   # back to back: every replay after the first finds the previous one's bytes in its buffers
        g.replay()
    torch.cuda.synchronize()
    worst = float((out.float() - eager).abs().max())
    images.copy_(torch.randn(16, 3, 224, 224, generator=torch.Generator().manual_seed(25)).to(DEV))
    g.replay()
    torch.cuda.synchronize()
    other = float((out.float() - step().float()).abs().max())
    q.put({"replay_vs_eager": worst, "new_input_vs_eager": other})


def test_eval_forward_captured_in_a_hip_graph_replays_bit_exact():
    """The first capture of this forward faulted on its SECOND replay: the router's redo counter was cleared with
    hipMemsetAsync, whose graph node did not order against the kernels around it, so the f32 pass appended to its redo
    list at whatever the counter's bytes held from the previous replay.  The counters are now cleared by a kernel
    (smoe_zero_words) and every list append is bounds-checked (list_push).  Child process, finite timeout."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_graph_worker, args=(q,))
    p.start()
    _join_or_kill([p], 240)
    res = q.get(timeout=10)
    print("graph replay:", res)
    assert res["replay_vs_eager"] == 0.0 and res["new_input_vs_eager"] == 0.0, res


def test_baseline_shapes_never_leave_the_own_kernels_and_other_shapes_say_so():
    """vit.py's dense pieces fall back to torch / vendor kernels for shapes the library does not cover -- loudly
    (SlimMoEFallbackWarning, once per piece / reason / shape).  BASELINE cfg 2 (ViT-B/16 @224, 1000 classes) and cfg 4
    (ViT-L/16 @384) raise none; a class count that is not a multiple of 8 runs on the padded-N GEMM (no fallback, same
    numbers as F.linear); a sequence beyond the attention kernel's reach (N = 785 > 640) warns, once."""
    import warnings
    from slim_switch_moe_vit_amd import vit
    vit._fallbacks_seen.clear()
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("error", vit.SlimMoEFallbackWarning)
        for name, size, kw in (("moe_base_patch16_224_expert8_top1", 224, dict(num_classes=1000)),
                               ("moe_large_patch16_384_expert32_top1", 384, dict(num_classes=1000)),
                               ("moe_tiny_patch16_224_expert4_top1", 224, dict(num_classes=10))):
            model = _init(sm.create_model(name, depth=1, **kw), 41).eval().to(DEV)
            images = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(42)).to(DEV)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                out = model(images)
                feats = model.forward_features(images)
            ref = torch.nn.functional.linear(feats.half(), model.head.weight.half(), model.head.bias.half())
            assert out.shape == (2, kw["num_classes"])
            assert (out.float() - ref.float()).abs().max().item() <= 2e-3 * max(1.0, float(ref.abs().max()))
    model = _init(sm.create_model("moe_tiny_patch16_224_expert4_top1", depth=1, num_classes=16, img_size=448), 43).eval().to(DEV)
    images = torch.randn(1, 3, 448, 448, generator=torch.Generator().manual_seed(44)).to(DEV)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always", vit.SlimMoEFallbackWarning)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            model(images)
            model(images)
    hits = [w for w in rec if issubclass(w.category, vit.SlimMoEFallbackWarning)]
    assert len(hits) == 1 and "attention" in str(hits[0].message), [str(w.message) for w in rec]


def _static_exchange_worker(rank, world, port, q):
    """One rank of the capacity gate's static exchange: SwitchGate (capacity factor 1.0 -> drops), W ranks sharing cuda:0
    over gloo; the operator with this rank's expert slice must reproduce the single-rank operator holding all experts on
    this rank's tokens (capacity is per source rank, so the kept set is the same), with no host sync in the layer."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d, h, E = 192, 768, 8
        E_local = E // world
        g = torch.Generator().manual_seed(77)
        wg = torch.randn(E, d, generator=g) * 0.3
        bg = torch.zeros(E); bg[1] = 1.0                      # skew: expert 1 overflows its capacity
        w1 = torch.randn(E, h, d, generator=g) * 0.02; b1 = torch.randn(E, h, generator=g) * 0.02
        w2 = torch.randn(E, d, h, generator=g) * 0.02; b2 = torch.randn(E, d, generator=g) * 0.02
        T = 937

        def build(ws):
            m = sm.FMoETransformerMLP(E // ws, d, h, torch.nn.GELU(), top_k=1, gate="switch", capacity_factor=1.0, world_size=ws)
            sl = slice(rank * E_local, (rank + 1) * E_local) if ws > 1 else slice(0, E)
            with torch.no_grad():
                m.gate.gate.weight.copy_(wg); m.gate.gate.bias.copy_(bg)
                m.experts.htoh4.weight.copy_(w1[sl]); m.experts.htoh4.bias.copy_(b1[sl])
                m.experts.h4toh.weight.copy_(w2[sl]); m.experts.h4toh.bias.copy_(b2[sl])
            return m.to(DEV).eval()

        full, part = build(1), build(world)
        from slim_switch_moe_vit_amd import ep
        calls = {"n": 0}
        real = ep.PendingCounts.finish

        def counting(self):
            calls["n"] += 1
            return real(self)
        ep.PendingCounts.finish = counting                     # the dynamic path's host sync: must never run here

        def run(T):
            x = torch.randn(T, d, generator=torch.Generator().manual_seed(500 + rank)).to(DEV)
            res = torch.randn(T, d, generator=torch.Generator().manual_seed(600 + rank)).to(DEV)
            with torch.no_grad():
                got = part.forward_add(x, res)
                ln = torch.nn.LayerNorm(d, eps=1e-6).to(DEV)
                got_ln = part.forward_norm_add(x, ln)
                if T == 0:                                     # a rank without rows: takes part in every collective, returns nothing
                    return float(got.numel() + got_ln.numel()), 0.0, True, True
                counts_part = part.last_plan[2].clone()
                ref = full.forward_add(x, res)
                ref_ln = full.forward_norm_add(x, ln)
            return float((got - ref).abs().max()), float((got_ln - ref_ln).abs().max()), int(full.last_plan[3][-1]) < T, \
                bool(torch.equal(counts_part, full.last_plan[2]))

        err, err_ln, dropped, same_counts = run(T + 13 * rank)  # ragged first batch: the slot size is agreed once (the largest)
        err_r, err_ln_r, _, same_r = run(T - 7 * rank)          # later, smaller batches use the same buffers, no communication
        err_0, err_ln_0, _, same_0 = run(0 if rank == world - 1 else T)      # one rank has NO rows this time
        ep.check_static_overflow(flush=True)                    # everything fitted: silent on every rank
        # a LARGER batch on one rank only: nobody raises alone in front of a collective (that was a deadlock); every rank raises
        # at the same check, and the module is re-sized so that repeating the step works
        big = T + 13 * world + 50
        run(big if rank == 0 else T)
        try:
            ep.check_static_overflow(flush=True)
            overflow = "not raised"
        except ep.StaticExchangeOverflow:
            overflow = "raised"
        err_b, err_ln_b, _, same_b = run(big if rank == 0 else T)            # the repeated step fits the re-sized buffers
        ep.check_static_overflow(flush=True)
        q.put((rank, max(err, err_r, err_0, err_b), max(err_ln, err_ln_r, err_ln_0, err_ln_b), calls["n"], int(dropped),
               same_counts and same_r and same_0 and same_b, overflow == "raised" and part.ep_static_tokens >= big))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_static_capacity_padded_exchange_ranks_on_one_gpu(world):
    """BASELINE cfg 5's layout (capacity gate under expert parallelism): [W, E_local, cap, d] exchange buffers with equal
    all-to-all splits, received counts kept on the device as the end of each slot's row range (smoe_grouped_gemm's
    group_end) -- the layer never waits for the host.  W processes on one GPU over gloo."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_static_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    _join_or_kill(procs, 120)
    got = sorted(q.get(timeout=10) for _ in range(world))
    for rank, err, err_ln, host_syncs, dropped, same_counts, too_big in got:
        print(f"rank {rank}/{world}: |ep - single| {err:.2e} (with LayerNorm fused {err_ln:.2e}), count read-backs {host_syncs}")
        assert host_syncs == 0, "the static exchange must not read the counts back"
        assert too_big, "a batch larger than the agreed slot size must raise"
        assert dropped == 1, "the test must exercise dropping"
        assert same_counts
        assert err <= 2e-3 and err_ln <= 2e-3, (rank, err, err_ln)


def test_eval_harness_on_hip_graphs_reproduces_the_eager_harness():
    """engine.evaluate(hip_graph=True): every batch shape's forward is captured once (GraphedForward) and replayed -- the same
    logits BIT FOR BIT, hence the same loss / accuracy, and the same token-skip counters (host-side ``_total_tokens`` replayed by
    hand, device-side skip counters by the captured kernels) as the eager harness, over two batches of one shape and a ragged last
    one.  The reference's live model (resmoe_tiny_patch16_224_expert8, gates firing) and the stock-block one."""
    for name, kw in (("resmoe_tiny_patch16_224_expert8", dict(starting_threshold=0.55, target_threshold=0.5)),
                     ("moe_tiny_patch16_224_expert8", {})):
        torch.manual_seed(0)
        model = _init(sm.create_model(name, num_classes=100, depth=3, **kw), 51)
        with torch.no_grad():
            for blk in model.blocks:
                for gt in (getattr(blk, "dense_gate", None), getattr(blk, "moe_gate", None)):
                    if gt is not None:
                        gt.head[1].weight.normal_(0, 0.3, generator=torch.Generator().manual_seed(5))
        model = model.to(DEV)
        g = torch.Generator().manual_seed(52)
        batches = [(torch.randn(b, 3, 224, 224, generator=g), torch.randint(0, 100, (b,), generator=g)) for b in (6, 6, 3)]
        gates = [m for m in model.modules() if isinstance(m, sm.Gate)]

        def run(graph):
            for gt in gates:
                gt._total_tokens, gt._skipped_tokens = 0, 0.0
            stats = sm.evaluate(batches, model, DEV, hip_graph=graph)
            return stats, [(gt._total_tokens, gt._skipped_tokens) for gt in gates]

        eager, c_eager = run(False)
        graphed, c_graph = run(True)
        again, c_again = run(True)            # graphs are per evaluate() call: captured afresh, same numbers
        assert graphed["hip_graph"] and not eager["hip_graph"]
        for key in ("loss", "acc1", "acc5"):
            assert graphed[key] == eager[key] == again[key], (name, key, graphed[key], eager[key])
        assert c_graph == c_eager == c_again, (name, c_graph, c_eager)
        if gates:
            assert sum(s for _, s in c_eager) > 0, "the gates must fire in this test"
