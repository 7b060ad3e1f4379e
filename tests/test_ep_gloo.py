"""Expert-parallel exchange on CPU with the gloo backend (world_size 2 and 4): the product's data-movement
code (slim_switch_moe_vit_amd.ep: exchange_counts / all_to_all_rows / segment_table / chunk_bounds) is driven
exactly as ep_forward drives it, with the oracle standing in for the HIP compute kernels, and the result must
equal the single-rank oracle over all W*E_local experts."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _mp import join_or_kill

from oracle import moe_oracle as mo


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(W, E_local, d, h, T_per_rank, k):
    g = torch.Generator().manual_seed(100 + W)
    E = W * E_local
    xs = [torch.randn(T_per_rank + 3 * w, d, generator=g) for w in range(W)]  # ragged: ranks hold different T
    wg = torch.randn(E, d, generator=g) * 0.3
    bg = torch.randn(E, generator=g) * 0.1
    w1 = torch.randn(E, h, d, generator=g) * 0.1
    b1 = torch.randn(E, h, generator=g) * 0.1
    w2 = torch.randn(E, d, h, generator=g) * 0.1
    b2 = torch.randn(E, d, generator=g) * 0.1
    return xs, wg, bg, w1, b1, w2, b2


def _worker(rank, W, port, E_local, k, n_chunks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        from slim_switch_moe_vit_amd import ep

        d, h, T0 = 16, 32, 41
        xs, wg, bg, w1, b1, w2, b2 = _inputs(W, E_local, d, h, T0, k)
        x = xs[rank]
        T = x.shape[0]
        E = W * E_local
        idx, score, _ = mo.naive_gate(x, wg, bg, k)
        bounds = ep.chunk_bounds(T, n_chunks)
        plans = [mo.dispatch_plan(idx[t0:t1].numpy(), E) for t0, t1 in bounds]
        lec, gec = ep.exchange_counts([torch.from_numpy(p.counts) for p in plans], W)
        # local experts of this rank
        lw1, lb1 = w1[rank * E_local:(rank + 1) * E_local], b1[rank * E_local:(rank + 1) * E_local]
        lw2, lb2 = w2[rank * E_local:(rank + 1) * E_local], b2[rank * E_local:(rank + 1) * E_local]
        out = torch.zeros(T, d)
        inflight = []
        for c, (t0, t1) in enumerate(bounds):
            p = plans[c]
            kept = int(p.offsets[E])
            assert np.array_equal(lec[c].numpy().reshape(-1), p.counts)
            send = x[t0:t1][torch.from_numpy(p.pos[:kept]) // k]
            recv, work = ep.all_to_all_rows(send, lec[c].sum(1).tolist(), gec[c].sum(1).tolist(), async_op=True)
            inflight.append((recv, work))
        returning = []
        for c in range(len(bounds)):
            recv, work = inflight[c]
            work.wait()
            offs, gexp = ep.segment_table(gec[c])
            y = torch.zeros(recv.shape[0], d)
            for gi, e in enumerate(gexp):  # what smoe_grouped_gemm does with group_expert
                lo, hi = offs[gi], offs[gi + 1]
                if hi > lo:
                    y[lo:hi] = mo.gelu_erf(recv[lo:hi] @ lw1[e].t() + lb1[e]) @ lw2[e].t() + lb2[e]
            back, work2 = ep.all_to_all_rows(y, gec[c].sum(1).tolist(), lec[c].sum(1).tolist(), async_op=True)
            returning.append((back, work2))
        for c, (t0, t1) in enumerate(bounds):
            back, work2 = returning[c]
            work2.wait()
            p = plans[c]
            kept = int(p.offsets[E])
            Z = torch.zeros((t1 - t0) * k, d)
            Z[torch.from_numpy(p.pos[:kept])] = back
            out[t0:t1] = torch.bmm(score[t0:t1].view(-1, 1, k), Z.view(-1, k, d)).reshape(-1, d)
        ref = mo.moe_forward(x, wg, bg, w1, b1, w2, b2, k).out
        q.put((rank, float((out - ref).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("W,E_local,k,n_chunks", [(2, 2, 1, 1), (2, 2, 2, 2), (2, 1, 1, 3), (4, 2, 1, 2)])
def test_ep_exchange_matches_single_rank_oracle(W, E_local, k, n_chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, W, port, E_local, k, n_chunks, q)) for r in range(W)]
    for p in procs:
        p.start()
    join_or_kill(procs, 120)
    got = sorted(q.get(timeout=5) for _ in range(W))
    assert [r for r, _ in got] == list(range(W))
    for _, err in got:
        assert err < 1e-5, err


def test_segment_table_and_chunk_bounds():
    from slim_switch_moe_vit_amd import ep

    offs, gexp = ep.segment_table(torch.tensor([[3, 0], [1, 2]]))
    assert offs == [0, 3, 3, 4, 6] and gexp == [0, 1, 0, 1]
    assert ep.chunk_bounds(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert ep.chunk_bounds(2, 4) == [(0, 1), (1, 2)]
    assert ep.chunk_bounds(0, 2) == [(0, 0)]


def _a2a_grad_worker(rank, W, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        from slim_switch_moe_vit_amd.autograd import _AllToAll

        g = torch.Generator().manual_seed(7)
        counts = torch.randint(0, 5, (W, W), generator=g)           # counts[src, dst] rows
        send = counts[rank].tolist()
        recv = counts[:, rank].tolist()
        rows = torch.arange(sum(send) * 3, dtype=torch.float32).reshape(-1, 3) + 100 * rank
        rows.requires_grad_(True)
        out = _AllToAll.apply(rows, send, recv, None)
        w = torch.arange(out.numel(), dtype=torch.float32).reshape(out.shape) + 7 * rank  # d(out): known per receiver
        (out * w).sum().backward()
        # the gradient of a row is the weight it met at its destination: send it back with a plain exchange
        from slim_switch_moe_vit_amd.ep import all_to_all_rows
        back, _ = all_to_all_rows(w, recv, send)
        q.put((rank, bool(torch.equal(rows.grad, back)), out.shape[0] == sum(recv)))
    finally:
        dist.destroy_process_group()


def test_all_to_all_autograd_adjoint_is_the_reverse_exchange():
    W = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_a2a_grad_worker, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    join_or_kill(procs, 120)
    got = sorted(q.get(timeout=5) for _ in range(W))
    assert all(ok and shp for _, ok, shp in got)


class _StubGate:
    tot_expert = 6

    def capacity(self, n):
        return -(-n // 6)            # ceil(1.0 * n * 1 / E)


class _StubMoE:
    """The attributes ep's control plane reads from an FMoETransformerMLP (no kernels involved)."""

    def __init__(self, W):
        self.world_size, self.moe_group, self.top_k, self.num_expert = W, None, 1, 6 // W
        self.gate, self.gemm_variant, self.d_model, self.d_hidden = _StubGate(), 9, 64, 128
        self._fused_gelu, self._drop_p, self.training = True, 0.0, False


def _static_control_worker(rank, W, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        from slim_switch_moe_vit_amd import ep
        res = {}
        mod = _StubMoE(W)
        res["static_by_config"] = ep.use_static_exchange(mod, torch.float16)           # no batch size in the decision
        rows = [5, 0, 9][rank]                                                          # unequal batches, one rank without rows
        agreed = ep.static_slot_tokens(mod, rows, "cpu")
        res["agreed"] = agreed
        res["agreed_again"] = ep.static_slot_tokens(mod, 1, "cpu")                      # later calls: no collective, same value
        counts = torch.arange(6, dtype=torch.int32) + 10 * rank
        recv, peer = ep.exchange_counts_static(counts, rows, W)
        E_local = 6 // W
        want = torch.cat([torch.arange(6, dtype=torch.int32)[rank * E_local:(rank + 1) * E_local] + 10 * w for w in range(W)])
        res["counts_ok"] = bool(torch.equal(recv, want)) and peer.tolist() == [5, 0, 9]
        # what the headers of an exchange deliver on every rank: [W, 1 + E] = (rows, pre-clamp count per global expert) per source
        stats = lambda rows_v: torch.cat([rows_v.to(torch.int32).reshape(W, 1), torch.zeros((W, 6), dtype=torch.int32)], 1)
        st = ep._slot_state(mod, "capacity", agreed, "cpu")
        res["slots"] = (st.table.caps, st.table.rows, st.table.in_splits, st.table.out_splits)
        ep._watch_overflow(stats(peer), st)
        ep.check_static_overflow(flush=True)                                            # everything fits: silent
        # a later, LARGER batch on one rank: nobody raises alone; every rank raises at the same check, after re-sizing
        rows2 = [20, 0, 9][rank]
        _, peer2 = ep.exchange_counts_static(counts, rows2, W)
        ep._watch_overflow(stats(peer2), st)
        ep.check_static_overflow()                                                      # younger than the lag: not read yet
        try:
            ep.check_static_overflow(flush=True)
            res["overflow"] = "not raised"
        except ep.StaticExchangeOverflow as exc:
            res["overflow"] = "raised" if "[20, 0, 9]" in str(exc) else str(exc)
        res["resized"] = mod.ep_static_tokens
        res["slots_after"] = ep._slot_state(mod, "capacity", ep.static_slot_tokens(mod, 1, "cpu"), "cpu").table.caps
        # presets that differ between ranks: the same error everywhere
        mod2 = _StubMoE(W)
        if rank == 0:
            mod2.ep_static_tokens = 10
        try:
            ep.static_slot_tokens(mod2, 4, "cpu")
            res["preset_mismatch"] = "not raised"
        except RuntimeError:
            res["preset_mismatch"] = "raised"
        # the same preset everywhere wins over the row counts
        mod3 = _StubMoE(W)
        mod3.ep_static_tokens = 64
        res["preset"] = ep.static_slot_tokens(mod3, rows, "cpu")
        # micro-batches: the agreement is per divisor; a preset counts rows of the whole batch, cut between images (`ep_rows_unit`)
        mod3.ep_rows_div, mod3.ep_rows_unit = 3, 4
        res["preset_div3"] = ep.static_slot_tokens(mod3, 1, "cpu")                      # ceil(64 / (3 * 4)) * 4 = 24
        mod4 = _StubMoE(W)
        mod4.ep_rows_div = 2
        res["agreed_div2"] = ep.static_slot_tokens(mod4, [7, 3, 11][rank], "cpu")       # no preset: the largest micro-batch
        mod4.ep_rows_div = 1
        res["agreed_div1"] = ep.static_slot_tokens(mod4, [14, 6, 22][rank], "cpu")      # its own collective, its own entry
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_static_exchange_agreement_and_overflow_are_collective():
    """ADVICE r3 (medium x2) / VERDICT r3 weak #5: which exchange a module uses depends on its configuration only; the slot size
    is agreed by a collective every rank runs; a rank with no rows takes part; a later larger batch is reported by ALL ranks at the
    same call (no rank-local raise in front of a collective), after re-sizing; presets that differ raise everywhere."""
    W = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_static_control_worker, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    join_or_kill(procs, 120)
    got = dict(q.get(timeout=5) for _ in range(W))
    for r in range(W):
        res = got[r]
        # capacity(9) = 2 rows per (source, expert) slot + one header row each: 6 x 3 = 18 rows, 2 experts (6 rows) per peer
        assert res.pop("slots") == ([2] * 6, 18, [6, 6, 6], [6, 6, 6]), (r, res)
        assert res.pop("slots_after") == [4] * 6, (r, res)                              # capacity(20) after the re-size
        assert res == {"static_by_config": True, "agreed": 9, "agreed_again": 9, "counts_ok": True, "overflow": "raised",
                       "resized": 20, "preset_mismatch": "raised", "preset": 64, "preset_div3": 24, "agreed_div2": 11,
                       "agreed_div1": 22}, (r, res)


class _NaiveStubGate:
    tot_expert = 6

    def capacity(self, n):
        return -1


def _speculative_control_worker(rank, W, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    try:
        from slim_switch_moe_vit_amd import ep
        ep.SIGMA = 0.0                                       # (the toy counts below test the proportional rule: HEADROOM x the largest group)
        res = {}
        mods = []
        for _ in range(3):                                   # three "layers"
            m = _StubMoE(W)
            m.gate = _NaiveStubGate()
            mods.append(m)
        # off by default: a gate without a capacity exchanges dynamically unless the harness opted in (somebody must own the redo)
        res["default_kind"] = ep.static_kind(mods[0], torch.float16)
        for m in mods:
            m.ep_speculative = 1.5
        res["kind"] = ep.static_kind(mods[0], torch.float16)
        with ep.dynamic_only():
            res["kind_in_redo"] = ep.static_kind(mods[0], torch.float16)
        agreed = [ep.static_slot_tokens(m, [40, 0, 24][rank], "cpu") for m in mods]
        res["agreed"] = agreed
        sts = [ep._slot_state(m, "speculative", a, "cpu") for m, a in zip(mods, agreed)]
        res["first_caps"] = sts[0].table.caps                # uniform: ceil(1.5 * 40 * 1 / 6) = 10
        # [W, 1 + E]: source w brought rows[w] rows and routed hist[w][e] of them to global expert e (before any clamp)
        rows = [40, 0, 24]
        mk = lambda hist: torch.tensor([[rows[w]] + hist[w] for w in range(W)], dtype=torch.int32)
        zero = [0] * 6
        h0 = [[13, 5, 6, 6, 5, 5], zero, [4, 4, 4, 4, 4, 4]]           # layer 0: source 0 overflows expert 0 (13 > 10)
        h1 = [[10, 6, 6, 6, 6, 6], zero, [4, 4, 4, 4, 4, 4]]           # layer 1: fits
        h2 = [[7, 7, 7, 7, 6, 6], zero, [2, 2, 2, 2, 17, 1]]           # layer 2: source 2 overflows expert 4 (17 > 10)
        calls = {"n": 0, "dyn": 0}

        def step():
            calls["n"] += 1
            if ep.static_kind(mods[0], torch.float16) is None:
                calls["dyn"] += 1
                return "dynamic"
            for st, h in zip(sts, (h0, h1, h2)):
                ep._watch_overflow(mk(h), st)
            return "static"
        out, again = ep.run_guarded(step)
        res["redo"] = (out, again, calls["n"], calls["dyn"])
        # EVERY overflowing layer was re-sized by the one raise (per expert: max(old, ceil(1.12 x the largest group seen))), the
        # fitting one was not
        res["caps"] = [st.table.caps for st in sts]
        res["pending_after"] = len(ep._overflow_pending)
        # steps 2 and 3: the same routing fits the re-sized slots: no repeat; after ADAPT_MIN_OBS observations without an overflow
        # every layer's slots are cut to ceil(1.12 x its largest group) PER EXPERT (the all-to-all then carries ~1.12 x the routed rows)
        res["second"] = ep.run_guarded(step)
        res["third"] = ep.run_guarded(step)
        res["fitted"] = [st.table.caps for st in sts]
        res["splits"] = (sts[1].table.in_splits, sts[1].table.out_splits, sts[1].table.rows, sts[1].table.recv_rows)
        # micro-batch siblings share a slot state: the deferred check in the MIDDLE of a forward has read some micro-batches' matrices
        # only and must not cut the slots to them (bench.py's grid at E = 16 with three micro-batches did: the sibling overflowed)
        mods[1].ep_speculative = 3.0
        st = ep._slot_state(mods[1], "speculative", agreed[1], "cpu")          # re-made: ceil(3.0 * 40 / 6) = 20 per expert
        res["roomy_caps"] = list(st.table.caps)
        hA = [[5, 5, 5, 5, 5, 5], zero, [4, 4, 4, 4, 4, 4]]
        hB = [[5, 5, 5, 5, 5, 18], zero, [4, 4, 4, 4, 4, 4]]                   # the sibling's routing: fits 20, not ceil(1.12 x 5)
        lag, ep.OVERFLOW_LAG = ep.OVERFLOW_LAG, 2
        try:
            for h in (hA, hA, hA, hB):
                ep._watch_overflow(mk(h), st)
            ep.check_static_overflow()                                         # reads the two oldest (both hA): enough "observations"
            res["caps_mid_forward"] = list(st.table.caps)
            ep.check_static_overflow(flush=True)                               # the step boundary: everything has reported
            res["caps_at_boundary"] = list(st.table.caps)
        finally:
            ep.OVERFLOW_LAG = lag
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_speculative_exchange_control_plane_resizes_every_layer_and_repeats_once():
    """VERDICT r4 item 1 / ADVICE r4 (low): the speculative static exchange of a capacity-less gate is opt-in; which exchange runs is
    a function of shared switches only (incl. the redo context); an overflow of SEVERAL layers in one step re-sizes every one of
    them in the one raise (a 12-layer model must not need 12 repeats), the step is repeated ONCE, on the dynamic exchange, by all
    ranks together, and the next step fits."""
    W = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_speculative_control_worker, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    join_or_kill(procs, 120)
    got = dict(q.get(timeout=5) for _ in range(W))
    import math
    fit = lambda hist: [max(1, math.ceil(1.12 * max(col))) for col in zip(*hist)]
    for r in range(W):
        res = got[r]
        assert res["default_kind"] is None and res["kind"] == "speculative" and res["kind_in_redo"] is None, (r, res)
        assert res["agreed"] == [40, 40, 40] and res["first_caps"] == [10] * 6, (r, res)
        assert res["redo"] == ("dynamic", True, 2, 1), (r, res)
        assert res["caps"] == [[15, 10, 10, 10, 10, 10], [10] * 6, [10, 10, 10, 10, 20, 10]], (r, res)
        assert res["pending_after"] == 0 and res["second"] == ("static", False) and res["third"] == ("static", False), (r, res)
        assert res["fitted"] == [fit([[13, 5, 6, 6, 5, 5], [4] * 6]), fit([[10, 6, 6, 6, 6, 6], [4] * 6]),
                                 fit([[7, 7, 7, 7, 6, 6], [2, 2, 2, 2, 17, 1]])], (r, res)
        # layer 1 after the cut: caps [12, 7, 7, 7, 7, 7] + one header row each; two experts per rank
        E_local, caps = 2, res["fitted"][1]
        blocks = [sum(caps[w * E_local:(w + 1) * E_local]) + E_local for w in range(W)]
        assert res["splits"] == (blocks, [blocks[r]] * W, sum(blocks), W * blocks[r]), (r, res)
        assert res["roomy_caps"] == [20] * 6 and res["caps_mid_forward"] == [20] * 6, (r, res)
        assert res["caps_at_boundary"] == fit([[5, 5, 5, 5, 5, 18], [4] * 6]), (r, res)
