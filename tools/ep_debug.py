"""debug: W-rank expert-parallel forward of ONE MoE operator as W processes on one GPU (gloo)."""
import os, sys, socket
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, q):
    import torch.distributed as dist
    import slim_switch_moe_vit_amd as sm
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, h, E = 192, 768, 4
    E_local = E // world
    g = torch.Generator().manual_seed(5)
    wg = torch.randn(E, d, generator=g) * 0.1; bg = torch.zeros(E)
    w1 = torch.randn(E, h, d, generator=g) * 0.05; b1 = torch.randn(E, h, generator=g) * 0.05
    w2 = torch.randn(E, d, h, generator=g) * 0.05; b2 = torch.randn(E, d, generator=g) * 0.05
    T = [300, 257, 410, 129][rank]
    x = torch.randn(T, d, generator=torch.Generator().manual_seed(50 + rank)).cuda()

    def load(mod, sl):
        with torch.no_grad():
            mod.gate.gate.weight.copy_(wg); mod.gate.gate.bias.copy_(bg)
            mod.experts.htoh4.weight.copy_(w1[sl]); mod.experts.htoh4.bias.copy_(b1[sl])
            mod.experts.h4toh.weight.copy_(w2[sl]); mod.experts.h4toh.bias.copy_(b2[sl])
        return mod.cuda().eval()
    full = load(sm.FMoETransformerMLP(E, d, h, torch.nn.GELU(), top_k=1, compute_dtype=torch.float16), slice(0, E))
    part = load(sm.FMoETransformerMLP(E_local, d, h, torch.nn.GELU(), top_k=1, world_size=world, compute_dtype=torch.float16),
                slice(rank * E_local, (rank + 1) * E_local))
    ln = torch.nn.LayerNorm(d, eps=1e-6).cuda()
    with torch.no_grad():
        ref = full(x); got = part(x)
        ref2 = full.forward_norm_add(x, ln); got2 = part.forward_norm_add(x, ln)
    counts = part.last_plan[2].tolist()
    q.put((rank, float((got - ref).abs().max()), float((got2 - ref2).abs().max()), counts))
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]; [p.join(200) for p in ps]
    for _ in range(world):
        print(q.get(timeout=5))
