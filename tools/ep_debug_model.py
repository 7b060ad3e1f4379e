"""debug: block-by-block comparison of the expert-parallel ViT against the single-rank model, W processes on one GPU."""
import os, sys, socket
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, world, port, q):
    import torch.distributed as dist
    import slim_switch_moe_vit_amd as sm
    import test_gpu_model as t
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E, E_local = 4, 4 // world
    torch.manual_seed(0)
    full = t._init(sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=50), 11).eval()
    torch.manual_seed(0)
    part = sm.create_model("moe_tiny_patch16_224_expert4_top1", num_classes=50, world_size=world).eval()
    sd = full.state_dict()
    sl = slice(rank * E_local, (rank + 1) * E_local)
    for k in list(sd):
        if ".experts." in k:
            sd[k] = sd[k][sl].clone()
    part.load_state_dict(sd)
    full, part = full.cuda(), part.cuda()
    images = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(100 + rank)).cuda()
    out = []
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        xf = full._embed(images); xp = part._embed(images)
        for i in range(len(full.blocks)):
            xf = full.blocks[i](xf); xp = part.blocks[i](xp)
            cf = full.blocks[i].mlp.last_plan[2].tolist(); cp = part.blocks[i].mlp.last_plan[2].tolist()
            out.append((i, round(float((xf - xp).abs().max()), 5), cf, cp))
    q.put((rank, out))
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]; [p.join(280) for p in ps]
    for _ in range(world):
        r, out = q.get(timeout=5)
        for o in out:
            print("RES", r, o)
