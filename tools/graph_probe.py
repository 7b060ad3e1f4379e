#!/usr/bin/env python3
"""Does capturing the eval forward in a HIP graph buy anything?  Eager vs graph replay of the bench model (same kernels).
usage: graph_probe.py [batch] [iters]"""
import os, sys, time, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sys.argv = [sys.argv[0], "--batch", str(batch)]
args = bench.parse()
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
model, _ = bench.build_model(args, 1, 0, device)
images = torch.randn(batch, 3, 224, 224, generator=torch.Generator().manual_seed(100)).to(device)

def step():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        return model(images)

def timed(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

ref = step().float().clone()
t_eager = timed(step)
print(f"eager: {t_eager:.3f} ms/step", flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
for i in range(4):
    g.replay(); torch.cuda.synchronize()
    print(f"graph replay {i} == eager:", bool(torch.equal(out.float(), ref)), flush=True)
t_graph = timed(g.replay)
print(f"graph: {t_graph:.3f} ms/step  ({(t_eager / t_graph - 1) * 100:+.1f} %)", flush=True)
