"""Host-side enqueue time vs GPU time of one bench step (diagnostic): how close the step is to being launch-bound.
usage: python tools/host_overhead.py [--force-ep] [--ep-micro-batches N] [--batch B]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--force-ep", action="store_true")
    ap.add_argument("--ep-micro-batches", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    args = argparse.Namespace(experts=8, compute_dtype="f16", gemm_variant=4, ep_chunks=1, force_ep=a.force_ep,
                              no_cpu_baseline=True, ep_micro_batches=a.ep_micro_batches)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if a.force_ep:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=dev)
    model, _ = bench.build_model(args, 1, 0, dev)
    model = model.to(dev).eval()
    x = torch.randn(a.batch, 3, 224, 224, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        enq, tot = [], []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model(x)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            enq.append((t1 - t0) * 1e3)
            tot.append((t2 - t0) * 1e3)
        print(f"force_ep={a.force_ep} micro={a.ep_micro_batches}: host enqueue {min(enq):.2f} ms, step {min(tot):.2f} ms")
        if a.profile:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(3):
                model(x)
            pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("tottime").print_stats(60)
    if a.force_ep:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
