"""eval-forward timing of BASELINE cfg 4's model (ViT-L/16 @384, E = 32, top-1; models/vision_transformer.py:1227-1236 dims) on one
GPU, batch from argv (default 64 = BASELINE's 512 images over 8 ranks), with a per-kernel table from HIP events (launches per step,
average us, TFLOP/s or TB/s).  usage: cfg4_bench.py [batch=64] [steps=5]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slim_switch_moe_vit_amd as sm
from slim_switch_moe_vit_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
m = sm.create_model("moe_large_patch16_384_expert32_top1", num_classes=1000).eval()
g = torch.Generator().manual_seed(1)
with torch.no_grad():
    for blk in m.blocks:
        blk.mlp.gate.gate.weight.copy_(torch.randn(blk.mlp.gate.gate.weight.shape, generator=g) * 0.02)
        for lin in (blk.mlp.experts.htoh4, blk.mlp.experts.h4toh):
            lin.weight.normal_(0, 0.02, generator=g).clamp_(-0.04, 0.04)
m = m.cuda()
x = torch.randn(B, 3, 384, 384, generator=g).cuda()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        m(x); ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    ops.profile_begin()
    for _ in range(2): m(x)
    torch.cuda.synchronize()
prof = ops.profile_end()
agg = {}
for n, meta, t in prof:
    key = n
    if n == "grouped_gemm":
        key = f"expert GEMM K={meta['K']} N={meta['N']}" + (" +GELU" if meta.get("epilogue") == ops.EPI_GELU else "")
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += t; a[2] += meta.get("flops", 0.0); a[3] += meta.get("bytes", 0.0)
med = ms[len(ms) // 2]
print(f"cfg4 (ViT-L/16 @384, E 32, top-1) batch {B}: {med:.2f} ms/step (min {ms[0]:.2f}, max {ms[-1]:.2f}) = {B / med * 1e3:.0f} images/s; "
      f"timed kernels {sum(a[1] for a in agg.values()) / 2:.2f} ms/step")
for n, (c, t, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    extra = (f"{fl / (t * 1e-3) / 1e12:7.1f} TF/s ({fl / (t * 1e-3) / 1e12 / 2500:.3f} of 2.5 PF)" if fl else "") + \
            (f"{by / (t * 1e-3) / 1e12:6.2f} TB/s ({by / (t * 1e-3) / 1e12 / 6.3:.3f} of 6.3)" if by else "")
    print(f"  {n:34s} x{c / 2:5.1f}  {1e3 * t / c:8.1f} us  {t / 2:8.3f} ms/step  {extra}")
