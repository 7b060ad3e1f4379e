"""eval-forward timing of BASELINE cfg 4's model (ViT-L/16 @384, E = 32, top-1) on one GPU, batch from argv (default 32)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slim_switch_moe_vit_amd as sm
from slim_switch_moe_vit_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = sm.create_model("moe_large_patch16_384_expert32_top1", num_classes=1000).cuda().eval()
x = torch.randn(B, 3, 384, 384, device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    for _ in range(2): m(x)
    torch.cuda.synchronize()
    ops.profile_begin()
    t0 = time.perf_counter()
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
prof = ops.profile_end()
agg = {}
for n, meta, ms in prof:
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += ms
print(f"cfg4 batch {B}: {dt*1e3:.1f} ms/step  {B/dt:.0f} img/s")
for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:20s} {c/3:6.1f}/step  {ms/3:8.3f} ms/step")
