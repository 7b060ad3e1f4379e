#!/usr/bin/env python3
"""LayerNorm alone (f32 rows -> f16), the 16-lanes-per-token layout against one wave per row, per width: child processes (the
layout switch SMOE_LN_WAVE is read once per process), alternating, HIP events over 50 launches each.
usage: python tools/ln_ab.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, torch
sys.path.insert(0, %r)
from slim_switch_moe_vit_amd import ops
out = {}
for T, d in ((50432, 768), (36928, 1024), (25216, 192), (50432, 384)):
    x = torch.randn(T, d, device="cuda"); g = torch.randn(d, device="cuda"); b = torch.randn(d, device="cuda")
    big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    for _ in range(5): ops.layernorm(x, g, b, 1e-6, torch.float16)
    ts = []
    for _ in range(30):
        big.zero_()                      # push x out of the caches: the model's LayerNorm reads rows written a GEMM ago
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); y = ops.layernorm(x, g, b, 1e-6, torch.float16); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    ref = torch.nn.functional.layer_norm(x.double(), (d,), g.double(), b.double(), 1e-6)
    out["%%dx%%d" %% (T, d)] = {"us": round(1e3 * ts[len(ts) // 2], 1), "tb_s": round(T * d * 6 / ts[len(ts) // 2] / 1e9, 2),
                               "max_err": float((y.double() - ref).abs().max())}
print(json.dumps(out))
''' % ROOT
res = {}
for rnd in range(2):
    for mode in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SMOE_LN_WAVE=mode), capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:]); sys.exit(1)
        res.setdefault("wave_per_row" if mode == "1" else "16_lanes_per_token", []).append(json.loads(r.stdout.strip().splitlines()[-1]))
print(json.dumps(res, indent=1))
