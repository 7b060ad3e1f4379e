#!/usr/bin/env python3
"""Board power / shader clock while a grouped-GEMM loop runs (is the expert GEMM power-limited?).  Starts tools/gemm_prof.py as a
child process and samples `rocm-smi --showpower --showclocks --json` beside it.
usage: power_probe.py [variant] [shape] [iters] [data: randn|zeros|uniform]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = sys.argv[1] if len(sys.argv) > 1 else "9"
shape = sys.argv[2] if len(sys.argv) > 2 else "fc2"
iters = sys.argv[3] if len(sys.argv) > 3 else "20000"
data = sys.argv[4] if len(sys.argv) > 4 else "randn"


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True,
                             timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(k for k in d if k.startswith("card"))[0]]
        return card
    except Exception as exc:  # noqa: BLE001
        return {"error": str(exc)}


print("idle:", {k: v for k, v in sample().items() if "ower" in k or "sclk" in k}, flush=True)
env = dict(os.environ, SMOE_DATA=data)
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gemm_prof.py"), variant, shape, iters], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
time.sleep(7.0)   # import + warm-up
rows = []
while child.poll() is None and len(rows) < 60:
    rows.append(sample())
    time.sleep(0.25)
out = child.communicate()[0]
print(out.strip().splitlines()[-1] if out.strip() else "(no output)")
pw = []
ck = []
for r in rows:
    for k, v in r.items():
        if "Average Graphics Package Power" in k or "Current Socket Graphics Package Power" in k:
            try:
                pw.append(float(v))
            except ValueError:
                pass
        if k.startswith("sclk clock level"):
            ck.append(str(v))
print(f"data {data}: {len(rows)} samples under load; package power W min/avg/max = "
      f"{min(pw) if pw else None} / {sum(pw) / len(pw) if pw else None} / {max(pw) if pw else None}; sclk levels seen: {sorted(set(ck))[:6]}")
print("last sample keys:", {k: v for k, v in (rows[-1] if rows else {}).items() if "ower" in k or "sclk" in k})
