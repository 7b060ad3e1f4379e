#!/usr/bin/env python3
"""Copies / condenses what `tools/diag.sh profile TAG` left under gpurun_out/prof_TAG into profiles/ (tracked): the rocprofv3 kernel
stats of bench / train steps / dispatch, the per-dispatch averages of every --pmc pass, and a markdown digest.
usage: tools/summarise_profiles.py r04"""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(os.path.join(dst, f"{tag}_pmc"), exist_ok=True)


def own(name: str) -> bool:
    return not ("at::native" in name or name.startswith("void at::") or name.startswith("Cijk") or "rocprim" in name or "hipcub" in name
                or "Memcpy" in name or "Memset" in name or "elementwise_kernel" in name and "at::" in name)


for what, out in (("bench/bench_kernel_stats.csv", f"{tag}_bench_kernel_stats.csv"), ("train/train_kernel_stats.csv", f"{tag}_train_model_kernel_stats.csv"),
                  ("train_resmoe/train_kernel_stats.csv", f"{tag}_train_resmoe_kernel_stats.csv"),
                  ("dispatch/dispatch_kernel_stats.csv", f"{tag}_dispatch_kernel_stats.csv"),
                  ("tiny_resmoe_tiny_patch16_224_expert8/tiny_kernel_stats.csv", f"{tag}_tiny_resmoe_eval_kernel_stats.csv"),
                  ("tiny_moe_tiny_patch16_224_expert8/tiny_kernel_stats.csv", f"{tag}_tiny_moe_eval_kernel_stats.csv"),
                  ("tiny_resmoe_tiny_patch16_224_expert8.json", f"{tag}_tiny_resmoe_eval.json"),
                  ("tiny_moe_tiny_patch16_224_expert8.json", f"{tag}_tiny_moe_eval.json"),
                  ("tiny_resmoe_tiny_patch16_224_expert8.txt", f"{tag}_tiny_resmoe_eval.txt"),
                  ("tiny_moe_tiny_patch16_224_expert8.txt", f"{tag}_tiny_moe_eval.txt"),
                  ("tiny_train/train_kernel_stats.csv", f"{tag}_tiny_resmoe_train_kernel_stats.csv"),
                  ("cfg4/cfg4_kernel_stats.csv", f"{tag}_cfg4_kernel_stats.csv"), ("cfg4_b64.txt", f"{tag}_cfg4_b64.txt"),
                  ("force_ep_line.json", f"{tag}_force_ep_line.json")):
    p = os.path.join(src, what)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, out))

# ---- PMC passes: per-kernel, per-counter averages per dispatch
digest = {}
for d in sorted(glob.glob(os.path.join(src, "*"))):
    base = os.path.basename(d)
    cc = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not cc:
        continue
    shutil.copy(cc[0], os.path.join(dst, f"{tag}_pmc", base + ".csv"))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if not any(s in k for s in ("grouped_gemm_ps", "expert_ffn", "attn_", "router16", "plan_", "scatter_rows", "layernorm")):
            continue
        for c, vs in cs.items():
            digest.setdefault(base, {}).setdefault(k, {})[c] = (sum(vs) / len(vs), len(vs))
json.dump({b: {k: {c: v[0] for c, v in cs.items()} for k, cs in ks.items()} for b, ks in digest.items()},
          open(os.path.join(dst, f"{tag}_pmc_digest.json"), "w"), indent=1)


def short(k):
    for s in ("expert_ffn_fused", "attn_fwd_kernel", "attn_bwd_kernel", "grouped_gemm_ps"):
        if s in k:
            return s + ("<" + k.split("grouped_gemm_psI")[1][:34] + ">" if s == "grouped_gemm_ps" and "grouped_gemm_psI" in k else "")
    return k[:60]


lines = [f"# {tag}: per-dispatch averages of the rocprofv3 --pmc passes (raw rows: profiles/{tag}_pmc/*.csv; recipe: tools/diag.sh profile {tag})", ""]
for b in sorted(digest):
    for k, cs in digest[b].items():
        lines.append(f"## {b} -- {short(k)}")
        for c, (v, n) in sorted(cs.items()):
            lines.append(f"    {c:34s} {v:18,.0f}   ({n} dispatches)")
        lines.append("")
open(os.path.join(dst, f"{tag}_pmc_digest.md"), "w").write("\n".join(lines))

# ---- training steps: own-kernel share and the top kernels
out = [f"# {tag}: training-step kernel tables (rocprofv3 --kernel-trace --stats -- python3 tools/train_bench.py model 128 6 [name]; 9 steps = 3 warm-up + 6 timed)", ""]
for what, title in (("train", "moe_base_patch16_224_expert8_top1 (SwitchGate, capacity_factor 1.0, aux loss: BASELINE cfg 5)"),
                    ("train_resmoe", "resmoe_base_patch16_224_expert8_top1 (the reference's live block: token-skip gates, residual on the normed activations, drop-path 0.1)"),
                    ("tiny_train", "resmoe_tiny_patch16_224_expert8 (the reference's OWN model: DeiT-Tiny, E 8, top-2, token-skip gates, batch 128 -- models/resMoE.py:151-187, cmd.sh:7-13)")):
    p = os.path.join(src, what, "train_kernel_stats.csv")
    if not os.path.exists(p):
        continue
    rows = list(csv.DictReader(open(p)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    mine = sum(float(r["TotalDurationNs"]) for r in rows if own(r["Name"]))
    log = open(os.path.join(src, what + ".log")).read().strip().splitlines()
    line = next((l for l in log if "train step" in l), "")
    plain = os.path.join(src, what + "_unprofiled.log")
    plain_line = " / ".join(l for l in open(plain).read().splitlines() if "train step" in l) if os.path.exists(plain) else ""
    out += [f"## {title}", ""]
    if plain_line:
        out += [f"without the profiler, same box: `{plain_line}`", ""]
    out += [f"under rocprofv3 (host-side tracing slows the launches: wall time is NOT the step's): `{line}`", "",
            f"GPU time of the 9 profiled steps {tot / 1e6:.1f} ms = {tot / 9e6:.2f} ms per step; on kernels of libslimmoe_hip.so **{100 * mine / tot:.1f} %**", "",
            "| ms (9 steps) | calls | own | kernel |", "|---|---|---|---|"]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
        out.append(f"| {float(r['TotalDurationNs']) / 1e6:.2f} | {r['Calls']} | {'yes' if own(r['Name']) else 'torch'} | `{r['Name'][:120]}` |")
    out.append("")
    out.append("torch kernels left: " + "; ".join(f"{r['Name'][:70]} {float(r['TotalDurationNs']) / 1e6:.2f} ms" for r in
                                              sorted((r for r in rows if not own(r["Name"])), key=lambda r: -float(r["TotalDurationNs"]))[:8]))
    out.append("")
open(os.path.join(dst, f"{tag}_train_step.md"), "w").write("\n".join(out))
print("wrote", [f for f in sorted(os.listdir(dst)) if f.startswith(tag)])
