#!/bin/bash
# Register / spill / scratch numbers of the kernels in an object file of csrc/ (from the code object's metadata notes).
# usage: tools/kernel_regs.sh slim-switch-moe-vit_amd/csrc/gemm.o [name-filter]
set -e
obj=$1; filt=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$obj" $tmp/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/k.co > $tmp/notes.txt
python3 - "$tmp/notes.txt" "$filt" <<'PY'
import re, sys, subprocess
t = open(sys.argv[1]).read()
for b in t.split('- .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    except Exception:
        pass
    if not re.search(sys.argv[2], name):
        continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, b).group(1)
    print(f"{name[:110]:110s} agpr {b.split()[0]:>3s} vgpr {g('vgpr_count'):>3s} vspill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>3s} "
          f"sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s}")
PY
rm -rf $tmp
