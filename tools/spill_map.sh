#!/bin/bash
# Where a kernel's spill traffic sits: for every kernel of libmixgan_hip.so whose name matches $1, the scratch loads (L) and
# stores (S) and barriers (B) in program order, each tagged with the number of MFMA instructions in front of it (static
# count: loops are rolled, so ops between two equal counts that lie inside a loop body run every iteration).
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cp "$here/mixgan-tts_amd/libmixgan_hip.so" "$tmp/lib.so"
cd "$tmp"
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null
for f in *.hipv4-amdgcn-amd-amdhsa--gfx950; do
    for k in $(/opt/rocm/lib/llvm/bin/llvm-readelf -s "$f" 2>/dev/null | awk '$4 == "FUNC" {print $8}' | grep -E "${1:-.}" | sort -u); do
        /opt/rocm/lib/llvm/bin/llvm-objdump -d "$f" --disassemble-symbols="$k" > k.s
        python3 - "$k" <<'PY'
import re, sys
m = s = 0
out = []
for l in open('k.s'):
    g = re.match(r'\s+(\S+)\s', l)
    if not g or '//' not in l:
        continue
    op = g.group(1)
    if 'mfma' in op:
        m += 1
    elif op.startswith('scratch_'):
        out.append("%d:%s" % (m, 'L' if 'load' in op else 'S'))
    elif op == 's_barrier':
        out.append("%d:B" % m)
    elif op.startswith('s_cbranch') or op == 's_branch':
        out.append("%d:j" % m)
print(sys.argv[1], "mfma", m)
print(' '.join(out))
PY
    done
done
