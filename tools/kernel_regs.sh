#!/bin/bash
# Register / LDS / spill figures of every kernel in libmixgan_hip.so whose name matches $1 (default: all).
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cd "$tmp"
cp "$here/mixgan-tts_amd/libmixgan_hip.so" lib.so
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null
for f in *.hipv4-amdgcn-amd-amdhsa--gfx950; do
    /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | awk -v pat="${1:-.}" '
        /\.agpr_count:/ {a=$2} /\.group_segment_fixed_size:/ {l=$2} /\.name:/ {n=$2} /\.private_segment_fixed_size:/ {p=$2}
        /\.sgpr_count:/ {s=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {sp=$2; if (n ~ pat) printf "%-110s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d spills %d\n", n, v, a, s, l, p, sp}'
done | sort -u
rm -rf "$tmp"
