#!/bin/bash
# A second build of libmixgan_hip.so with extra compiler flags for ONE source file, for A/B timing inside one gpurun call:
#   tools/ab_build.sh denoiser.hip "-DDP_EARLY=0" mixgan-tts_amd/libmixgan_hip_B.so
#   MG_HIP_LIB=$PWD/mixgan-tts_amd/libmixgan_hip_B.so python bench.py ...
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
src=$1; flags=$2; out=$3
cd "$here/mixgan-tts_amd/csrc"
make -j8 >/dev/null
mkdir -p build_ab
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function $flags -c "$src" -o "build_ab/${src%.hip}.o"
objs=""
for f in build/*.o; do
    b=$(basename "$f")
    if [ "$b" = "${src%.hip}.o" ]; then objs="$objs build_ab/$b"; else objs="$objs $f"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/$out" $objs
echo "built $out"
