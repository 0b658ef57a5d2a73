#!/bin/bash
# Developer aid: build variant libraries of ONE source of the library with extra -D flags (loaded with XDTTS_LIB).
#   [SRC=decoder_persistent8.hip] tools/variants.sh tag1 "-DA=1 -DB=2" tag2 "-DC" ...   -> xd-tts_amd/libxdtts_hip_v_<tag>.so
set -e
cd "$(dirname "$0")/../xd-tts_amd"
SRC=${SRC:-decoder.hip}
make -j8 >/dev/null
extra=""
case $SRC in decoder_persistent*.hip) extra="-fno-slp-vectorize";; esac
others=$(ls build/*.o | grep -v "build/$SRC.o")
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  mkdir -p build_v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $extra $flags -x hip -c csrc/$SRC -o build_v/${SRC}_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libxdtts_hip_v_$tag.so $others build_v/${SRC}_$tag.o
  echo "built libxdtts_hip_v_$tag.so ($SRC: $flags)"
done
