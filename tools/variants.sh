#!/bin/bash
# Developer aid: build variant libraries of csrc/decoder.hip with extra -D flags (loaded with XDTTS_LIB).
#   tools/variants.sh tag1 "-DA=1 -DB=2" tag2 "-DC" ...   -> xd-tts_amd/libxdtts_hip_v_<tag>.so
set -e
cd "$(dirname "$0")/../xd-tts_amd"
make -j8 >/dev/null
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  mkdir -p build_v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result $flags -x hip -c csrc/decoder.hip -o build_v/decoder_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libxdtts_hip_v_$tag.so build/api.cpp.o build/host.cpp.o build/weights.cpp.o build/onnx_load.cpp.o build_v/decoder_$tag.o build/decoder_persistent.hip.o build/encoder.hip.o build/gemm.hip.o build/griffinlim.hip.o
  echo "built libxdtts_hip_v_$tag.so ($flags)"
done
