#!/bin/bash
# Build named variants of libdsp_frontend.so for kernel A/B runs: tools/build_variants.sh name "flags" [name "flags" ...]
set -e
cd "$(dirname "$0")/../dsp-speech-recognition_amd/csrc"
mkdir -p ../lib/variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wall -Wno-unused-function \
      -ffp-contract=fast -fno-gpu-rdc -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 $flags -shared -o ../lib/variants/$name.so dsp_frontend.hip 2>&1 | grep -E "error|warning: v" ; echo "built $name" ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
