#!/bin/bash
# Run tools/kbench.py once per variant library (alphabetical; "base" first as the parity reference).
# usage: bash tools/kbench_variants.sh [kbench args]
mkdir -p gpurun_out
rm -f /tmp/kb_ref.npy
V=dsp-speech-recognition_amd/lib/variants
for so in $V/base.so $(ls $V/*.so | grep -v /base.so); do
  DSP_FRONTEND_LIB=$PWD/$so python tools/kbench.py --check /tmp/kb_ref.npy "$@" 2>&1 | grep -v amdgpu.ids
done
