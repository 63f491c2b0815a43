#!/bin/bash
# GPU box (after `make -C gnss-dsp-tools_amd/csrc timing` here): swap the -DGACQ_PHASE_TIMING build of the split engine into lib/,
# run tools/phase_timing.py [split_dt], put the product build back.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cp "$ROOT/gnss-dsp-tools_amd/lib/libgacq.so" /tmp/libgacq_product.so
cp "$ROOT/gnss-dsp-tools_amd/build/timing/libgacq.so" "$ROOT/gnss-dsp-tools_amd/lib/libgacq.so"
python "$ROOT/tools/phase_timing.py" "$@" 2>&1 | grep -v amdgpu.ids
cp /tmp/libgacq_product.so "$ROOT/gnss-dsp-tools_amd/lib/libgacq.so"
