#!/bin/bash
# Phase timing of the Stockham inner kernel (diagnostic).  Build container: tools/phase_timing.sh --build  (a -DGACQ_PHASE_TIMING variant of
# gacq_split.hip under build/variants/timing/).  GPU box: tools/phase_timing.sh [split_dt].  The product library in lib/ is never touched:
# the variant is loaded through tools/variant.py.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" = "--build" ]; then exec "$ROOT/tools/build_variant.sh" timing -DGACQ_PHASE_TIMING gacq_split.hip; fi
python "$ROOT/tools/variant.py" timing "$ROOT/tools/phase_timing.py" "$@" 2>&1 | grep -v amdgpu.ids
