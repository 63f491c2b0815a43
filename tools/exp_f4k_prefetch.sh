#!/bin/bash
# A/B of the headline kernel (lds_fused4k_kernel<4, PREA, false>) with the next item's code-spectrum row prefetched:
#   f4k_pf  (-DGACQ_F4K_PF=1): loads issued right after the magnitudes (into the registers the row just freed), under the peak search + barrier
#   f4k_pf2 (-DGACQ_F4K_PF=2): loads issued a whole row ahead; the 32 registers come from giving up the resident pass-1 twiddle powers (PREA off)
# Same box, interleaved, two rounds.  Build: tools/build_variant.sh f4k_pf "-DGACQ_F4K_PF=1" gacq_ldsfft.hip (same for f4k_pf2).
cd "$(dirname "$0")/.."
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-others --no-pmc --no-latency"
for round in 1 2; do
  echo "== product"; python bench.py $ARGS | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('self_check'))"
  for v in "$@"; do
    echo "== $v"; python tools/variant.py $v bench.py $ARGS | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('self_check'))"
  done
done
