#!/bin/bash
# A/B of the headline kernel (lds_fused4k_kernel<4, PREA, false>) against variant builds, same box, interleaved, two rounds:
#   tools/exp_f4k_prefetch.sh <variant> [<variant> ...]
#   f4k_pf  (-DGACQ_F4K_PF=1): next item's code-spectrum loads issued right after the magnitudes (into the registers the row just freed)
#   f4k_pf2 (-DGACQ_F4K_PF=2): loads issued a whole row ahead; the 32 registers come from giving up the resident pass-1 twiddle powers
#   f4k_prio1 / f4k_prio2 (-DGACQ_F4K_PRIO=1|2): falling / rising wave priority through the four segments of a row
#   f4k_nosqrt (-DGACQ_ABL_NOSQRT): ablation, wrong results: the row without its 16 square roots per lane
# Build: tools/build_variant.sh <variant> "<flag>" gacq_ldsfft.hip
cd "$(dirname "$0")/.."
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-others --no-pmc --no-latency --no-self-check"
for round in 1 2; do
  echo "== product"; python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  for v in "$@"; do
    echo "== $v"; python tools/variant.py $v bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  done
done
