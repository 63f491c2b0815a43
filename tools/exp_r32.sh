#!/bin/bash
# Same-box A/B of the N = 16384 kernels: the radix-32 form (default), the radix-16 form (lds_variant=16) and every r32_* variant build under
# gnss-dsp-tools_amd/build/variants/: tools/bench_configs.py cfg5_b1i cfg5_glonass, 'ms' per search, two alternating rounds.
cd "$(dirname "$0")/.."
CASES=${CASES:-"cfg5_b1i cfg5_glonass"}
run() { python -c "
import sys, json
out=[]
for l in sys.stdin.read().splitlines():
    if l.startswith('{'):
        d=json.loads(l); out.append('%s %.4f' % (d['case'], d['ms']))
print('  '.join(out))
"; }
for r in 1 2; do
  printf "%-12s" radix16; python tools/bench_configs.py --reps 8 --option lds_variant=16 $CASES 2>/dev/null | run
  printf "%-12s" radix32; python tools/bench_configs.py --reps 8 $CASES 2>/dev/null | run
  for v in gnss-dsp-tools_amd/build/variants/r32_*; do
    [ -d "$v" ] || continue
    v=$(basename $v); printf "%-12s" $v; python tools/variant.py $v tools/bench_configs.py --reps 8 $CASES 2>/dev/null | run
  done
done
