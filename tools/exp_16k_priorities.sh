#!/bin/bash
# One round over the product and the qABCD variants (s_setprio levels GACQ_P1..P4 of the 16384-point inverse transform:
# last radix-16 pass + magnitudes | C x + radix-4 | radix-16 | radix-16): tools/bench_configs.py cfg5_b1i cfg5_glonass, 'ms' per search.
cd "$(dirname "$0")/.."
run() { python -c "
import sys, json
out=[]
for l in sys.stdin.read().splitlines():
    if l.startswith('{'):
        d=json.loads(l); out.append('%s %.4f' % (d['case'], d['ms']))
print('  '.join(out))
"; }
for r in 1 2; do
  printf "%-8s" product; python tools/bench_configs.py --reps 8 cfg5_b1i cfg5_glonass 2>/dev/null | run
  for v in gnss-dsp-tools_amd/build/variants/q*; do
    v=$(basename $v); printf "%-8s" $v; python tools/variant.py $v tools/bench_configs.py --reps 8 cfg5_b1i cfg5_glonass 2>/dev/null | run
  done
done
