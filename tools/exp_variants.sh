#!/bin/bash
# Same-box A/B of the product library against variant builds (tools/build_variant.sh):
#   tools/exp_variants.sh "<command run from the repo root, e.g. bench.py --config 4 ...>" <variant> [<variant> ...]
# The command is a python script + arguments; it is run for the product and for every variant (through tools/variant.py), interleaved, two rounds.
# bench.py lines are reduced to "ms_per_step value"; other output is passed through without the JSON lines.
cd "$(dirname "$0")/.."
CMD=$1; shift
show() { python -c "
import sys, json
for l in sys.stdin.read().splitlines():
    if l.startswith('{'):
        d = json.loads(l)
        if 'ms_per_step' in d: print('  %.4f ms/step  %.4g %s' % (d['ms_per_step'], d['value'], d.get('unit', '')))
        elif 'ms' in d: print('  %-22s %.4f ms  %.4g cells/s' % (d.get('case', ''), d['ms'], d.get('cells_per_s', 0)))
    elif l.strip() and 'amdgpu.ids' not in l: print(l)
"; }
for round in 1 2; do
  echo "== product"; python $CMD 2>/dev/null | show
  for v in "$@"; do echo "== $v"; python tools/variant.py $v $CMD 2>/dev/null | show; done
done
