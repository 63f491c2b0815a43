#!/bin/bash
# A/B the LDS correlate kernel variants on the GPU box: prints avg kernel ms + cells/s per variant.
for v in ${VARIANTS:-0 1 2 3 4 5}; do
  timeout 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --option lds_variant=$v "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line)
        st = j['pipeline']['stages']
        print('variant $v: %.4g cells/s  ms/step %.3f  correlate %.3f ms  forward %.3f ms' % (j['value'], j['ms_per_step'], st.get('lds_correlate',{}).get('avg_ms',-1), st.get('mix_nco',{}).get('avg_ms',-1)))
"
done
