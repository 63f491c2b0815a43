#!/bin/bash
# Config 5's E1B shape (50 items x 200 bins x 65536 lags, B = 1: 5.24 GB of Z') against the workspace limit = the size of one Z' pass,
# every point in a FRESH process (first search of a process included separately): ms per search, tools/bench_configs.py cfg5_e1b
cd "$(dirname "$0")/.."
for ws in 512 1024 2048 3072 4096 6144 8192 16384 32768; do
  printf "ws %6d MiB  " $ws
  python tools/bench_configs.py --reps 12 --ws $ws --stages cfg5_e1b 2>/dev/null | python -c "
import sys, json
st=''
for l in sys.stdin.read().splitlines():
    if l.strip().startswith('stages'): st=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%.4f ms   %s' % (d['ms'], st))
"
done
