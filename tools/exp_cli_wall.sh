#!/bin/bash
# Wall time of the command-line shim as a PROGRAM (one process per run, like the reference's scripts), all defaults, 85 ms of int8 IQ at 69.984 MS/s.
cd "$(dirname "$0")/.."
python - <<'PY'
import numpy as np
rng = np.random.default_rng(3)
n = int(69984000 * 0.085)
(rng.standard_normal(2 * n) * 20).clip(-127, 127).astype(np.int8).tofile("/tmp/rec.iq")
PY
for name in gps-l1 beidou-b1i gps-l5i beidou-b2ad galileo-e1b; do
  for i in 1 2; do
    s=$(date +%s.%N); python -m gnss_dsp_tools_amd.cli $name /tmp/rec.iq 69984000 0 > /tmp/out_$name.txt 2>/tmp/err_$name.txt; rc=$?; e=$(date +%s.%N)
    echo "$name run $i: rc $rc, $(wc -l < /tmp/out_$name.txt) lines, $(python -c "print('%.2f s' % ($e - $s))")"
  done
done
