#!/bin/bash
# usage: tools/pmc.sh <tag> "<COUNTER ...>" [bench args]   -- one rocprofv3 --pmc pass, per-kernel averages
TAG=$1; CNT=$2; shift 2
OUT=$(pwd)/gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
rocprofv3 --pmc $CNT --output-format csv -d $OUT -o p -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc "$@" > $OUT/bench.log 2>&1
cd $REPO
python - $OUT <<'PY'
import csv, glob, sys, os
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"] or 0)
for k, cs in acc.items():
    if "correlate" in k or "forward" in k or "fft" in k.lower():
        print(k[:100])
        for c, disp in sorted(cs.items()):
            v = list(disp.values())
            print("   %-28s avg/launch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
