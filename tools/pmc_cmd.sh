#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> "<COUNTER ...>" <kernel-name-substring> -- <command...>   (one rocprofv3 --pmc pass, per-kernel averages)
TAG=$1; CNT=$2; FILT=$3; shift 4
OUT=$(pwd)/gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd); cd /tmp
if [ "$CNT" = "TRACE" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- "$@" > $OUT/cmd.log 2>&1
  cd $REPO; python - $OUT <<'PY'
import csv, glob, sys, os
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(path)))[:14]:
        print("%-90s calls %-5s avg %.1f us  total %.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
else
  rocprofv3 --pmc $CNT --output-format csv -d $OUT -o p -- "$@" > $OUT/cmd.log 2>&1
  cd $REPO; python - $OUT "$FILT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
d, filt = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"] or 0)
for k, cs in acc.items():
    if filt in k:
        print(k[:110])
        for c, disp in sorted(cs.items()):
            v = list(disp.values())
            print("   %-28s avg/launch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
fi
