#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace -o bench -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-latency "$@" > $REPO/$OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/$OUT/pmc_fetch -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-latency "$@" > $REPO/$OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/$OUT/pmc_write -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-latency "$@" > $REPO/$OUT/bench_pmc_write.log 2>&1
cd $REPO
python tools/summarize_profile.py $OUT $TAG > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
