#!/bin/bash
# Samples the shader clock / socket power with rocm-smi while bench.py's headline step runs for ~30 s (is the kernel power-limited?)
cd "$(dirname "$0")/.."
python bench.py --steps 6000 --warmup 5 --no-cpu-baseline --no-others --no-pmc --no-latency --sustained-s 0 > /tmp/b.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/\t//g'; echo
  sleep 1
done
tail -1 /tmp/b.log | cut -c1-260
