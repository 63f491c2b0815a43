python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3 4; do
  for v in packed fma; do
    if [ $v = packed ]; then r=$(bash tools/ablate.sh --run norm -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-latency 2>/dev/null | grep '^{'); else r=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-latency 2>/dev/null | grep '^{'); fi
    echo "$v $(echo "$r" | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.4g cells/s  %.4f ms  sustained %.4g' % (j['value'], j['ms_per_step'], j['sustained']['value']))")"
  done
done
