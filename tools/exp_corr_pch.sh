cd $GRAFT_REPO_ROOT
for p in 0 1 2 4 8 16; do
  echo "== lds_pch=$p"; python tools/bench_configs.py --reps 10 --option lds_pch=$p gps_l1_ms10 gps_l1_cli_ms80 2>/dev/null | python -c "
import sys, json
for l in sys.stdin.read().splitlines():
    if l.startswith('{'):
        d=json.loads(l); print('  %-18s %.4f ms' % (d['case'], d['ms']))
"
done
