python tools/parity_sweep.py > gpurun_out/parity_sweep.log 2>&1
for c in 2 3 4 5; do python bench.py --config $c --steps 20 --warmup 5 --no-others 2>/dev/null | grep '^{' ; done > gpurun_out/bench_lines.json
python bench.py > gpurun_out/bench_default_line.json 2>/dev/null
python tools/profile_round.py r03 > gpurun_out/profile_round.log 2>&1
tail -12 gpurun_out/parity_sweep.log
python - <<'PY'
import json
for l in open("gpurun_out/bench_lines.json"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["baseline_config"], "%.4g"%j["value"], "%.3f ms"%j["ms_per_step"], r["bound"], r["kernel"], "%.3f"%r["frac"], r.get("traffic"), (r.get("traffic_source") or {}).get("measured_in_this_run"), "sustained %.4g"%j["sustained"]["value"], "cpu %.3g"%j["cpu_baseline"]["value"])
PY
