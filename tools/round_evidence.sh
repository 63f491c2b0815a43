# Round evidence on the GPU box (repo root): parity sweep, per-config bench lines, the default driver line, rocprofv3 kernel statistics +
# PMC counters (tools/profile_round.py), the near-tie census and a randomised differential run.  Everything lands under gpurun_out/ and
# is copied into profiles/ by hand afterwards.  usage: bash tools/round_evidence.sh r04
TAG=${1:-r06}
# second argument: the commit the evidence is taken at (the GPU box has no .git): bash tools/round_evidence.sh r06 $(git rev-parse --short HEAD)
export GACQ_EVIDENCE_HEAD=${2:-unknown}
echo "$GACQ_EVIDENCE_HEAD" > gpurun_out/evidence_head.txt
timeout 900 python tools/parity_sweep.py > gpurun_out/parity_sweep.log 2>&1
for c in 2 3 4 5; do timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-others 2>/dev/null | grep '^{' ; done > gpurun_out/bench_lines.json
timeout 900 python bench.py > gpurun_out/bench_default_line.json 2>/dev/null
timeout 1500 python tools/profile_round.py $TAG > gpurun_out/profile_round.log 2>&1
timeout 400 python tools/tie_census.py 20 4242 > gpurun_out/tie_census.log 2>&1
timeout 400 python tools/fuzz_engines.py 300 5501 > gpurun_out/fuzz_engines.log 2>&1
timeout 200 python tools/exp_epl_latency.py > gpurun_out/epl_latency.log 2>&1
tail -12 gpurun_out/parity_sweep.log
python - <<'PY'
import json
for l in open("gpurun_out/bench_lines.json"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["baseline_config"], "%.4g"%j["value"], "%.3f ms"%j["ms_per_step"], r["bound"], r["kernel"], "%.3f"%r["frac"], r.get("traffic"), (r.get("traffic_source") or {}).get("measured_in_this_run"), "sustained %.4g"%j["sustained"]["value"], "cpu %.3g"%j["cpu_baseline"]["value"], j.get("tie_safe"))
PY
tail -3 gpurun_out/tie_census.log | cut -c1-300
tail -1 gpurun_out/fuzz_engines.log | cut -c1-600
