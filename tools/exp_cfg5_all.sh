# Config 5 on one GPU: where the time between "sum of the four searches alone" (6.6 ms) and the step (7.3-7.9 ms) goes, and what keeping
# work side by side buys.  Everything into gpurun_out/config5_clocks_and_overlap.log.  usage: bash tools/exp_cfg5_all.sh
L=gpurun_out/config5_clocks_and_overlap.log
f() { grep "^{"; }
{
echo "== 1. shader clock / socket power while each search loops alone, while the step loops, with one and two steps in flight (tools/exp_cfg5_clocks.py)"
timeout 300 python tools/exp_cfg5_clocks.py 2>&1 | f
echo; echo "== 2. every queueing order of the four searches on one stream (tools/exp_cfg5_order.py): best, listing order, worst"
timeout 300 python tools/exp_cfg5_order.py 2>&1 | f | sed -n '1,3p;22,24p'
echo; echo "== 3. two steps in flight on plain torch streams: overlap or not depends on which hardware queues the streams land on (tools/exp_lanes_debug.py; second run: one more, idle, context created first)"
timeout 200 python tools/exp_lanes_debug.py 2>&1 | f | tail -4
timeout 200 python tools/exp_lanes_debug.py --extra-engine 2>&1 | f | tail -4
echo; echo "== 4. steps / jobs side by side on hardware queues of their own (tools/exp_cfg5_overlap.py)"
timeout 400 python tools/exp_cfg5_overlap.py 2>&1 | f
echo; echo "== 5. bench.py --config N --lanes 1 / 2 / 3 (fresh process each)"
for c in 2 3 4 5; do for Ln in 1 2 3; do timeout 200 python bench.py --config $c --lanes $Ln --steps 20 --warmup 5 --no-others --no-cpu-baseline --no-latency --no-pmc 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); c=j['sustained']['clocks']; print('config', j['config']['baseline_config'], 'steps in flight', j['config'].get('steps_in_flight'), '%.4g cells/s'%j['value'], '%.3f ms/step'%j['ms_per_step'], 'sustained %.4g'%j['sustained']['value'], '%.0f MHz %.0f W'%(c['sclk_mhz_mean'], c['power_w_mean']))
"; done; done
} > $L 2>&1
tail -30 $L
