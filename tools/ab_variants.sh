# Same-box A/B of library variants (tools/build_variant.sh): per-config bench lines of the product library and of each named variant, interleaved.
# usage (on the GPU box, repo root): bash tools/ab_variants.sh "<configs>" <rounds> <variant> [variant ...]
CONFIGS=$1; ROUNDS=$2; shift 2
F="--steps 20 --warmup 5 --no-others --no-cpu-baseline --no-latency --no-pmc"
show() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-10s'%sys.argv[1], j['config']['baseline_config'], '%.4g'%j['value'], '%.3f ms'%j['ms_per_step'], 'sustained %.3f ms'%j['sustained']['ms_per_step'])" "$1"; }
for i in $(seq $ROUNDS); do for c in $CONFIGS; do
  python bench.py --config $c $F 2>/dev/null | show product
  for v in "$@"; do python tools/variant.py $v bench.py --config $c $F 2>/dev/null | show $v; done
done; done
