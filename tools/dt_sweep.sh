# Doppler bins per workgroup of the Stockham inner kernel: tests, then three alternating passes over 1/2/3.
python -m pytest tests -m gpu -x -q -k "radix31 or teams or full_size or golden or split" 2>&1 | tail -3
for rep in 1 2; do
for dt in 1 2 3 4; do
  echo "== split_dt=$dt rep $rep"
  python tools/bench_configs.py --stages --option split_dt=$dt cfg4_l5i cfg4_b2ad_b1 gal_e6b 2>&1 | grep -E "stages|case" | sed -e 's/"cell_blocks.*//' -e 's/"signal.*"ms"/"ms"/'
done
done
