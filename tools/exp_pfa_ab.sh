# Same-box A/B of the forms of engine 3 at N = 61380 / 30690 (round 5): prime-factor form with the DFT-31 on the VALU (split_mfma=0) / on the
# matrix pipe (split_mfma=1), Cooley-Tukey form with the Stockham kernel of rounds 1-4 (fused_inner=2; removed after this log was taken) or
# rocFFT inner transforms (fused_inner=0).  usage (GPU box, repo root): bash tools/exp_pfa_ab.sh
F="--steps 20 --warmup 5 --no-others --no-cpu-baseline --no-latency --no-pmc"
show() { python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s'%sys.argv[1], 'config 4: %.4g cells/s'%j['value'], '%.3f ms/step'%j['ms_per_step'], 'sustained %.3f ms'%j['sustained']['ms_per_step'])" "$1"; }
for r in 1 2; do
  for o in "fused_inner=1 split_mfma=0" "fused_inner=1 split_mfma=1" "fused_inner=2 split_mfma=0" "fused_inner=0 split_mfma=0"; do
    opts=""; for kv in $o; do opts="$opts --option $kv"; done
    python bench.py --config 4 $F $opts 2>/dev/null | show "$o"
  done
done
echo "# per-signal stage times (ms): mix_nco = forward stage, lds_correlate = inner (writer of Z'), mag_peak = outer inverse (reader of Z')"
for o in "fused_inner=1 split_mfma=0" "fused_inner=1 split_mfma=1" "fused_inner=2 split_mfma=0"; do
  opts=""; for kv in $o; do opts="$opts --option $kv"; done
  echo "== $o"; python tools/bench_configs.py --stages --reps 20 $opts cfg4_l5i cfg4_b2ad_b1 gal_e6b 2>/dev/null | grep -v "^{"
done
