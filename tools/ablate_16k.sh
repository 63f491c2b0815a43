# N = 16384 correlate kernel, profiling-only ablations (wrong results by construction): bit 0 no LDS traffic in the 4096-point
# sub-transforms, bit 1 no barriers there, bit 3 no inter-pass twiddles, bit 4 no X loads, bit 5 no workgroup-wide barriers in fft16k
for a in 0 16 32 48 50 59; do
  echo "== ablation $a"
  if [ $a = 0 ]; then python tools/bench_configs.py --stages cfg5_b1i 2>&1 | grep -E "stages"
  else bash tools/ablate.sh --run $a -- python tools/bench_configs.py --stages cfg5_b1i 2>&1 | grep -E "stages"; fi
done
