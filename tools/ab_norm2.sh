#!/bin/bash
# A/B of the |z|^2 spelling (norm2 in gacq_cplx.h): product build (mul + fma) against -DGACQ_NORM2_PACKED (the packed multiply + add
# hipcc's vectoriser picks for x*x + y*y).
#   build container:  tools/ab_norm2.sh --build      -> gnss-dsp-tools_amd/build/ablnorm/libgacq.so
#   GPU box:          tools/ab_norm2.sh              -> GPU tests on the product build, then four alternations of the config-2 bench line
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" = "--build" ]; then
  cd "$ROOT/gnss-dsp-tools_amd/csrc" && mkdir -p ../build/ablnorm
  for f in gacq_ldsfft gacq_split; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -DGACQ_NORM2_PACKED -c $f.hip -o ../build/ablnorm/$f.o || exit 1
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/ablnorm/libgacq.so $(ls ../build/*.o | grep -v -e gacq_split.o -e gacq_ldsfft.o) \
    ../build/ablnorm/gacq_split.o ../build/ablnorm/gacq_ldsfft.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
  exit $?
fi
cd "$ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3 4; do
  for v in packed fma; do
    if [ $v = packed ]; then r=$(bash tools/ablate.sh --run norm -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-latency 2>/dev/null | grep '^{'); else r=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-latency 2>/dev/null | grep '^{'); fi
    echo "$v $(echo "$r" | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.4g cells/s  %.4f ms  sustained %.4g' % (j['value'], j['ms_per_step'], j['sustained']['value']))")"
  done
done
