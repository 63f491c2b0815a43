#!/bin/bash
# Build profiling-only ablation variants of the engine (wrong results by construction) into lib/libgacq_abl<N>.so
# usage (build container): tools/ablate.sh 1 2 3 ...   ; on the GPU box: GACQ_LIB=.../libgacq_abl1.so python bench.py ...
set -e
cd "$(dirname "$0")/../gnss-dsp-tools_amd/csrc"
for a in "$@"; do
  mkdir -p ../build/abl$a
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -DGACQ_ABL=$a -c gacq_ldsfft.hip -o ../build/abl$a/gacq_ldsfft.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgacq_abl$a.so $(ls ../build/*.o | grep -v gacq_ldsfft.o) ../build/abl$a/gacq_ldsfft.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
