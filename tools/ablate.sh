#!/bin/bash
# Build profiling-only ablation variants of the engine (wrong results by construction) into build/abl<N>/libgacq.so.
# usage (build container): tools/ablate.sh 1 2 3 ... (bits of GACQ_ABL in gacq_ldsfft.hip) or s1 ... (GACQ_ABL_SPLIT in gacq_split.hip) or nat
# on the GPU box (scratch copy of the repo): tools/ablate.sh --run N -- python bench.py --no-self-check ...
#   swaps lib/libgacq.so for the variant for the duration of the command and puts the product build back afterwards
#   (the product never loads a library from anywhere but lib/libgacq.so).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" = "--run" ]; then
  a=$2; shift 3
  cp "$ROOT/gnss-dsp-tools_amd/lib/libgacq.so" "$ROOT/gnss-dsp-tools_amd/lib/libgacq.so.product"
  cp "$ROOT/gnss-dsp-tools_amd/build/abl$a/libgacq.so" "$ROOT/gnss-dsp-tools_amd/lib/libgacq.so"
  "$@" || true
  mv "$ROOT/gnss-dsp-tools_amd/lib/libgacq.so.product" "$ROOT/gnss-dsp-tools_amd/lib/libgacq.so"
  exit 0
fi
cd "$ROOT/gnss-dsp-tools_amd/csrc"
for a in "$@"; do
  mkdir -p ../build/abl$a
  if [ "$a" = "nat" ]; then              # engine 4 with Z' rows in natural order (8-byte stores): the A/B partner of the lane-pair layout
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -DGACQ_ABL=64 -c gacq_ldsfft.hip -o ../build/abl$a/gacq_ldsfft.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -DGACQ_ABL_SPLIT=128 -c gacq_split.hip -o ../build/abl$a/gacq_split.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/abl$a/libgacq.so $(ls ../build/*.o | grep -v -e gacq_split.o -e gacq_ldsfft.o) ../build/abl$a/gacq_split.o ../build/abl$a/gacq_ldsfft.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
    continue
  fi
  if [ "${a:0:1}" = "s" ]; then        # sN: ablation N of the split engines (GACQ_ABL_SPLIT in gacq_split.hip)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -DGACQ_ABL_SPLIT=${a:1} -c gacq_split.hip -o ../build/abl$a/gacq_split.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/abl$a/libgacq.so $(ls ../build/*.o | grep -v gacq_split.o) ../build/abl$a/gacq_split.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
    continue
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -DGACQ_ABL=$a -c gacq_ldsfft.hip -o ../build/abl$a/gacq_ldsfft.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/abl$a/libgacq.so $(ls ../build/*.o | grep -v gacq_ldsfft.o) ../build/abl$a/gacq_ldsfft.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
done
