#!/bin/bash
# Builds a tuning / diagnostic variant of libgacq.so WITHOUT touching the product library:
#   tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> [file.hip ...]
# compiles the listed translation units with the extra flags into gnss-dsp-tools_amd/build/variants/<name>/ and links them with the
# product objects of all other units -> build/variants/<name>/libgacq.so.  Load it with tools/variant.py (GACQ_TUNING_LIB hook of
# _native.py).  The product build (make -C gnss-dsp-tools_amd/csrc) must exist.
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/gnss-dsp-tools_amd/csrc; BUILD=$ROOT/gnss-dsp-tools_amd/build; OUT=$BUILD/variants/$NAME
ROCM=${ROCM:-/opt/rocm}
mkdir -p "$OUT"
OBJS=""
for o in gacq_engine gacq_ldsfft gacq_lds16k gacq_split gacq_pfa gacq_frontend gacq_longcode gacq_tracking gacq_host gacq_verify gacq_tiesafe gacq_probe prn_codes; do
  use=$BUILD/$o.o
  for f in "$@"; do
    if [ "$(basename "$f" .hip)" = "$o" ]; then
      $ROCM/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I$ROCM/include $FLAGS -c "$CSRC/$o.hip" -o "$OUT/$o.o"
      use=$OUT/$o.o
    fi
  done
  OBJS="$OBJS $use"
done
$ROCM/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgacq.so" $OBJS -L$ROCM/lib -lrocfft -Wl,-rpath,$ROCM/lib
echo "$OUT/libgacq.so"
