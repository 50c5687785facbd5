#!/bin/bash
# Build a variant of the working-tree engine into build/var_<name>/libflacgpu.so: the listed sources recompiled with extra
# compiler flags, everything else taken from flac_amd/lib/*.o.  usage: scripts/variant_build.sh <name> "<flags>" <source.hip> [...]
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/var_$NAME
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fconstexpr-steps=50000000 -I$ROOT/include -I$ROOT/flac_amd/csrc -Wall -Wno-unused-function"
OBJS=""
for o in $ROOT/flac_amd/lib/flacgpu_*.o; do
  b=$(basename $o .o); use=$o
  for s in "$@"; do if [ "$(basename $s .hip)" = "$b" ] || [ "$(basename $s .cpp)" = "$b" ]; then
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $ROOT/flac_amd/csrc/$s -o $OUT/$b.o; use=$OUT/$b.o; fi; done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libflacgpu.so $OBJS
echo "variant $NAME: $EXTRA ($*)"
