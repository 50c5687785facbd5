#!/bin/bash
# Device ISA of one source (development aid): scripts/isa.sh <file.hip> [extra flags] -> /tmp/isa/<name>.s
# then: scripts/isa_count.py /tmp/isa/<name>.s <kernel substring>   (instruction mix per kernel / per label range)
set -e
SRC=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fconstexpr-steps=50000000 -I$ROOT/include -I$ROOT/flac_amd/csrc \
  -Wno-unused-function "$@" --cuda-device-only -S -x hip $SRC -o /tmp/isa/$(basename ${SRC%.*}).s
echo /tmp/isa/$(basename ${SRC%.*}).s
