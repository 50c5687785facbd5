#!/bin/bash
# The host C layer (flac_amd/csrc/host/*.c: the libFLAC encoder API, MD5, Ogg, windows, verify) under AddressSanitizer + UndefinedBehaviorSanitizer
# on the CPU: a scratch copy of the tree, the host objects rebuilt with -fsanitize=address,undefined (the HIP library as shipped), the CPU
# tests that drive the host layer run with the sanitizer runtime preloaded.  GPU sanitizers are not available on the pool; this is the
# part of the product that is plain C.  usage: scripts/asan_host.sh [pytest args]   -> profiles-ready summary on stdout
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=${ASAN_TMP:-/tmp/asan_repo}
rm -rf $T && mkdir -p $T
(cd $ROOT && tar cf - --exclude=.git --exclude=gpurun_out --exclude=__pycache__ --exclude=.pytest_cache .) | (cd $T && tar xf -)
cd $T/flac_amd/csrc
rm -f ../lib/host_*.o ../lib/libFLACgpu.so ../lib/libFLAC.so.14 $T/tests/fake_engine/libflacgpu.so
make -s HOST_CFLAGS="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=gnu11 -fPIC -Wall -Wextra -ffp-contract=off -fno-fast-math -I$T/include -I$T/flac_amd/csrc/host" CC="gcc -fsanitize=address,undefined" 2>&1 | grep -v "^make\|Nothing" | tail -5
cd $T
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:detect_odr_violation=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
python -m pytest tests/test_host_pipeline_cpu.py tests/test_md5_multi.py tests/test_ogg_cpu.py tests/test_verify_cpu.py tests/test_abi.py tests/test_kat.py tests/test_decode_pin.py -x -q -m "not gpu" -p no:cacheprovider "$@" 2>&1 | tail -15
