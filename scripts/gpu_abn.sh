#!/bin/bash
# On the GPU box: alternate bench runs between the working-tree engine (A), build/alt_lib (B, if there) and every build/var_*/
# variant; the order rotates every round.  usage: scripts/gpu_abn.sh [rounds] [bench args]
R=${1:-3}; shift || true
names="A"; cp flac_amd/lib/libflacgpu.so /tmp/v_A.so
[ -f build/alt_lib/libflacgpu.so ] && { cp build/alt_lib/libflacgpu.so /tmp/v_B.so; names="$names B"; }
for d in build/var_*; do [ -f $d/libflacgpu.so ] && { n=${d#build/var_}; cp $d/libflacgpu.so /tmp/v_$n.so; names="$names $n"; }; done
set -- $names -- "$@"
arr=(); while [ "$1" != "--" ]; do arr+=("$1"); shift; done; shift
n=${#arr[@]}
for i in $(seq $R); do
  for j in $(seq 0 $((n-1))); do
    v=${arr[$(((i + j) % n))]}
    cp /tmp/v_$v.so flac_amd/lib/libflacgpu.so
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['kernel_ms'])"
  done
done
cp /tmp/v_A.so flac_amd/lib/libflacgpu.so
