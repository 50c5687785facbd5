#!/bin/bash
# where a tiny `flac` invocation on the drop-in spends its time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_k; cd gpurun_out/r06_k
python - <<'PY'
import numpy as np
np.random.default_rng(1).integers(-100,100,size=20000,dtype=np.int16).tofile('t.raw')
PY
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/flac_amd/lib
F=$GRAFT_REPO_ROOT/oracle/_ref/dropin/flac
A="--silent --force --force-raw-format --endian=little --sign=signed --sample-rate=44100 --bps=16 --channels=1 -5 t.raw"
$F $A; 
time (for i in 1 2 3 4 5 6 7 8 9 10; do $F $A; done)
FLACGPU_HOST_TIMING=1 $F $A 2>&1 | tail -5
strace -c -f $F $A 2>&1 | tail -25
time (for i in 1 2 3 4 5 6 7 8 9 10; do python -c "import ctypes; l=ctypes.CDLL('$GRAFT_REPO_ROOT/flac_amd/lib/libflacgpu.so'); l.flacgpu_device_count()"; done)
