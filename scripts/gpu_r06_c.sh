#!/bin/bash
# one visit: a bench line with the decode_only side figure, and a kernel trace of the decode
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench.json")); print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("decode_only"), indent=1)); print(d.get("device_verify"))
except Exception as e:
    print("no line", e); print(open("$OUT/bench.err").read()[-3000:])
PY
