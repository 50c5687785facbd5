#!/bin/bash
TAG=$1
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench.json")); print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("libflac_api"), indent=1)); print(json.dumps(d.get("decode_only")))
except Exception as e:
    print("no line", e); print(open("$OUT/bench.err").read()[-3000:])
PY
