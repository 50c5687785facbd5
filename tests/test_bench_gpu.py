"""-m gpu: bench.py's multi-rank code path at world size 1 (--force-dist: process group on the nccl backend, windowed gather, the
check of every rank's frames) in its three gather modes; 8-GPU runs are the driver's, this keeps the line it will print honest."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gather", ["rccl", "hostshm", "none"])
def test_multi_rank_line_checks_every_rank(gather):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--gather", gather, "--frames", "768", "--steps", "6", "--warmup", "1",
                        "--window", "4", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    v = line["verified"]
    assert v["ranks_checked"] == 1 and v["ok"], v
    assert v["crc16_frames_checked"] == 768 and v["frames_compared_with_oracle"] >= 16
    assert line["gather"]["mode"] == gather and line["n_gpus"] == 1
    if gather != "none":
        assert line["gather"]["bytes_gathered_last_step"] > 0
