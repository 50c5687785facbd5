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
    assert line["gather"]["world_size_seen"] == 1
    if gather != "none":
        assert line["gather"]["bytes_gathered_last_step"] > 0
    if gather == "rccl":
        # the rccl line carries the encode-only and the hostshm figure of the same run, every rank's frames checked in both
        for key in ("encode_only", "hostshm"):
            side = line[key]
            assert side["value"] > 0 and side["verified"]["ok"] and side["verified"]["ranks_checked"] == 1, (key, side)
    else:
        assert "encode_only" not in line


@pytest.mark.parametrize("fault", ["raise", "hang"])
def test_the_main_line_is_printed_when_the_side_figures_fail(fault):
    """the multi-rank rccl line measures the encode-only and hostshm figures behind the main measurement: an exception there, or a
    hang (another rank gone, a collective that never completes), must not cost the finished, verified main line -- rank 0 prints it
    with the reason and exits 1 (flac_amd.dist.Watchdog)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29650 + os.getpid() % 300), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", FLACGPU_BENCH_SIDE_FAULT=fault)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--frames", "512", "--steps", "4", "--warmup", "1", "--window", "2",
                        "--no-cpu-baseline", "--no-extras", "--side-timeout", "8"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 1, (r.returncode, r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] > 0 and line["n_gpus"] == 1 and line["verified"]["ok"] and line["verified"]["ranks_checked"] == 1
    why = line["side_figures"]["error"]
    assert ("injected fault" in why) if fault == "raise" else ("did not finish within 8 s" in why), why
    assert "encode_only" not in line


def _run_bench(argv, env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=600)
    return r


def test_gpus_flag_and_the_launchers_world_size_agree():
    """`bench.py --gpus 1` started plainly and `WORLD_SIZE=1 ... --gpus 1 --force-dist` print the same n_gpus; a --gpus that the
    launcher's world size contradicts, or that this box has no devices for, ends with exit code 3 and no line (--gpus
    was parsed and never read)."""
    import torch
    clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    small = ["--frames", "512", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    a = _run_bench(["--gpus", "1"] + small, clean)
    assert a.returncode == 0, a.stderr[-3000:]
    la = json.loads(a.stdout.strip().splitlines()[-1])
    env = dict(clean, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29300 + os.getpid() % 300), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    b = _run_bench(["--gpus", "1", "--force-dist", "--no-side-gathers"] + small, env)
    assert b.returncode == 0, b.stderr[-3000:]
    lb = json.loads(b.stdout.strip().splitlines()[-1])
    assert la["n_gpus"] == lb["n_gpus"] == 1 and lb["gather"]["world_size_seen"] == 1 and lb["verified"]["ranks_checked"] == 1
    c = _run_bench(["--gpus", "2", "--force-dist"] + small, env)                       # the launcher says 1
    assert c.returncode == 3 and not c.stdout.strip(), (c.returncode, c.stdout[-300:])
    n = torch.cuda.device_count()
    d = _run_bench(["--gpus", str(n + 1)] + small, clean)                             # no launcher, too few devices
    assert d.returncode == 3 and not d.stdout.strip(), (d.returncode, d.stdout[-300:])


def test_self_launched_ranks_on_the_devices_there_are():
    """`python bench.py --gpus N` with N = every device of this box (1 on the test boxes: then the plain path; more: bench.py
    replaces itself by torch.distributed.run and rank 0 prints the one line, n_gpus == N, every rank checked)"""
    import torch
    n = torch.cuda.device_count()
    clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = _run_bench(["--gpus", str(n), "--frames", "512", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], clean)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == n
    if n > 1:
        assert line["gather"]["world_size_seen"] == n and line["verified"]["ranks_checked"] == n and line["verified"]["ok"]
        assert line["encode_only"]["value"] > 0 and line["encode_only"]["verified"]["ranks_checked"] == n


def test_the_self_launch_path_on_this_box():
    """FLAC_AMD_FORCE_LAUNCH=1: `python bench.py --gpus 1` replaces itself by `python -m torch.distributed.run --nproc-per-node 1 ...`
    even for one rank -- the exec, the launcher's environment, the process group over RCCL and rank 0's ONE line on the command's
    stdout, end to end on hardware (N > 1 needs a box with N GPUs: the driver's run)"""
    clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = _run_bench(["--gpus", "1", "--force-dist", "--frames", "768", "--steps", "6", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], dict(clean, FLAC_AMD_FORCE_LAUNCH="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and "flac_amd.dist.ensure_ranks" in line["launched_by"], line["launched_by"]
    assert line["gather"]["world_size_seen"] == 1 and line["verified"]["ok"] and line["verified"]["ranks_checked"] == 1
    assert line["encode_only"]["value"] > 0 and line["hostshm"]["value"] > 0
