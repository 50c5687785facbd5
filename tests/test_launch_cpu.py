"""CPU: `--gpus N` starts N ranks by itself (bench.py parsed --gpus and never read it -- `python bench.py --gpus 8`
printed a 1-GPU line).  flac_amd.dist.ensure_ranks is the one place that decides; bench.py and flac_amd.corpus call it first thing.
The reference's analogue: FLAC__stream_encoder_set_num_threads -- one call, N workers, frames out in order
(stream_encoder.c:2151, :3530-3574)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE = os.path.join(ROOT, "tests", "launch_probe.py")


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}


def test_launch_command_is_the_drivers():
    from flac_amd.dist import launch_command
    cmd = launch_command(8, "bench.py", ["--gpus", "8", "--steps", "20"], port=29501, python="python")
    assert cmd == ["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29501",
                   "bench.py", "--gpus", "8", "--steps", "20"]


def test_under_a_launcher_the_world_size_must_be_the_one_asked_for():
    from flac_amd.dist import ensure_ranks, EXIT_BAD_WORLD
    assert ensure_ranks(4, "x.py", [], env={"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2"}) == (2, 2, 4)
    assert ensure_ranks(None, "x.py", [], env={"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"}) == (3, 3, 4)
    with pytest.raises(SystemExit) as e:
        ensure_ranks(8, "x.py", [], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert e.value.code == EXIT_BAD_WORLD
    with pytest.raises(SystemExit) as e:
        ensure_ranks(1, "x.py", [], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert e.value.code == EXIT_BAD_WORLD


def test_one_rank_needs_no_launcher_and_too_few_devices_refuse():
    from flac_amd.dist import ensure_ranks, EXIT_BAD_WORLD, visible_devices
    assert ensure_ranks(None, "x.py", [], env={}) == (0, 0, 1)
    assert ensure_ranks(1, "x.py", [], env={}) == (0, 0, 1)
    if visible_devices() < 8:                 # (this container: none)
        with pytest.raises(SystemExit) as e:
            ensure_ranks(8, "x.py", [], env={}, _exec=lambda cmd, env: pytest.fail("must not launch"))
        assert e.value.code == EXIT_BAD_WORLD


def test_not_under_a_launcher_the_job_is_replaced_by_its_launcher():
    from flac_amd.dist import ensure_ranks
    seen = {}

    def fake_exec(cmd, env):
        seen["cmd"], seen["env"] = cmd, env
        return "launched"
    assert ensure_ranks(4, "/x/bench.py", ["--gpus", "4", "--steps", "3"], need_devices=False, env={"PATH": "/bin"}, _exec=fake_exec) == "launched"
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "WORLD_SIZE" not in seen["env"]


@pytest.mark.parametrize("world", [2, 4])
def test_a_job_started_plainly_runs_as_n_ranks(world):
    """the real thing over gloo: `python launch_probe.py --gpus N`, no launcher in front"""
    r = subprocess.run([sys.executable, PROBE, "--gpus", str(world)], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                     # ONE line, rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["world_size_seen"] == world and line["self_launched"]
    assert line["ranks"] == list(range(world)) and line["local_ranks"] == list(range(world))
    assert line["shards"][0][0] == 0 and line["shards"][-1][1] == 1000
    assert all(line["shards"][r][1] == line["shards"][r + 1][0] for r in range(world - 1))


def test_a_job_under_the_drivers_launcher_with_the_wrong_n_fails():
    """torch.distributed.run --nproc-per-node 2 ... --gpus 4: every rank refuses (exit code 3), no line is printed"""
    from flac_amd.dist import launch_command
    r = subprocess.run(launch_command(2, PROBE, ["--gpus", "4"]), env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "launcher started 2 ranks" in r.stderr


def test_plain_start_without_enough_devices_refuses():
    r = subprocess.run([sys.executable, PROBE, "--gpus", "64", "--need-devices"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "HIP device(s) visible" in r.stderr and not r.stdout.strip()


def test_bench_and_corpus_decide_through_ensure_ranks():
    """the two jobs' command lines reach ensure_ranks before anything else happens (bench.py needs a GPU to go further)"""
    for rel in ("bench.py", os.path.join("flac_amd", "corpus.py")):
        src = open(os.path.join(ROOT, rel)).read()
        assert "ensure_ranks(" in src, rel
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "--gpus 64" in r.stderr and not r.stdout.strip()


def test_forced_launch_of_one_rank():
    """FLAC_AMD_FORCE_LAUNCH=1 (the GPU test of the exec path on a one-GPU box): one rank still goes through the launcher; the
    variable does not travel to the ranks"""
    from flac_amd.dist import ensure_ranks
    seen = {}
    assert ensure_ranks(1, "/x/bench.py", ["--gpus", "1"], need_devices=False, env={"FLAC_AMD_FORCE_LAUNCH": "1"}, _exec=lambda c, e: seen.update(cmd=c, env=e) or "launched") == "launched"
    assert seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "1" and "FLAC_AMD_FORCE_LAUNCH" not in seen["env"]
    assert ensure_ranks(None, "/x/bench.py", [], env={"FLAC_AMD_FORCE_LAUNCH": "1"}) == (0, 0, 1)          # (no --gpus: nothing to launch)
    r = subprocess.run([sys.executable, PROBE, "--gpus", "1"], env=dict(_clean_env(), FLAC_AMD_FORCE_LAUNCH="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1


WATCHDOG_SCRIPT = r"""
import os, sys, threading, time, json
sys.path.insert(0, %r)
from flac_amd.dist import Watchdog
mode = sys.argv[1]
def bail(why):
    os.write(1, (json.dumps({"value": 123.0, "side_figures": {"error": why}}) + "\n").encode())
    os._exit(1)
with Watchdog(1.5 if mode == "timeout" else 30, bail, "the side figures"):
    if mode == "ok":
        time.sleep(0.2)
    elif mode == "timeout":
        lock = threading.Lock(); lock.acquire(); lock.acquire()          # a main thread that never comes back
    elif mode == "sigterm":
        os.write(1, b"READY\n")
        lock = threading.Lock(); lock.acquire(); lock.acquire()
if mode == "ok":
    import signal
    assert signal.getsignal(signal.SIGTERM) == signal.SIG_DFL
    print(json.dumps({"value": 123.0, "side": "measured"}))
"""


@pytest.mark.parametrize("mode", ["ok", "timeout", "sigterm"])
def test_the_main_line_survives_side_figures_that_hang_or_a_rank_that_dies(mode, tmp_path):
    """bench.py, multi-rank: the encode-only / hostshm figures are measured behind the main line's measurement.  If they hang, or
    another rank dies and the launcher sends SIGTERM, rank 0 -- possibly sitting in a collective that never returns, where no
    Python signal handler runs -- must still print the main line (flac_amd.dist.Watchdog: a thread on the interpreter's wake-up
    descriptor) and exit non-zero."""
    import signal
    import time
    script = tmp_path / "w.py"
    script.write_text(WATCHDOG_SCRIPT % ROOT)
    p = subprocess.Popen([sys.executable, str(script), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_clean_env())
    if mode == "sigterm":
        assert p.stdout.readline().strip() == b"READY"
        time.sleep(0.3)
        p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    lines = [json.loads(l) for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["value"] == 123.0, (out, err)
    if mode == "ok":
        assert p.returncode == 0 and lines[0]["side"] == "measured"
    else:
        assert p.returncode == 1
        why = lines[0]["side_figures"]["error"]
        assert ("did not finish within 1 s" in why) if mode == "timeout" else ("signal 15" in why), why
