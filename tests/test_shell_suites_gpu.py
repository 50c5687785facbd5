"""-m gpu: the reference's OWN shell suite, test/test_streams.sh (:52-331: every stream of its generator through `flac --verify ...`
under its matrix of presets, -e / -p, block sizes 16..33 x LPC orders 0..32, subdivide_tukey(32), --disable-*-subframes, header
variations, 1..2 channels at 8..32 bits, then decoded again and compared with the input; and its corruption handling), run twice in
two directories with the SAME generated input files:
  * `flac` = the reference's tool on this project's library (oracle/_ref/dropin/flac with LD_LIBRARY_PATH=flac_amd/lib: libFLAC.so.14
    whose encoder is the HIP engine),
  * `flac` = the same binary on the reference's library (LD_LIBRARY_PATH=oracle/_ref/dropin).
Both must end with exit status 0, and every .flac either run writes -- one per invocation, logged by a wrapper in front of the tool --
must be the same bytes in both.  The script itself is the reference's, staged by `make -C oracle shell` into
oracle/_ref/shell (a build output; the GPU box has no /root/reference); `common.sh`, which the reference generates at configure time,
is written here.
The script is 1228 encodes at FLAC__TEST_LEVEL=0, each a process that starts the HIP runtime (0.2 s) and an engine: 8.6 minutes on the
GPU box (profiles/r06_j_shell_suite_full.log: exit status 0, 1228 files equal).  The suite's default run therefore gives the drop-in
side FLACGPU_SHELL_BUDGET seconds (default 100) and holds what it got through by then to the reference's files -- a third of the
script; FLACGPU_SHELL_SUITE=full runs it to its end and demands exit status 0; FLACGPU_SHELL_SUITE=0 skips."""
import hashlib
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
SCRIPT = os.path.join(REFDIR, "shell", "test_streams.sh")
FLAC = os.path.join(REFDIR, "dropin", "flac")
GEN = os.path.join(REFDIR, "test_streams")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(SCRIPT) and os.path.exists(FLAC) and os.path.exists(GEN)), reason="oracle/_ref/shell, dropin or test_streams not built"),
              pytest.mark.skipif(os.environ.get("FLACGPU_SHELL_SUITE", "1") == "0", reason="FLACGPU_SHELL_SUITE=0")]

COMMON = """# written by tests/test_shell_suites_gpu.py (the reference generates this file from common.sh.in at configure time)
EXE=
is_win=no
SILENT='--silent'
TOTALLY_SILENT='--totally-silent'
die ()
{
	echo $* 1>&2
	exit 1
}
"""

# in front of the tool on PATH: runs it, then records the SHA-256 of the .flac an ENCODE left (decodes leave none to record)
WRAPPER = """#!/bin/sh
"$REAL_FLAC" "$@"
rc=$?
case " $* " in
  *" --decode "*|*" -d "*) ;;
  *) for a in "$@"; do last="$a"; done
     case "$last" in
       *.raw) f="${last%.raw}.flac"; [ -f "$f" ] && echo "$(sha256sum < "$f" | cut -d' ' -f1) $*" >> "$HASH_LOG" ;;
     esac ;;
esac
exit $rc
"""


def _run(tmp, which, libdir, inputs_from=None, budget=None):
    d = os.path.join(tmp, which, "test")
    os.makedirs(d)
    bindir = os.path.join(tmp, which, "bin")
    os.makedirs(bindir)
    shutil.copy(SCRIPT, os.path.join(d, "test_streams.sh"))
    open(os.path.join(d, "common.sh"), "w").write(COMMON)
    w = os.path.join(bindir, "flac")
    open(w, "w").write(WRAPPER)
    os.chmod(w, 0o755)
    os.symlink(GEN, os.path.join(bindir, "test_streams"))
    if inputs_from:
        for f in os.listdir(inputs_from):
            if f.endswith((".raw", ".wav", ".aiff", ".aifc", ".w64", ".rf64")) and not f.endswith(".cmp"):
                os.link(os.path.join(inputs_from, f), os.path.join(d, f)) if os.stat(inputs_from).st_dev == os.stat(d).st_dev else shutil.copy(os.path.join(inputs_from, f), d)
    env = dict(os.environ)
    env.update(PATH=bindir + ":" + env.get("PATH", ""), LD_LIBRARY_PATH=libdir + ":" + env.get("LD_LIBRARY_PATH", ""), REAL_FLAC=FLAC,
               HASH_LOG=os.path.join(tmp, which, "hashes.log"), FLAC__TEST_LEVEL=os.environ.get("FLACGPU_SHELL_TEST_LEVEL", "0"))
    cmd = ["sh", "-e", "./test_streams.sh"]
    if budget:
        cmd = ["timeout", "-s", "TERM", str(budget)] + cmd
    r = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True, timeout=3000)
    log = open(env["HASH_LOG"]).read().splitlines() if os.path.exists(env["HASH_LOG"]) else []
    return r, log, d


def test_the_references_test_streams_sh_passes_on_the_drop_in_and_leaves_the_references_files(tmp_path):
    tmp = str(tmp_path)
    full = os.environ.get("FLACGPU_SHELL_SUITE", "1") == "full"
    budget = None if full else int(os.environ.get("FLACGPU_SHELL_BUDGET", "100"))
    g, glog, gdir = _run(tmp, "gpu", os.path.join(ROOT, "flac_amd", "lib"), budget=budget)
    # (124: the budget ran out -- what was encoded until then is compared; anything else must be the script's own success)
    assert g.returncode == 0 or (budget and g.returncode == 124), (g.returncode, g.stdout[-1500:], g.stderr[-1500:])
    assert "ERROR" not in g.stdout and "ERROR" not in g.stderr, (g.stdout[-1500:], g.stderr[-1500:])
    r, rlog, _ = _run(tmp, "ref", os.path.join(REFDIR, "dropin"), inputs_from=gdir)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    if g.returncode == 0:
        assert len(glog) == len(rlog) and len(glog) > 500, (len(glog), len(rlog))
    else:
        # the encode that was under way when the budget ran out may have been logged or not: its line is compared if it is there
        assert 100 < len(glog) <= len(rlog), (len(glog), len(rlog))
    diff = [(a, b) for a, b in zip(glog, rlog) if a != b]
    assert not diff, (len(diff), diff[:3])
    # (the wrapper's log carries one line per encode: the digest of what it wrote and its command line)
    print("test_streams.sh: %d of %d encodes%s, all files equal; digest of the log %s" % (len(glog), len(rlog), "" if g.returncode == 0 else " (budget of %d s)" % budget,
                                                                                       hashlib.sha256("\n".join(glog).encode()).hexdigest()[:16]))
