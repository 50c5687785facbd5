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
side FLACGPU_SHELL_BUDGET seconds (default 60) and holds what it got through by then to the reference's files -- a fifth of the
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
    budget = None if full else int(os.environ.get("FLACGPU_SHELL_BUDGET", "60"))
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


# ---- test/test_flac.sh: the tool's own suite --------------------------------------------------------------------------------------
FLAC_SH = os.path.join(REFDIR, "shell", "test", "test_flac.sh")
METAFLAC = os.path.join(REFDIR, "dropin", "metaflac")

# in front of the tool on PATH: runs it, then records the SHA-256 of every regular file of the working directory the call created
# or changed (what an encode wrote -- by name, by -o, or through the shell's redirection of -c --, and what a decode wrote), with the
# command line.  `--ogg` is refused like a tool built without Ogg support refuses it: the drop-in writes Ogg FLAC, but its decoder
# half is the reference's, which needs libogg (not in this image), and the script decodes what it encodes.
WRAPPER2 = """#!/bin/sh
case " $* " in *" --ogg "*) exit 1 ;; esac
marker="$HASH_LOG.marker"
: > "$marker"
"$REAL_FLAC" "$@"
rc=$?
for f in $(find . -maxdepth 1 -type f -cnewer "$marker" | sort); do
  echo "$(sha256sum < "$f" | cut -d' ' -f1) $f rc=$rc $*" >> "$HASH_LOG"
done
exit $rc
"""


def _run_flac_sh(tmp, which, libdir, inputs_from, budget=None):
    top = os.path.join(tmp, which)
    d = os.path.join(top, "test")
    os.makedirs(d)
    bindir = os.path.join(top, "bin")
    os.makedirs(bindir)
    staged = os.path.join(REFDIR, "shell", "test")
    shutil.copy(FLAC_SH, os.path.join(d, "test_flac.sh"))
    for sub in ("cuesheets", "foreign-metadata-test-files", "flac-to-flac-metadata-test-files"):
        shutil.copytree(os.path.join(staged, sub), os.path.join(d, sub))
    open(os.path.join(d, "common.sh"), "w").write(COMMON + "top_srcdir=%s\nECHO_N=-n\nECHO_C=\n" % top)
    w = os.path.join(bindir, "flac")
    open(w, "w").write(WRAPPER2)
    os.chmod(w, 0o755)
    os.symlink(GEN, os.path.join(bindir, "test_streams"))
    os.symlink(METAFLAC, os.path.join(bindir, "metaflac"))
    # the generator's streams (its noise is seeded by the clock): generated once by the caller, the same files for both runs -- the
    # script generates only when wacky1.wav is missing (:71-73)
    for f in os.listdir(inputs_from):
        shutil.copy2(os.path.join(inputs_from, f), d)
    env = dict(os.environ)
    env.update(PATH=bindir + ":" + env.get("PATH", ""), LD_LIBRARY_PATH=libdir + ":" + env.get("LD_LIBRARY_PATH", ""), REAL_FLAC=FLAC,
               HASH_LOG=os.path.join(top, "hashes.log"), FLAC__TEST_LEVEL=os.environ.get("FLACGPU_SHELL_TEST_LEVEL", "1"), top_srcdir=top)
    cmd = ["sh", "-e", "./test_flac.sh"]
    if budget:
        cmd = ["timeout", "-s", "TERM", str(budget)] + cmd
    r = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True, timeout=3000)
    log = open(env["HASH_LOG"]).read().replace(top, "$TOP").splitlines() if os.path.exists(env["HASH_LOG"]) else []
    return r, log, d



@pytest.mark.skipif(not (os.path.exists(FLAC_SH) and os.path.exists(METAFLAC)), reason="oracle/_ref/shell/test or dropin/metaflac not built")
def test_the_references_test_flac_sh_passes_on_the_drop_in_and_leaves_the_references_files(tmp_path):
    """test/test_flac.sh (:75-1392: overwrite protection, fractional block sizes, --skip / --until on encode and decode in every
    container the tool reads, --cue, --input-size, piped input with a fixed-up STREAMINFO, several files at once, foreign metadata
    round trips, FLAC-to-FLAC re-encoding with its metadata rules, --limit-min-bitrate, --ignore-chunk-sizes, over-long files) with
    `flac` = the reference's tool on this project's libFLAC.so.14, then on the reference's: exit status 0 both times, and every file a
    `flac` call wrote -- encodes and decodes alike -- the same bytes in both runs (second half).  1699 files from some
    1500 calls: 5.3 minutes on the GPU box (profiles/r06_ag_shell_test_flac_full.log); the default run gives the drop-in side
    FLACGPU_SHELL_BUDGET seconds (60) and compares what it wrote until then, FLACGPU_SHELL_SUITE=full runs it to its end."""
    tmp = str(tmp_path)
    full = os.environ.get("FLACGPU_SHELL_SUITE", "1") == "full"
    budget = None if full else int(os.environ.get("FLACGPU_SHELL_BUDGET", "60"))
    gen = os.path.join(tmp, "streams")
    os.makedirs(gen)
    subprocess.run([GEN], cwd=gen, check=True, capture_output=True, timeout=600)
    g, glog, gdir = _run_flac_sh(tmp, "gpu", os.path.join(ROOT, "flac_amd", "lib"), gen, budget=budget)
    assert g.returncode == 0 or (budget and g.returncode == 124), (g.returncode, g.stdout[-2500:], g.stderr[-1500:])
    r, rlog, _ = _run_flac_sh(tmp, "ref", os.path.join(REFDIR, "dropin"), gen)
    assert r.returncode == 0, (r.stdout[-2500:], r.stderr[-1500:])
    if g.returncode == 0:
        assert len(glog) == len(rlog) and len(glog) > 300, (len(glog), len(rlog))
    else:
        assert 50 < len(glog) <= len(rlog) + 2, (len(glog), len(rlog))
        glog = glog[:-2]                 # (the call that was under way when the budget ran out may have left a partial file)
    diff = [(a, b) for a, b in zip(glog, rlog) if a != b]
    assert not diff, (len(diff), diff[:3])
    print("test_flac.sh: %d of %d files written by flac calls%s, all equal; digest of the log %s" % (len(glog), len(rlog), "" if g.returncode == 0 else " (budget of %d s)" % budget,
                                                                                                hashlib.sha256("\n".join(glog).encode()).hexdigest()[:16]))
