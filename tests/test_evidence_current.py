"""CPU: the committed counter pass (profiles/pmc_traffic.json) was taken on THIS tree's kernel sources -- the git blob hashes recorded next
to the counters equal the files' (what bench.py reports as roofline.stale).  A tree whose kernels were edited after the last pass is
not a broken tree: the test then reports an expected failure naming the files, it does not fail the suite."""
import hashlib
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_counter_pass_is_of_this_trees_kernels():
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        pmc = json.load(f)
    hashes = pmc.get("source_hashes") or {}
    assert len(hashes) >= 15, "the counter pass records the kernel sources it was taken on"
    stale = []
    for rel, h in hashes.items():
        try:
            data = open(os.path.join(ROOT, rel), "rb").read()
        except OSError:
            stale.append(rel)
            continue
        if hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest() != h:
            stale.append(rel)
    if stale:
        pytest.xfail("kernel sources edited since the counter pass %s: %s (bench.py will say stale: true)" % (pmc.get("tag"), ", ".join(sorted(stale))))
    # every workload the bench line quotes has its kernels in the pass
    for w in ("level8", "level5", "level0", "white8", "hires8"):
        assert w in pmc["workloads"] and pmc["workloads"][w]["kernels"], w
