import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # in-tree build products are git-ignored; (re)build what is missing. hipcc cross-compiles without a GPU.
    from oracle import pyoracle as po
    if not os.path.exists(po.ORACLE_SO):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    if not po.have_ref() and os.path.isdir("/root/reference/src/libFLAC"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref", "-j8"])
    lib = os.path.join(ROOT, "flac_amd", "lib")
    if not (os.path.exists(os.path.join(lib, "libflacgpu.so")) and os.path.exists(os.path.join(lib, "libFLACgpu.so"))):
        if os.path.exists("/opt/rocm/bin/hipcc"):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "flac_amd", "csrc")])


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """Load torch's HIP runtime before libflacgpu.so loads one.  The torch wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7, asked for by libtorch_hip as plain "libamdhip64.so"); libflacgpu.so asks for libamdhip64.so.7.  With torch
    first, the loader satisfies our request with torch's copy -- one runtime in the process.  With ours first (the system's
    /opt/rocm copy), torch's request does not match the loaded SONAME and a SECOND runtime is mapped; its device enumeration
    then answered "No HIP GPUs are available" on some boxes (seen when a test file was run on its own, never in a full run,
    where an earlier test imports torch first).  bench.py and flac_amd/corpus.py import torch at the top for the same reason."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:            # no torch, no GPU: the CPU tests do not care
        pass
    yield


@pytest.fixture(scope="session")
def ref():
    from oracle import pyoracle as po
    if not po.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    return po.load_ref()
