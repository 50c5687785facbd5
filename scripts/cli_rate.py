#!/usr/bin/env python3
"""End-to-end rate of the reference's `flac` tool (oracle/_ref/flac_cli_*): the same 30-minute 16-bit stereo WAVE through the
tool on the reference library (1 thread, and -j 8) and through the same tool on libFLACgpu.so.  Wall time of the whole
command: process start, file read, WAVE parsing, MD5, encode, file write."""
import os
import subprocess
import sys
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flac_amd import signals  # noqa: E402

minutes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
base = signals.music(44100 * 60, 2, 16, seed=11)
path = "/tmp/cli_rate.wav"
w = wave.open(path, "wb")
w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100)
for m in range(minutes):
    w.writeframes(np.clip(base * (1.0 - 0.01 * (m % 7)), -32768, 32767).astype("<i2").tobytes())
w.close()
n = 44100 * 60 * minutes
ref, gpu = os.path.join(ROOT, "oracle", "_ref", "flac_cli_ref"), os.path.join(ROOT, "oracle", "_ref", "flac_cli_gpu")
outs = {}
for name, cmd in (("reference, 1 thread", [ref, "-8"]), ("reference, -j 8", [ref, "-8", "-j", "8"]), ("libFLACgpu", [gpu, "-8"]), ("libFLACgpu (again)", [gpu, "-8"])):
    out = "/tmp/cli_rate_%s.flac" % ("gpu" if "gpu" in name.lower() else "ref")
    t0 = time.perf_counter()
    r = subprocess.run(cmd + ["-s", "-f", "-o", out, path], capture_output=True, text=True, env=dict(os.environ, FLACGPU_HOST_TIMING="1"))
    dt = time.perf_counter() - t0
    for line in r.stderr.splitlines():
        if "timing" in line:
            print("    " + line)
    if r.returncode != 0:
        print("%-22s failed: %s" % (name, r.stderr[-300:]))
        continue
    outs[name] = open(out, "rb").read()
    print("%-22s %6.2f s  %7.1f M samples/s" % (name, dt, n / dt / 1e6))
print("files identical:", outs.get("reference, 1 thread") == outs.get("libFLACgpu"))

# several files in one command: one process, one encoder after another (the second and third get the first one's engine)
import shutil
paths = []
for i in range(3):
    q = "/tmp/cli_rate_%d.wav" % i
    shutil.copy(path, q)
    paths.append(q)
for name, cmd in (("reference -j 8, 3 files", [ref, "-8", "-j", "8"]), ("libFLACgpu, 3 files", [gpu, "-8"])):
    t0 = time.perf_counter()
    r = subprocess.run(cmd + ["-s", "-f"] + paths, capture_output=True, text=True, env=dict(os.environ, FLACGPU_HOST_TIMING="1"))
    dt = time.perf_counter() - t0
    for line in r.stderr.splitlines():
        if "timing" in line:
            print("    " + line)
    print("%-24s %6.2f s  %7.1f M samples/s  rc=%d" % (name, dt, 3 * n / dt / 1e6, r.returncode))
    outs[name] = [open(q[:-4] + ".flac", "rb").read() for q in paths]
print("3-file outputs identical:", outs.get("reference -j 8, 3 files") == outs.get("libFLACgpu, 3 files"))
