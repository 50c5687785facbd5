#!/usr/bin/env python3
"""Same-box A/B of two builds of libflacgpu.so: bench.py's -8 step (and optionally the 96 kHz / 24-bit and -5 steps) alternately with
FLACGPU_ENGINE_SO pointing at build A and build B, `rounds` times each, per-kernel HIP-event times side by side.

    python scripts/ab_engine.py flac_amd/lib/libflacgpu_prev.so flac_amd/lib/libflacgpu.so [rounds] [extra bench.py flags ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:]
res = {a: [], b: []}
for r in range(rounds):
    for lib in (a, b):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras"] + extra,
                             env=dict(os.environ, FLACGPU_ENGINE_SO=lib), capture_output=True, text=True)
        if out.returncode != 0:
            print(lib, "FAILED", out.stderr[-2000:])
            sys.exit(1)
        line = json.loads(out.stdout.strip().splitlines()[-1])
        res[lib].append(line)
        print("%-40s round %d: %8.1f M samples/s  %s  verified %s" % (os.path.basename(lib), r, line["value"], {k: round(v, 4) for k, v in line["kernel_ms"].items()}, line.get("verified", {}).get("ok")))
for lib in (a, b):
    ls = res[lib]
    best = max(ls, key=lambda l: l["value"])
    mean = {k: sum(l["kernel_ms"][k] for l in ls) / len(ls) for k in ls[0]["kernel_ms"]}
    print("%-40s best %8.1f  mean %8.1f M samples/s; mean kernel ms %s" % (os.path.basename(lib), best["value"], sum(l["value"] for l in ls) / len(ls), {k: round(v, 4) for k, v in mean.items()}))
