"""Generates tests/golden/frames.json from the UNMODIFIED reference (oracle/_ref/libFLAC_ref.so).

Run in the build container (where /root/reference exists):  python tests/golden/make_golden.py
Each entry: sha256 of the audio frames (everything after the metadata blocks), frame count, byte count,
plus the reference's vendor string and compiler so the pin can be traced."""
import hashlib, json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
from oracle import pyoracle as po
from cases import golden_cases, case_key, case_pcm, case_search

def main():
    po.build(ref=True)
    out = {"_meta": {"vendor": po.load_ref().ref_vendor_string().decode(),
                     "compiler": subprocess.check_output(["gcc", "--version"]).decode().splitlines()[0],
                     "flags": "-O3 -DNDEBUG -fassociative-math -fno-signed-zeros -fno-trapping-math -freciprocal-math; FMA+AVX2 dispatch"}}
    for c in golden_cases():
        r = po.ref_encode(case_pcm(c), c["bps"], c["rate"], c["level"], **case_search(c))
        frames = r["data"][r["header_bytes"]:]
        out[case_key(c)] = {"sha256": hashlib.sha256(frames).hexdigest(), "frames": int(len(r["frame_bytes"])), "bytes": len(frames)}
    with open(os.path.join(HERE, "frames.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote %d golden entries" % (len(out) - 1))

if __name__ == "__main__":
    main()
