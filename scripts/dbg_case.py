"""debug aid: encode one case on the GPU and with the oracle, print the per-subframe choices of the first differing frame.
usage: dbg_case.py family n channels bps rate level [key=value ...]"""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import ctypes as C
import numpy as np, signals, flac_amd
from oracle import pyoracle as po
fam, n, ch, bps, rate, level = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[7:])}
pcm = signals.FAMILIES[fam](n, ch, bps)
s = flac_amd.make_settings(ch, bps, rate, level, streamable_subset=0, **kw)
eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=64)
data, fb = eng.encode(pcm)
nf = len(fb)
sub, ca = eng.last_batch_info(nf)
okw = dict(kw)
if "loose_mid_side" in okw: okw["loose"] = okw.pop("loose_mid_side")
o = po.oracle_encode(pcm, bps, rate, level, **okw)
print("frame bytes gpu", fb, "oracle", o["frame_bytes"])
lib = po.load_oracle()
N = eng.blocksize
for f in range(nf):
    if f < len(o["frame_bytes"]) and fb[f] == o["frame_bytes"][f]:
        continue
    blk = pcm[f * N:(f + 1) * N]
    cfg = po.OracleConfig(ch, bps, rate, level, **dict(okw, blocksize=len(blk), stream_blocksize=N))
    planar = np.ascontiguousarray(blk.T)
    ptrs = (C.c_void_p * ch)(*[planar[i].ctypes.data for i in range(ch)])
    out = np.empty(len(blk) * ch * 5 + 4096, dtype=np.uint8)
    info = po.FoFrameInfo()
    r = lib.fo_encode_frame(C.byref(cfg.c), ptrs, f, out.ctypes.data, len(out), C.byref(info))
    print("frame", f, "oracle bytes", r, "ca", info.channel_assignment, "gpu ca", ca[f])
    for c in range(ch):
        x = sub[f * ch + c]; y = info.sub[c]
        print("  gpu    sub", c, "type", x.type, "order", x.order, "wasted", x.wasted_bits, "po", x.partition_order, "prec", x.precision, "shift", x.shift, "bits", x.bits)
        print("  oracle sub", c, "type", y.type, "order", y.order, "wasted", y.wasted_bits, "po", y.partition_order, "prec", y.precision, "shift", y.shift, "bits", y.bits)
    break
