import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, signals, flac_amd
from oracle import pyoracle as po
fam = sys.argv[1]; bs = int(sys.argv[2]); ch = int(sys.argv[3]); bps = int(sys.argv[4]); level = int(sys.argv[5])
kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[6:])}
pcm = signals.FAMILIES[fam](bs + 777, ch, bps)
s = flac_amd.make_settings(ch, bps, 44100, level, blocksize=bs, streamable_subset=0, **kw)
eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=2)
print("created", flush=True)
data, fb = eng.encode(pcm)
o = po.oracle_encode(pcm, bps, 44100, level, blocksize=bs, **kw)
print(fb, o["frame_bytes"], data == o["data"])
