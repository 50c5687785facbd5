import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, signals, flac_amd
from oracle import pyoracle as po
order = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ex = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pcm = signals.FAMILIES["square"](1152 * 2 + 301, 2, 16)
s = flac_amd.make_settings(2, 16, 44100, 0, max_lpc_order=order, exhaustive=ex, prec_search=ps, mid_side=1, loose_mid_side=0, streamable_subset=0)
eng = flac_amd.FrameEngine(s, device=0, max_batch_frames=8)
data, fb = eng.encode(pcm)
sub, ca = eng.last_batch_info(1)
for c in range(2):
    x = sub[c]
    print("gpu sub", c, "type", x.type, "order", x.order, "wasted", x.wasted_bits, "po", x.partition_order, "prec", x.precision, "shift", x.shift, "bits", x.bits)
print("ca", ca)
o = po.oracle_encode(pcm, 16, 44100, 0, max_lpc_order=order, exhaustive=ex, prec_search=ps, mid_side=1, loose=0)
print(len(data), len(o["data"]), data == o["data"])
d = np.frombuffer(data, np.uint8); e = np.frombuffer(o["data"], np.uint8)
m = min(len(d), len(e)); idx = np.nonzero(d[:m] != e[:m])[0]
print("ndiff", len(idx), idx[:20])
print(bytes(d[:48]).hex()); print(bytes(e[:48]).hex())
