"""Quick GPU-vs-oracle sweep (development aid; the real tests live in tests/)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po
import signals
import flac_amd

def check(name, pcm, bps, rate, level, **kw):
    s = flac_amd.make_settings(pcm.shape[1], bps, rate, level, **kw)
    eng = flac_amd.FrameEngine(s, max_batch_frames=512)
    t0 = time.time()
    data, fb = eng.encode(pcm)
    dt = time.time() - t0
    okw = {}
    if 'limit_min_bitrate' in kw: okw['limit_min_bitrate'] = kw['limit_min_bitrate']
    if 'blocksize' in kw: okw['blocksize'] = kw['blocksize']
    o = po.oracle_encode(pcm, bps, rate, level, **okw)
    ok = data == o['data']
    bad = []
    if not ok:
        offg = np.concatenate([[0], np.cumsum(fb.astype(np.int64))]); offo = np.concatenate([[0], np.cumsum(o['frame_bytes'].astype(np.int64))])
        for i in range(min(len(fb), len(o['frame_bytes']))):
            if data[offg[i]:offg[i+1]] != o['data'][offo[i]:offo[i+1]]: bad.append(i)
    print(f"{name:12s} L{level} bps{bps} ch{pcm.shape[1]} frames={len(fb)} {'OK' if ok else 'MISMATCH '+str(bad[:8])+' n='+str(len(bad))} ({dt*1e3:.0f} ms)", flush=True)
    eng.close()
    return ok

if __name__ == "__main__":
    allok = True
    n = 4096 * 6 + 1000
    levels = [int(a) for a in sys.argv[1:]] or list(range(9))
    for fam in ['music', 'white', 'sine', 'constant', 'silence', 'wasted', 'square', 'quiet', 'mixed']:
        for level in levels:
            allok &= check(fam, signals.FAMILIES[fam](n, 2, 16), 16, 44100, level)
    for level in levels:
        allok &= check("music24", signals.music(4096*3+123, 2, 24), 24, 96000, level)
        allok &= check("mono", signals.music(4096*3+77, 1, 16), 16, 44100, level)
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
