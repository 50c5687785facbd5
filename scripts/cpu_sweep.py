import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, signals
from oracle import pyoracle as po
from concurrent.futures import ThreadPoolExecutor
pcm=signals.music(4096*512,2,16,seed=1234)
long_pcm=np.concatenate([pcm]*16,axis=0)
for t in (16,32,64,128):
    sec=min(po.ref_encode(long_pcm,16,44100,8,want_bytes=False,num_threads=t)["seconds"] for _ in range(2))
    print("pool",t,long_pcm.shape[0]/sec/1e6, flush=True)
for t in (64,128,256):
    t0=time.perf_counter()
    with ThreadPoolExecutor(t) as ex: list(ex.map(lambda _: po.ref_encode(pcm,16,44100,8,want_bytes=False)["seconds"], range(t*2)))
    print("indep",t, t*2*pcm.shape[0]/(time.perf_counter()-t0)/1e6, flush=True)
