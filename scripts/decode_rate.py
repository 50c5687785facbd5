"""The device stream decoder's rate on streams of growing length (scripts: development aid; the bench line carries the 16384-frame
figure): a -8 stream of N frames x 4096 x 2 ch encoded by the engine in batches, taken as a bare stream and decoded
(flacgpu_decode_stream_device), PCM compared with the input.  usage: python scripts/decode_rate.py [frames ...]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flac_amd                                    # noqa: E402
from flac_amd.stream_decoder import StreamDecoder, StreamInfo   # noqa: E402
from flac_amd import signals                       # noqa: E402

LEVEL = int(os.environ.get("LEVEL", "8"))
dev = torch.device("cuda", 0)
B = 16384
block = 4096 if LEVEL >= 3 else 1152
settings = flac_amd.make_settings(2, 16, 44100, LEVEL)
eng = flac_amd.FrameEngine(settings, device=0, max_batch_frames=B)
cap = eng.max_output_bytes(B)
sdec = StreamDecoder(0)
rows = []
for nframes in [int(a) for a in sys.argv[1:]] or [16384, 65536, 262144]:
    pcm = signals.music(B * block, 2, 16, seed=11)
    d_pcm1 = torch.from_numpy(pcm).to(dev)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_fb = torch.empty(B, dtype=torch.int32, device=dev)
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    parts = []
    for b0 in range(0, nframes, B):
        nb = min(B, nframes - b0)
        eng.encode_device(d_pcm1.data_ptr(), nb, d_out.data_ptr(), cap, d_fb.data_ptr(), d_total.data_ptr(), first_frame_number=b0, stream=0)
        torch.cuda.synchronize()
        parts.append(d_out[:int(d_total.item())].clone())
    d_stream = torch.cat(parts + [torch.zeros(64, dtype=torch.uint8, device=dev)])
    nbytes = d_stream.numel() - 64
    del parts
    info = StreamInfo(1, block, block, 44100, 2, 16)
    d_dec = torch.empty(nframes * block * 2, dtype=torch.int32, device=dev)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        rc, r, ev = sdec.decode_device(d_stream.data_ptr(), nbytes, 0, info, d_dec.data_ptr(), d_dec.numel())
        torch.cuda.synchronize()
        w = (time.perf_counter() - t0) * 1e3
        if best is None or w < best[0]:
            best = (w, r.ms_scan, r.ms_decode, r.ms_place, r.ms_total)
    ok = rc == 0 and r.nevents == 0 and int(r.samples) == nframes * block
    v = d_dec.view(-1, B * block * 2)
    same = all(bool(torch.equal(v[k][:min(B, nframes - k * B) * block * 2], d_pcm1.view(-1)[:min(B, nframes - k * B) * block * 2])) for k in range(v.shape[0]))
    rows.append(dict(frames=nframes, stream_MB=round(nbytes / 1e6, 1), ms_wall=round(best[0], 3), ms_scan=round(best[1], 3), ms_decode=round(best[2], 3), ms_place=round(best[3], 3),
                     Gsamples_per_s=round(nframes * block / best[0] / 1e6, 2), ok=bool(ok), pcm_equals_input=same, sync_codes=int(r.candidates), level=LEVEL))
    print(json.dumps(rows[-1]), flush=True)
    del d_dec, d_stream
