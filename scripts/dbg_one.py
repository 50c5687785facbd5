import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import flac_amd, signals
NF, N = int(sys.argv[1]), 4096
ch = int(sys.argv[2]); kw = dict(mid_side=0) if len(sys.argv) > 3 else {}
base = signals.music(64 * N, ch, 16, seed=5)
pcm = np.tile(base, ((NF + 63) // 64, 1))[: NF * N]
eng = flac_amd.FrameEngine(flac_amd.make_settings(ch, 16, 48000, 8, **kw), device=0, max_batch_frames=NF)
d_pcm = torch.from_numpy(pcm).cuda()
cap = eng.max_output_bytes(NF)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
d_fb = torch.empty(NF, dtype=torch.int32, device="cuda")
d_tot = torch.zeros(1, dtype=torch.int64, device="cuda")
for _ in range(3):
    eng.encode_device(d_pcm.data_ptr(), NF, d_out.data_ptr(), cap, d_fb.data_ptr(), d_tot.data_ptr())
torch.cuda.synchronize()
print(eng.last_phase_ms())
