"""Scratch: FM chain timing on 2^24 samples @2.4 Msps."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import fm, synth
n = 1 << 24
iq = synth.fm_iq(n, 2.4e6, seed=98)
x = torch.from_numpy(iq.view(np.float32)).cuda()
r = fm.FmRadio(samp_rate=2.4e6, max_samples=n)
r.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(3): r.process(x)  # streaming object: a continuous stream of repeats
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K): r.process(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("FM 2^24: %.3f ms/step, %.1f Msps, %.1f GB/s" % (dt * 1e3, n / dt / 1e6, n * 8 / dt / 1e9))
