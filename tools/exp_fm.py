"""Scratch: FM chain timing (2^26 samples @2.4 Msps, streaming object) with a per-kernel breakdown under
rocprofv3 (tools/trace_by_grid.py) when run there."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import fm, synth
n, m = 1 << int(os.environ.get("LOG2N", "26")), 1 << 22
iq = synth.fm_iq(m, 2.4e6, seed=98)
x = torch.from_numpy(iq.view(np.float32)).cuda().repeat(n // m, 1)
r = fm.FmRadio(samp_rate=2.4e6, max_samples=n)
for _ in range(3): r.process(x)
r._L.sdr_fm_fetch(r._h, 4, None, 0, __import__("ctypes").byref(__import__("ctypes").c_size_t()))   # synchronises the chain's stream
t0 = time.perf_counter()
K = 10
for _ in range(K): r.process(x)
r.fetch(4)
dt = (time.perf_counter() - t0) / K
print("FM 2^%d: %.3f ms/step, %.1f Msps, %.1f GB/s" % (int(np.log2(n)), dt * 1e3, n / dt / 1e6, n * 8 / dt / 1e9))
