"""Scratch: front-end ablations (SDR_DBG) and tile-size variants."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdr_amd import chain
dev = torch.device("cuda", 0)
ncap = 1250
iq = torch.randn((ncap, bench.N_PER_CAPTURE, 2), device=dev, dtype=torch.float32) * 0.1
taps = chain.low_pass(1.0, bench.FS, 20e3, 40e3)
fmt = os.environ.get("SDR_FMT", "fc32")
if fmt == "cu8":
    iq = torch.randint(100, 156, (ncap, bench.N_PER_CAPTURE, 2), device=dev, dtype=torch.uint8)
c = chain.Chain(bench.FS, decim=int(os.environ.get("SDR_D", "50")), center_freq=bench.F_OFF, taps=taps, decoder=chain.DEC_NONE, max_batch=ncap, max_frames=4, max_samples=bench.N_PER_CAPTURE, input_format=fmt)
c.profile_enable(True)
for _ in range(60):
    c.batch_process(iq)
f, b = c.profile_read()
ms = np.mean(f[10:]); print("back %.4f ms" % np.mean(b[10:]))
print(fmt, "PIPE=%s TPW=%s" % (os.environ.get("SDR_PIPE"), os.environ.get("SDR_TPW")), "SDR_DBG=%s front %.4f ms  %.0f GB/s" % (os.environ.get("SDR_DBG", "0"), ms, 8.02 * ncap * bench.N_PER_CAPTURE / ms / 1e6))
