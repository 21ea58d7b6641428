"""Scratch: the headline front end with 1 / 9 / 33 / 61 taps -- how much of its time is the staging alone."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdr_amd import chain
dev = torch.device("cuda", 0)
ncap = 1250
iq = torch.randn((ncap, bench.N_PER_CAPTURE, 2), device=dev, dtype=torch.float32) * 0.1
fmt = os.environ.get("SDR_FMT", "fc32")
if fmt == "cu8":
    iq = torch.randint(100, 156, (ncap, bench.N_PER_CAPTURE, 2), device=dev, dtype=torch.uint8)
full = chain.low_pass(1.0, bench.FS, 20e3, 40e3)
for nt in (61, 1, 9, 33, 61):
    taps = np.ascontiguousarray(full[:nt])
    c = chain.Chain(bench.FS, decim=int(os.environ.get("SDR_D", "50")), center_freq=bench.F_OFF, taps=taps, decoder=chain.DEC_NONE, max_batch=ncap, max_frames=4, max_samples=bench.N_PER_CAPTURE, input_format=fmt)
    c.profile_enable(True)
    for _ in range(150):
        c.batch_process(iq)
    f, b = c.profile_read()
    ms = np.mean(f[-60:])
    print("ntaps %3d front %.4f ms  %.0f GB/s" % (nt, ms, (2.02 if fmt == "cu8" else 8.02) * ncap * bench.N_PER_CAPTURE / ms / 1e6), flush=True)
    c.close()
