"""Scratch experiment: back-end kernel time by decoder (phase split)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdr_amd import chain
dev = torch.device("cuda", 0)
ncap = int(sys.argv[1]) if len(sys.argv) > 1 else 1250
iq, pkts = bench.make_batch(torch, dev, ncap, 1000)
taps = chain.low_pass(1.0, bench.FS, 20e3, 40e3)
for name, dec in [("none", chain.DEC_NONE), ("berner", chain.DEC_BERNER), ("ts33c", chain.DEC_TS33C)]:
    c = chain.Chain(bench.FS, decim=50, center_freq=bench.F_OFF, taps=taps, decoder=dec, max_batch=ncap, max_frames=4, max_samples=bench.N_PER_CAPTURE)
    c.profile_enable(True)
    for _ in range(10):
        c.batch_process(iq)
    f, b = c.profile_read()
    print(name, "front %.4f ms back %.4f ms" % (np.mean(f[2:]), np.mean(b[2:])))
    c.close()
