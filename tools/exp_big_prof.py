"""Scratch: front/back kernel split of the one-long-capture case of tools/exp_big.py."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import exp_big
from sdr_amd import chain
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
iq, pkts = exp_big.build(log2n, torch.device("cuda", 0))
n = iq.shape[0]
taps = chain.low_pass(1.0, exp_big.FS, 20e3, 40e3)
with chain.Chain(exp_big.FS, decim=exp_big.D, center_freq=exp_big.F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_samples=n,
                 max_frames=4 * len(pkts) + 8) as c:
    c.profile_enable(True)
    for _ in range(4):
        c.batch_process(iq.reshape(1, n, 2))
    f, b = c.profile_read()
    print("2^%d samples, one capture: front %.3f ms (%.0f GB/s), back %.3f ms, %d frames" % (log2n, f[-1], 8.02 * n / f[-1] / 1e6, b[-1], len(c.fetch_frames())))
