"""Scratch: the channelizer kernel alone (for rocprofv3 --pmc passes)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain
dev = torch.device("cuda", 0)
n = 1 << 26
fs = 20e6
iq = torch.randn((1, n, 2), device=dev, dtype=torch.float32) * 0.1
taps = chain.low_pass(1.0, fs, 100e3, 100e3)
for dec in (chain.DEC_NONE, chain.DEC_TS33C, chain.DEC_MUMBI):
    c = chain.Chain(fs, taps=taps, keep_n=16, decoder=dec, decoder_rate=fs / 64 / 16, max_samples=n, max_frames=64,
                    channels=64)
    c.profile_enable(True)
    for _ in range(8):
        c.batch_process(iq)
    f, b = c.profile_read()
    ms = np.mean(f[2:])
    print("decoder %d: k_pfb %.4f ms  %.0f GB/s   k_backend %.4f ms" % (dec, ms, 8.0625 * n / ms / 1e6, np.mean(b[2:])))
    c.close()
