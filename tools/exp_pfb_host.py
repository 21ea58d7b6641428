"""Scratch: is the overlapped channelizer step bound by the host's launch calls?  Time to ENQUEUE K steps against the
time until the GPU has finished them."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
n, fs = 1 << 26, 20e6
iq = torch.randn((1, n, 2), device=dev, dtype=torch.float32) * 0.1
taps = chain.low_pass(1.0, fs, 100e3, 100e3)
c = chain.Chain(fs, taps=taps, keep_n=16, decoder=chain.DEC_TS33C, decoder_rate=fs / 64 / 16, max_samples=n, max_frames=64, channels=64)
c.set_stream(torch.cuda.current_stream().cuda_stream)
rules = [chain.DEC_SOMFY, chain.DEC_BERNER, chain.DEC_MUMBI]
def step():
    c.batch_process(iq)
    c.redecode_multi(rules)
for ov in (False, True, False, True):
    c.set_overlap(ov)
    for _ in range(300):
        step()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("overlap %-5s enqueue %.1f us per step, until done %.1f us per step" % (ov, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), flush=True)
c.close()
