"""Scratch: does a back end (decode-only handle, stream 1) hide beside a front end (slicer-only handle, stream 0)?"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdr_amd import chain
dev = torch.device("cuda", 0)
ncap = 1250
iq, pkts = bench.make_batch(torch, dev, ncap, seed=1000)
taps = chain.low_pass(1.0, bench.FS, 20e3, 40e3)
kw = dict(decim=bench.DECIM, center_freq=bench.F_OFF, taps=taps, max_batch=ncap, max_frames=4, max_samples=bench.N_PER_CAPTURE)
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
F = chain.Chain(bench.FS, decoder=chain.DEC_NONE, **kw)
D = chain.Chain(bench.FS, decoder=chain.DEC_TS33C, **kw)
torch.cuda.synchronize()
D.batch_process(iq)
D.synchronize()
F.set_stream(s0.cuda_stream)
D.set_stream(s1.cuda_stream)
def run(f, d, n=40):
    for _ in range(3):
        if f: F.batch_process(iq)
        if d: D.redecode(chain.DEC_TS33C, 20000.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if f: F.batch_process(iq)
        if d: D.redecode(chain.DEC_TS33C, 20000.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("front only %.4f ms   decode only %.4f ms   both, two streams %.4f ms" % (run(1, 0), run(0, 1), run(1, 1)))
fr = D.fetch_frames()
print(len(fr), "frames")
