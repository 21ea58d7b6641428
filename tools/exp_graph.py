"""Scratch: sdr_batch_process_static under HIP graph capture (torch.cuda.graph), replayed; step time against direct calls."""
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdr_amd import chain, _lib
dev = torch.device("cuda", 0)
ncap = int(os.environ.get("NCAP", "1250"))
iq, pkts = bench.make_batch(torch, dev, ncap, seed=1000)
taps = chain.low_pass(1.0, bench.FS, 20e3, 40e3)
c = chain.Chain(bench.FS, decim=bench.DECIM, center_freq=bench.F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_batch=ncap, max_frames=4,
                max_samples=bench.N_PER_CAPTURE)
nb = bench.N_PER_CAPTURE // bench.DECIM
bits = torch.zeros(ncap * nb, dtype=torch.uint8, device=dev)
rec = torch.zeros(ncap * 4 * C.sizeof(_lib.Frame), dtype=torch.uint8, device=dev)
cnt = torch.zeros(ncap, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
c.set_stream(st.cuda_stream)
with torch.cuda.stream(st):
    c.batch_process_static(iq, bits, rec, cnt)
st.synchronize()
want = (bits.clone(), rec.clone(), cnt.clone())
print("direct call: frames", int(cnt.sum()))
for t in (bits, rec, cnt):
    t.zero_()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    c.batch_process_static(iq, bits, rec, cnt)
torch.cuda.synchronize()
print("captured; outputs untouched by the capture:", int(cnt.sum()) == 0)
g.replay()
torch.cuda.synchronize()
print("replay equals direct:", all(torch.equal(a, b) for a, b in zip(want, (bits, rec, cnt))))
def timed(f, n=40):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def direct():
    with torch.cuda.stream(st):
        c.batch_process_static(iq, bits, rec, cnt)
print("direct %.4f ms/step   graph replay %.4f ms/step   sdr_batch_process %.4f ms/step" % (timed(direct), timed(g.replay), timed(lambda: c.batch_process(iq))))
