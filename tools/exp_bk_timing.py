import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdr_amd import chain
dev = torch.device("cuda", 0)
iq, pkts = bench.make_batch(torch, dev, 1250, 1000)
taps = chain.low_pass(1.0, bench.FS, 20e3, 40e3)
c = chain.Chain(bench.FS, decim=50, center_freq=bench.F_OFF, taps=taps, decoder=int(os.environ.get("SDR_DEC", "1")), max_batch=1250, max_frames=4, max_samples=bench.N_PER_CAPTURE)
for _ in range(2):
    c.batch_process(iq)
c.synchronize()
