"""Scratch: what the handle's HIP-event profiling (sdr_profile_enable) costs per step on the headline batch."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdr_amd import chain
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ncap = 1250
iq, pkts = bench.make_batch(torch, dev, ncap, seed=1000)
taps = chain.low_pass(1.0, bench.FS, 20e3, 40e3)
c = chain.Chain(bench.FS, decim=bench.DECIM, center_freq=bench.F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_batch=ncap,
                max_frames=4, max_samples=bench.N_PER_CAPTURE)
c.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(400):
    c.batch_process(iq)
torch.cuda.synchronize()
for rep in range(3):
    for prof in (False, True):
        c.profile_enable(prof)
        for _ in range(20):
            c.batch_process(iq)
        torch.cuda.synchronize()
        K = 100
        t0 = time.perf_counter()
        for _ in range(K):
            c.batch_process(iq)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        if prof:
            c.profile_read()
        print("profile events %-5s step %.4f ms" % (prof, dt), flush=True)
c.close()
