"""Scratch: the headline batch (1250 x 2^18, TS33C) with sdr_set_overlap off / on; SDR_LIB_PATH picks the library build."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdr_amd import chain, _lib
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
ncap = 1250
iq, pkts = bench.make_batch(torch, dev, ncap, seed=1000)
taps = chain.low_pass(1.0, bench.FS, 20e3, 40e3)
c = chain.Chain(bench.FS, decim=bench.DECIM, center_freq=bench.F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_batch=ncap,
                max_frames=4, max_samples=bench.N_PER_CAPTURE)
c.set_stream(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    for ov in (False, True):
        c.set_overlap(ov)
        for _ in range(3):
            c.batch_process(iq)
        torch.cuda.synchronize()
        c.profile_enable(True)
        t0 = time.perf_counter()
        K = 40
        for _ in range(K):
            c.batch_process(iq)
        c.synchronize() if hasattr(c, "synchronize") else None
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        f, b = c.profile_read()
        c.profile_enable(False)
        frames = c.fetch_frames()
        ok = sum(1 for fr in frames if fr.status == 0)
        print("%s overlap %-5s step %.4f ms  front %.4f back %.4f  ok frames %d" % (os.path.basename(_lib.LIB_PATH), ov, dt, np.mean(f), np.mean(b), ok), flush=True)
c.close()
