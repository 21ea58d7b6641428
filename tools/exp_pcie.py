"""Scratch: the host-pointer path (numpy input copied through the handle's staging buffer) for the DESIGN note."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain
FS, n, ncap = 1e6, 1 << 18, 256
taps = chain.low_pass(1.0, FS, 20e3, 40e3)
rng = np.random.default_rng(1)
for fmt in ("fc32", "cu8"):
    x = (rng.standard_normal((ncap, n, 2)) * 0.1).astype(np.float32).view(np.complex64).reshape(ncap, n) if fmt == "fc32" \
        else rng.integers(100, 156, (ncap, n, 2), dtype=np.uint8)
    with chain.Chain(FS, decim=50, center_freq=100e3, taps=taps, decoder=chain.DEC_TS33C, max_batch=ncap, max_samples=n,
                     input_format=fmt) as c:
        for _ in range(2):
            c.batch_process(x)
        c.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            c.batch_process(x)
        c.synchronize()
        dt = (time.perf_counter() - t0) / 5
    print("%s host input: %.1f ms per %d captures = %.1f Gsps (%.1f GB/s over PCIe)" % (fmt, dt * 1e3, ncap, ncap * n / dt / 1e9, x.nbytes / dt / 1e9))
