"""Scratch: front-end kernel time for small-decimation shapes (fp32-VALU bound)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain  # noqa: E402

dev = torch.device("cuda", 0)


def run(name, ncap, n, D, taps, fc=0.0, demod=chain.DEMOD_MAG2, keep_n=1, reps=8):
    iq = torch.randn((ncap, n, 2), device=dev, dtype=torch.float32) * 0.1
    c = chain.Chain(1e6, decim=D, center_freq=fc, taps=taps, demod=demod, keep_n=keep_n, decoder=chain.DEC_NONE,
                    max_batch=ncap, max_samples=n, tap_demod=(demod == chain.DEMOD_NONE))
    c.profile_enable(True)
    for _ in range(reps + 2):
        c.batch_process(iq)
    f, _ = c.profile_read()
    ms = float(np.mean(f[2:]))
    ns = ncap * n
    cplx = fc != 0.0
    flop = (8 if cplx else 4) * len(taps) / D * ns
    print("%-28s D=%-3d taps=%-4d  %.3f ms  %7.1f Gsps  %6.0f GB/s  %6.1f TFLOP/s" % (
        name, D, len(taps), ms, ns / ms / 1e6, 8 * ns / ms / 1e6, flop / ms / 1e9), flush=True)
    c.close()
    del iq
    torch.cuda.empty_cache()


t121 = chain.low_pass(1.0, 1e6, 20e3, 40e3 / 2)   # 121 taps at fs 1e6 needs tw 20k
t121 = chain.low_pass(1.0, 2e6, 20e3, 40e3)
t61 = chain.low_pass(1.0, 1e6, 20e3, 40e3)
t97 = chain.low_pass(1.0, 2e6, 100e3, 50e3)
t27 = chain.low_pass(1.0, 2e6, 500e3, 250e3, chain.WIN_BLACKMAN)
t131 = chain.resampler_taps(1, 4)
t329 = chain.resampler_taps(1, 10)
which = sys.argv[1:] or ["all"]
if "all" in which or "r" in which:
    run("R-shape ts33c (ccf)", 64, 1 << 18, 1, t121, keep_n=100)
    run("R-shape 256 captures", 256, 1 << 18, 1, t121, keep_n=100)
    run("R-shape 64 none", 64, 1 << 18, 1, t121, demod=chain.DEMOD_NONE)
    run("R-shape one stream 2^24", 1, 1 << 24, 1, t121, keep_n=100)
    run("tfa lpf 27 quad", 64, 1 << 18, 1, t27, demod=chain.DEMOD_QUAD)
    run("mumbi stage1 97 none", 64, 1 << 18, 1, t97, demod=chain.DEMOD_NONE)
    run("xlating D=1 121 cplx", 64, 1 << 18, 1, t121, fc=100e3, keep_n=100)
if "all" in which or "d" in which:
    run("fm stageA 131 D=4 none", 64, 1 << 18, 4, t131, demod=chain.DEMOD_NONE)
    run("D=2 61", 64, 1 << 18, 2, t61)
    run("mumbi stage2 329 D=10", 64, 1 << 18, 10, t329)
    run("D=8 121", 64, 1 << 18, 8, t121)
    run("F-shape D=50 61 xl", 1250, 1 << 18, 50, t61, fc=100e3)
