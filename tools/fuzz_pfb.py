"""Randomised sweep of the 64-channel polyphase channelizer on the GPU: random prototype lengths (1 to 1024
taps: every branch count the kernel instantiates and ragged last branches), capture lengths from below one
frame of 64 to many tiles, batches, keep_n; channel floats and sliced bytes bit-identical to the oracle.
usage: python tools/fuzz_pfb.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dsp as odsp
from sdr_amd import chain

FS, M = 20e6, 64
TOTAL = [0]


def one(rng, k):
    ntaps = int(rng.choice([1, 63, 64, 65, 200, 256, 257, 481, 512, 513, 777, 1000, 1024]))
    n = int(rng.integers(1, 64 * 40)) if rng.random() < 0.3 else int(rng.integers(64 * 40, 64 * 6000))
    ncap = int(rng.choice([1, 1, 2, 3]))
    keep = int(rng.choice([1, 2, 16, 25]))
    desc = "case %d: ntaps=%d n=%d ncap=%d keep=%d" % (k, ntaps, n, ncap, keep)
    x = ((rng.standard_normal((ncap, n)) + 1j * rng.standard_normal((ncap, n))) * 0.5).astype(np.complex64)
    taps = (rng.standard_normal(ntaps) / max(1.0, ntaps ** 0.5) * 4).astype(np.float32)
    with chain.Chain(FS, taps=taps, demod=chain.DEMOD_NONE, tap_demod=True, max_batch=ncap, max_samples=n, channels=M) as c:
        c.batch_process(x)
        y = c.fetch_demod().reshape(ncap, M, -1)
    want = [odsp.pfb_channelize(x[i], taps, M) for i in range(ncap)]
    for i in range(ncap):
        TOTAL[0] += want[i].size
        if y[i].shape != want[i].shape or not np.array_equal(y[i].view(np.uint32), want[i].view(np.uint32)):
            return desc + " channel floats differ (capture %d)" % i
    lo, hi = 0.3, 0.8
    with chain.Chain(FS, taps=taps, keep_n=keep, thr=(lo, hi), max_batch=ncap, max_samples=n, channels=M) as c:
        c.batch_process(x)
        b = c.fetch_bits().reshape(ncap, M, -1) if n >= M else None
    for i in range(ncap):
        for ch in range(M if n >= M else 0):
            wb, _, _ = odsp.slice_bits(odsp.mag2(want[i][ch]), 1.0, lo, hi, keep)
            if not np.array_equal(b[i][ch][:wb.size], wb):
                return desc + " sliced bytes differ (capture %d channel %d)" % (i, ch)
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0, fails = time.time(), 0
    for k in range(cases):
        r = one(rng, k)
        if r:
            fails += 1
            print("FAIL", r, flush=True)
        if k % 10 == 9:
            print("%d cases, %d failures, %.0f s" % (k + 1, fails, time.time() - t0), flush=True)
    print("done: %d cases, %d failures, %d channel samples compared" % (cases, fails, TOTAL[0]))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
