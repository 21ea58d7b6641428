"""Randomised parity sweep on the GPU: random filter length / decimation / demodulator / keep_n / input
format / lengths / batch strides / streaming cuts, every sliced byte stream (and the demodulated floats)
compared with the oracle bit for bit.  usage: python tools/fuzz_parity.py [cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dsp as odsp
from sdr_amd import chain

FS = 1e6


TOTAL = [0, 0]


def special(rng, k, raw, z, taps, D, fc, cplx, quad, keep, n, ncap, lo, hi, gain, sq, post):
    desc = "case %d: fc32 D=%d ntaps=%d %s %s keep=%d n=%d ncap=%d squelch=%s post=%s" % (
        k, D, taps.size, "xlating" if cplx else "real", "quad" if quad else "mag2", keep, n, ncap, sq,
        None if post is None else post.size)
    x = raw.view(np.complex64).reshape(ncap, n)
    kw = dict(decim=D, center_freq=fc, taps=taps, demod=chain.DEMOD_QUAD if quad else chain.DEMOD_MAG2, demod_gain=gain,
              thr=(lo, hi), keep_n=keep, max_samples=n, max_batch=ncap)
    if sq:
        kw["squelch"] = sq
    else:
        kw["post_taps"] = post
    with chain.Chain(FS, **kw) as c:
        c.batch_process(x)
        b = c.fetch_bits().reshape(ncap, -1)
        counts = c.fetch_squelch_counts() if sq else None
    for i in range(ncap):
        y = odsp.freq_xlating_fir(z[i], taps, fc, FS, D) if cplx else odsp.fir_ccf(z[i], taps, D)
        if sq:
            y = odsp.pwr_squelch(y, sq[0], sq[1], sq[2])
            if int(counts[i]) != y.size:
                return desc + " squelch count %d vs oracle %d" % (int(counts[i]), y.size)
        v = odsp.quad_demod(y, gain) if quad else odsp.mag2(y)
        if post is not None:
            v = odsp.fir_fff(v, post, 1)
        wb, _, _ = odsp.slice_bits(v, 1.0 if quad else gain, lo, hi, keep)
        TOTAL[1] += wb.size
        if not np.array_equal(b[i][:wb.size], wb):
            return desc + " bits differ (capture %d)" % i
    return None


def one(rng, k):
    fmt = "cu8" if rng.random() < 0.5 else "fc32"
    D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 10, 16, 25, 50, 64, 100]))
    nt = int(rng.choice([1, 2, 7, 8, 9, 27, 61, 63, 64, 65, 97, 131]))
    cplx = rng.random() < 0.5
    quad = rng.random() < 0.3
    keep = int(rng.choice([1, 1, 2, 5, 10]))
    n = int(rng.integers(1, 60000)) if rng.random() < 0.7 else int(rng.integers(60000, 400000))
    ncap = int(rng.choice([1, 1, 2, 3]))
    fc = float(rng.choice([37e3, 100e3, -212.5e3])) if cplx else 0.0
    taps = rng.standard_normal(nt).astype(np.float32) / np.float32(max(1.0, nt ** 0.5))
    if fmt == "cu8":
        raw = rng.integers(0, 256, (ncap, n, 2), dtype=np.uint8)
        z = [odsp.cu8_to_fc32(raw[i]) for i in range(ncap)]
    else:
        raw = (rng.standard_normal((ncap, n, 2)) * 0.5).astype(np.float32)
        z = [raw[i].view(np.complex64).reshape(-1) for i in range(ncap)]
    lo, hi = (-0.3, 0.3) if quad else (0.2, 0.5)
    gain = np.float32(0.7)
    # one in five chains carries a power squelch in front of the demodulator, one in five a float FIR
    # behind mag^2 (TFA/decoder.grc and TFA/analyzer2.grc wirings); both bit-exact checks on the batch path
    extra = rng.random()
    sq = (float(rng.choice([-12.0, -6.0, 0.0])), float(rng.choice([1.0, 0.3, 0.01])), bool(rng.integers(0, 2))) \
        if extra < 0.2 and fmt == "fc32" else None
    post = (rng.standard_normal(int(rng.choice([1, 3, 7, 32]))) / 3).astype(np.float32) \
        if 0.2 <= extra < 0.4 and not quad and fmt == "fc32" else None
    if sq or post is not None:
        keep = 1 if post is not None else keep
        return special(rng, k, raw, z, taps, D, fc, cplx, quad, keep, n, ncap, lo, hi, gain, sq, post)
    desc = "case %d: %s D=%d ntaps=%d %s %s keep=%d n=%d ncap=%d" % (k, fmt, D, nt, "xlating" if cplx else "real",
                                                                    "quad" if quad else "mag2", keep, n, ncap)
    want_v, want_b = [], []
    for i in range(ncap):
        y = odsp.freq_xlating_fir(z[i], taps, fc, FS, D) if cplx else odsp.fir_ccf(z[i], taps, D)
        v = odsp.quad_demod(y, gain) if quad else odsp.mag2(y) * gain
        b, _, _ = odsp.slice_bits(v, 1.0, lo, hi, keep)
        want_v.append(np.asarray(v, np.float32))
        want_b.append(b)
    kw = dict(decim=D, center_freq=fc, taps=taps, demod=chain.DEMOD_QUAD if quad else chain.DEMOD_MAG2, demod_gain=gain,
              thr=(lo, hi), keep_n=keep, tap_demod=True, max_samples=n, input_format=fmt)
    x = raw if fmt == "cu8" else raw.view(np.complex64).reshape(ncap, n)
    with chain.Chain(FS, max_batch=ncap, **kw) as c:
        c.batch_process(x)
        v = c.fetch_demod().reshape(ncap, -1)
        b = c.fetch_bits().reshape(ncap, -1)
    for i in range(ncap):
        nv = want_v[i].size
        TOTAL[0] += nv
        TOTAL[1] += want_b[i].size
        if not np.array_equal(v[i][:nv].view(np.uint32), want_v[i].view(np.uint32)):
            bad = np.flatnonzero(v[i][:nv].view(np.uint32) != want_v[i].view(np.uint32))
            # quad demod: NaN-free inputs, so words must match; report
            return desc + " BATCH demod differs at %d of %d (first %d)" % (bad.size, nv, bad[0])
        if not np.array_equal(b[i][:want_b[i].size], want_b[i]):
            return desc + " BATCH bits differ"
    # streaming with ragged cuts on capture 0
    cuts = np.sort(rng.integers(0, n + 1, int(rng.integers(1, 6))))
    with chain.Chain(FS, **kw) as c:
        got = []
        for a, e in zip(np.r_[0, cuts], np.r_[cuts, n]):
            if e > a:
                c.process_iq(x[0][a:e])
                got.append(c.fetch_bits())
        got = np.concatenate(got) if got else np.zeros(0, np.int8)
    if not np.array_equal(got, want_b[0]):
        return desc + " STREAM bits differ (cuts %s): %d vs %d" % (cuts.tolist(), got.size, want_b[0].size)
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, fails = time.time(), 0
    for k in range(cases):
        r = one(rng, k)
        if r:
            fails += 1
            print("FAIL", r, flush=True)
        if k % 10 == 9:
            print("%d cases, %d failures, %.0f s" % (k + 1, fails, time.time() - t0), flush=True)
    print("done: %d cases, %d failures; %d demodulated floats and %d sliced bytes compared" % (cases, fails, TOTAL[0], TOTAL[1]))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
