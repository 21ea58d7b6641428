"""Run-to-run determinism: every chain shape of the bench (and the kernels the bench does not reach) over the same
input again and again, a large unrelated batch in between to disturb caches and timing; sliced bytes, demodulated
floats and frames must repeat bit for bit.  usage: python3 tools/stress_repeat.py [reps]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain



def run(reps, verbose=True):
    """-> number of repeats (over all shapes) whose outputs differed from the shape's first run"""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)


    def noise(ncap, n, scale=0.3):
        return torch.randn((ncap, n, 2), generator=g, device=dev, dtype=torch.float32) * scale


    def cu8(ncap, n):
        return torch.randint(0, 256, (ncap, 2 * n), generator=g, device=dev, dtype=torch.uint8)


    lp = lambda fs, fc, tw: chain.low_pass(1.0, fs, fc, tw)
    ctaps = (lp(1e6, 20e3, 40e3).astype(np.complex64) * np.exp(2j * np.pi * 0.07 * np.arange(lp(1e6, 20e3, 40e3).size))).astype(np.complex64)
    shapes = [
        ("headline D=50 xlating", dict(fs=1e6, decim=50, center_freq=100e3, taps=lp(1e6, 20e3, 40e3), decoder=chain.DEC_TS33C), noise(96, 1 << 18)),
        ("D=10 xlating", dict(fs=1e6, decim=10, center_freq=100e3, taps=lp(1e6, 20e3, 40e3), decoder=chain.DEC_SOMFY), noise(96, 1 << 18)),
        ("cu8 D=50", dict(fs=1e6, decim=50, center_freq=100e3, taps=lp(1e6, 20e3, 40e3), decoder=chain.DEC_TS33C, input_format="cu8"), cu8(96, 1 << 18)),
        ("R shape 121 taps D=1", dict(fs=2e6, decim=1, taps=lp(2e6, 20e3, 40e3), keep_n=100, decoder=chain.DEC_TS33C, decoder_rate=2e4), noise(8, 1 << 19)),
        ("TFA 27 taps D=1 quad", dict(fs=2e6, decim=1, taps=chain.low_pass(1, 2e6, 500e3, 250e3, chain.WIN_BLACKMAN), demod=chain.DEMOD_QUAD, demod_gain=3.0, thr=(-0.2, 0.2),
                                      decoder=chain.DEC_TFA_GRC), noise(8, 1 << 20)),
        ("xlating D=1 121 taps", dict(fs=1e6, decim=1, center_freq=100e3, taps=lp(1e6, 20e3, 40e3), keep_n=50, decoder=chain.DEC_TS33C, decoder_rate=2e4), noise(8, 1 << 19)),
        ("D=4 131 taps, no demod (two-team kernel)", dict(fs=2.4e6, decim=4, taps=lp(2.4e6, 250e3, 100e3), demod=chain.DEMOD_NONE, tap_demod=True), noise(2, 1 << 21)),
        ("D=2 mfma quad", dict(fs=1e6, decim=2, taps=lp(1e6, 100e3, 50e3), demod=chain.DEMOD_QUAD, demod_gain=1.0, thr=(-0.2, 0.2), tap_demod=True), noise(8, 1 << 19)),
        ("post-demod filter (TFA analyzer2 wiring)", dict(fs=2e6, decim=1, center_freq=500e3, demod=chain.DEMOD_MAG2, thr=(0.2, 0.4), keep_n=1,
                                                      post_taps=chain.low_pass(1.0, 2e6, 100e3, 300e3), tap_demod=True, decoder=chain.DEC_TFA_GRC), noise(4, 1 << 19)),
        ("prefilter + resampler 1/10 (mumbi FS300 wiring)", dict(fs=2e6, pre_taps=lp(2e6, 100e3, 50e3), pre_decim=1, decim=10, taps=lp(2e6, 80e3, 40e3), thr=(0.1, 0.4),
                                                             decoder=chain.DEC_MUMBI), noise(4, 1 << 19)),
        ("gating squelch + discriminator", dict(fs=2e6, decim=1, taps=chain.low_pass(1, 2e6, 500e3, 250e3, chain.WIN_BLACKMAN), demod=chain.DEMOD_QUAD, demod_gain=3.0,
                                            thr=(-0.2, 0.2), squelch=(-12.0, 1e-3), decoder=chain.DEC_TFA_GRC), noise(4, 1 << 19)),
        ("cu8 D=1 121 translated taps", dict(fs=1e6, decim=1, center_freq=100e3, taps=lp(1e6, 20e3, 40e3), keep_n=50, decoder=chain.DEC_TS33C, decoder_rate=2e4,
                                         input_format="cu8"), cu8(8, 1 << 18)),
        ("cu8 D=10", dict(fs=1e6, decim=10, taps=lp(1e6, 20e3, 40e3), decoder=chain.DEC_BERNER, input_format="cu8"), cu8(64, 1 << 18)),
        ("pfb64", dict(fs=20e6, taps=chain.low_pass(1.0, 20e6, 100e3, 100e3), keep_n=16, decoder=chain.DEC_TS33C, decoder_rate=20e6 / 64 / 16, channels=64), noise(1, 1 << 23)),
    ]
    big = noise(64, 1 << 18)
    bigc = chain.Chain(1e6, decim=50, center_freq=100e3, taps=lp(1e6, 20e3, 40e3), decoder=chain.DEC_TS33C, max_batch=64, max_frames=4, max_samples=1 << 18)
    bad_total = 0
    for name, kw, x in shapes:
        kw = dict(kw)
        fs = kw.pop("fs")
        ncap, n = x.shape[0], (x.shape[1] if x.dtype != torch.uint8 else x.shape[1] // 2)
        try:
            c = chain.Chain(fs, max_batch=ncap, max_frames=8, max_samples=n, **kw)
        except TypeError as e:
            print(name, "skipped:", e)
            continue
        ref, bad = None, 0
        for it in range(reps):
            bigc.batch_process(big)
            c.batch_process(x)
            out = [c.fetch_bits().copy()] if kw.get("demod", 1) != chain.DEMOD_NONE else []
            if kw.get("tap_demod"):
                out.append(c.fetch_demod().view(np.uint32).copy())
            if kw.get("decoder"):
                try:
                    fr = c.fetch_frames()
                except Exception as e:          # noise can overflow a frame table: the error text must repeat too
                    fr, out = [], out + [np.frombuffer(str(e).encode(), np.uint8)]
                out.append(np.frombuffer(b"".join(bytes([f.capture & 255, f.status & 255]) + f.packet() for f in fr), np.uint8))
            if ref is None:
                ref = out
            elif not all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(ref, out)):
                bad += 1
        if verbose:
            print("%-45s %d repeats, %d differ" % (name, reps, bad), flush=True)
        bad_total += bad
        c.close()
    bigc.close()
    return bad_total


if __name__ == "__main__":
    n_bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
    print("total differing repeats:", n_bad)
    sys.exit(1 if n_bad else 0)
