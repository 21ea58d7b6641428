"""One very long capture (lengths up to 2^28 samples): 2^LOG2N samples at 1 Msps with a TS33C
transmission every 0.5 s, through the headline chain in ONE batch call (64-bit indexing, the segment-parallel
run-length path on a multi-MiB row, hundreds of frames from one row), then the same capture streamed in
ragged chunks; every frame checked against what was sent.  usage: python tools/exp_big.py [log2n]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain, synth, _lib

FS, F_OFF, D = 1e6, 100e3, 50


def build(log2n, dev, seed=28):
    n = 1 << log2n
    rng = np.random.default_rng(seed)
    env = np.zeros(n, np.uint8)
    pkts, at = [], 5000
    while at + 400000 < n:
        pkt = synth.ts33c_packet(**synth.random_ts33c_fields(rng))
        sym = []
        for _ in range(3):
            sym += synth.ts33c_symbols(synth.ts33c_bits(pkt), rng)
        e = synth.envelope(sym, FS, lead_in_us=0.0)
        env[at:at + e.size] = e
        pkts.append(pkt)
        at += 500000 + int(rng.integers(0, 20000))
    iq = torch.empty((n, 2), dtype=torch.float32, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    step = 1 << 24
    for a in range(0, n, step):
        b = min(n, a + step)
        t = torch.arange(a, b, dtype=torch.float64, device=dev)
        ph = (2 * np.pi * F_OFF / FS) * t
        e = torch.from_numpy(env[a:b]).to(dev).to(torch.float32) * 0.7
        iq[a:b, 0] = e * torch.cos(ph).to(torch.float32)
        iq[a:b, 1] = e * torch.sin(ph).to(torch.float32)
        iq[a:b] += 0.02 * torch.randn((b - a, 2), generator=g, device=dev, dtype=torch.float32)
    return iq, pkts


def run(log2n=28):
    dev = torch.device("cuda", 0)
    iq, pkts = build(log2n, dev)
    n = iq.shape[0]
    taps = chain.low_pass(1.0, FS, 20e3, 40e3)
    want = [p for p in pkts for _ in range(3)]       # 3 repeats on air, the idle time behind them closes the third
    kw = dict(decim=D, center_freq=F_OFF, taps=taps, decoder=chain.DEC_TS33C, max_samples=n, max_frames=4 * len(pkts) + 8)
    with chain.Chain(FS, **kw) as c:
        t0 = time.time()
        c.batch_process(iq.reshape(1, n, 2))
        frames = c.fetch_frames()
        dt = time.time() - t0
        bits_one = c.fetch_bits()
    got = [f.packet() for f in frames if f.status == _lib.FRAME_OK]
    assert got == want, "one call: %d frames, expected %d" % (len(got), len(want))
    # streamed in ragged chunks (device tensors): same bytes, same frames
    rng = np.random.default_rng(5)
    cuts = np.sort(rng.integers(0, n, 7)) & ~1
    with chain.Chain(FS, **dict(kw, max_samples=int(np.max(np.diff(np.r_[0, cuts, n]))) + 2)) as c:
        bits, sframes = [], []
        for a, b in zip(np.r_[0, cuts], np.r_[cuts, n]):
            if b > a:
                c.process_iq(iq[a:b])
                bits.append(c.fetch_bits())
                sframes += [f.packet() for f in c.fetch_frames() if f.status == _lib.FRAME_OK]
    assert np.array_equal(np.concatenate(bits), bits_one), "streamed bytes differ from the one-call bytes"
    assert sframes == want, "streamed: %d frames, expected %d" % (len(sframes), len(want))
    return n, len(got), dt


if __name__ == "__main__":
    n, nf, dt = run(int(sys.argv[1]) if len(sys.argv) > 1 else 28)
    print("2^%d samples: %d frames, all as sent; one call %.3f s incl. fetch; streaming in 8 ragged chunks identical" % (int(np.log2(n)), nf, dt))
