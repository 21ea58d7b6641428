"""Scratch: back-end launches of the channelizer rows one by one and as one multi launch."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdr_amd import chain
dev = torch.device("cuda", 0)
n, fs = 1 << 26, 20e6
import bench
iq = bench.make_wideband(torch, dev, n, fs)[0].reshape(1, n, 2) if os.environ.get("REAL", "1") == "1" else torch.randn((1, n, 2), device=dev, dtype=torch.float32) * 0.1
taps = chain.low_pass(1.0, fs, 100e3, 100e3)
rate = fs / 64 / 16
c = chain.Chain(fs, taps=taps, keep_n=16, decoder=chain.DEC_TS33C, decoder_rate=rate, max_samples=n, max_frames=64, channels=64)
c.batch_process(iq)
rules = [chain.DEC_SOMFY, chain.DEC_BERNER, chain.DEC_MUMBI]
for label, fn in [("somfy", lambda: c.redecode(rules[0], rate)), ("berner", lambda: c.redecode(rules[1], rate)),
                  ("mumbi", lambda: c.redecode(rules[2], rate)), ("ts33c", lambda: c.redecode(chain.DEC_TS33C, rate)),
                  ("multi3", lambda: c.redecode_multi(rules, rate)), ("multi1", lambda: c.redecode_multi(rules[:1], rate)),
                  ("multi4", lambda: c.redecode_multi(rules + [chain.DEC_TS33C], rate))]:
    for _ in range(3):
        fn()
    c.profile_enable(True)
    for _ in range(10):
        fn()
    f, b = c.profile_read()
    c.profile_enable(False)
    print("%-7s %.4f ms" % (label, float(np.mean(b))), flush=True)
